"""ctypes binding of libdib_host.so (include/dib_host.h): native host code of the hot path."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DIB_HOST_LIB: another build of the same library (the sanitizer build of `make -C detectinblur_amd/csrc asan`,
# tests/test_host_asan.py)
LIB_PATH = os.environ.get("DIB_HOST_LIB") or os.path.join(_HERE, "libdib_host.so")


class MT19937(ctypes.Structure):
    _fields_ = [("key", ctypes.c_uint32 * 624), ("pos", ctypes.c_int32), ("has_gauss", ctypes.c_int32),
                ("gauss", ctypes.c_double)]


_dp = ctypes.POINTER(ctypes.c_double)
_llp, _ip, _u8p = ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_ubyte)
_SIGNATURES = {
    "dib_coco_accumulate_cat": (ctypes.c_int, [_dp, _ip, ctypes.c_int, _u8p, _u8p, _ip, ctypes.c_int, ctypes.c_int, _ip, ctypes.c_int, _dp,
                                               ctypes.c_int, _dp, _dp]),
    "dib_coco_match_image": (ctypes.c_int, [_dp, ctypes.c_int, ctypes.c_int, _llp, _dp, _dp, _llp, _llp, _dp, _llp, ctypes.c_int, ctypes.c_int, _dp,
                                            ctypes.c_int, _dp, ctypes.c_int, _ip, _ip, _u8p, _u8p, _ip, _ip]),
    "dib_trajectory_fit": (ctypes.c_int, [ctypes.POINTER(MT19937), ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                          ctypes.c_double, _dp, _dp, _dp]),
    "dib_psf_fit": (ctypes.c_int, [_dp, ctypes.c_int, _dp, ctypes.c_int, ctypes.c_int, _dp]),
    "dib_psf_center": (ctypes.c_int, [_dp, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    "dib_rng_uniform": (ctypes.c_double, [ctypes.POINTER(MT19937)]),
    "dib_rng_gauss": (ctypes.c_double, [ctypes.POINTER(MT19937)]),
    "dib_mask_or_polygon": (ctypes.c_int, [_dp, ctypes.c_long, ctypes.c_long, ctypes.c_long, _u8p]),
    "dib_mask_or_runs": (ctypes.c_int, [ctypes.POINTER(ctypes.c_uint), ctypes.c_long, ctypes.c_long, ctypes.c_long, _u8p]),
    "dib_coco_match": (ctypes.c_int, [_dp, ctypes.c_int, ctypes.c_int, _dp, ctypes.POINTER(ctypes.c_longlong), _dp, _dp, ctypes.c_int, _dp,
                                      ctypes.c_int, ctypes.POINTER(ctypes.c_ubyte), ctypes.POINTER(ctypes.c_ubyte), ctypes.POINTER(ctypes.c_int)]),
}
EXPORTS = tuple(_SIGNATURES)
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise ImportError("%s not found: build it with `make -C detectinblur_amd/csrc`" % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def dptr(a):
    return a.ctypes.data_as(_dp)


class NumpyGlobalStream:
    """Context manager that lends numpy's legacy global RandomState to the native code and writes
    the advanced state back, so native draws interleave with `np.random.*` calls exactly like the
    reference's own draws would."""

    def __enter__(self):
        st = np.random.get_state()
        if st[0] != "MT19937":
            raise RuntimeError("numpy global RandomState is not MT19937")
        self.s = MT19937()
        ctypes.memmove(self.s.key, np.ascontiguousarray(st[1], dtype=np.uint32).ctypes.data, 624 * 4)
        self.s.pos, self.s.has_gauss, self.s.gauss = int(st[2]), int(st[3]), float(st[4])
        return ctypes.byref(self.s)

    def __exit__(self, *exc):
        key = np.frombuffer(self.s.key, dtype=np.uint32).copy()
        np.random.set_state(("MT19937", key, int(self.s.pos), int(self.s.has_gauss), float(self.s.gauss)))
        return False
