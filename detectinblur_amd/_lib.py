"""ctypes binding of libdib_hip.so (C ABI: include/dib.h).

There is deliberately NO fallback: if the HIP library is missing or fails to load, importing
any compute entry point raises.  The CPU oracle under oracle/ is test infrastructure and is
never imported from this package.
"""
import ctypes
import os

# torch must be imported BEFORE libdib_hip.so is opened: the wheel bundles its own HIP runtime
# (same SONAME as /opt/rocm's); whichever is loaded first serves the whole process, and the
# streams / device pointers we are handed belong to torch's.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# DIB_HIP_LIB: another build of the same library (diagnostic builds of scratch/: stamps, timelines); never set in production
LIB_PATH = os.environ.get("DIB_HIP_LIB") or os.path.join(_HERE, "libdib_hip.so")

DIB_F16, DIB_F32 = 0, 1
DIB_ACC_BITEXACT, DIB_ACC_FP32, DIB_ACC_FMA16, DIB_ACC_FAST16 = 0, 1, 2, 3
DIB_EINVAL, DIB_ESHAPE, DIB_EHIP, DIB_ENOT128, DIB_ECAPTURE, DIB_ETIMEOUT = -1, -2, -3, -4, -5, -6
DIB_STEP_PSFS_COMPLETE, DIB_STEP_LARGE_WINDOW = 1, 2
DIB_COMPACT_LARGE_WINDOW, DIB_WINDOW_LARGE, DIB_COMPACT_VRUNS = 8, 0x100, 16

_lib = None

_c_int_p = ctypes.POINTER(ctypes.c_int)
_c_void_pp = ctypes.POINTER(ctypes.c_void_p)

_SIGNATURES = {
    "dib_abi_version": (ctypes.c_int, []),
    "dib_last_error": (ctypes.c_char_p, []),
    "dib_tap_table_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "dib_tap_tables_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "dib_psf_compact": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.c_void_p]),
    "dib_psf_compact_list": (ctypes.c_int, [_c_void_pp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_void_p, ctypes.c_void_p]),
    "dib_sparse_blur": (ctypes.c_int, [_c_void_pp, _c_void_pp, _c_int_p, _c_int_p, _c_int_p, _c_int_p, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_void_p]),
    "dib_blur_step": (ctypes.c_int, [_c_void_pp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     _c_void_pp, _c_void_pp, _c_int_p, _c_int_p, _c_int_p, _c_int_p, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "dib_blur_step_packed": (ctypes.c_int, [_c_void_pp, _c_int_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "dib_blur_step_release": (ctypes.c_int, []),
    "dib_blur_step_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "dib_blur_step_ws": (ctypes.c_int, [_c_void_pp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        _c_void_pp, _c_void_pp, _c_int_p, _c_int_p, _c_int_p, _c_int_p, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int,
                                        ctypes.c_void_p]),
    "dib_device_status": (ctypes.c_int, [ctypes.c_int]),
    "dib_normalize_pad": (ctypes.c_int, [_c_void_pp, ctypes.c_int, _c_int_p, _c_int_p, ctypes.c_int, ctypes.POINTER(ctypes.c_float),
                                         ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_void_p]),
    "dib_normalize_resize_pad": (ctypes.c_int, [_c_void_pp, ctypes.c_int, _c_int_p, _c_int_p, _c_int_p, _c_int_p, ctypes.c_int,
                                                ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "dib_sparse_blur_normalized": (ctypes.c_int, [_c_void_pp, _c_int_p, _c_int_p, _c_int_p, _c_int_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float),
                                                  ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "dib_expand_boxes": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_void_p]),
    "dib_clamp_boxes": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "dib_psf_rasterize_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "dib_psf_rasterize": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                                         ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p]),
    "dib_roi_align_forward": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_void_p, ctypes.c_void_p]),
    "dib_roi_align_backward": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_void_p, ctypes.c_void_p]),
    "dib_roi_align_nhwc_forward": (ctypes.c_int, [_c_void_pp, _c_int_p, _c_int_p, ctypes.POINTER(ctypes.c_float), ctypes.c_int,
                                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "dib_roi_align_nhwc_backward": (ctypes.c_int, [ctypes.c_void_p, _c_int_p, _c_int_p, ctypes.POINTER(ctypes.c_float),
                                                   ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                                   ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_void_pp, ctypes.c_void_p]),
    "dib_nms_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "dib_nms": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p,
                               ctypes.c_void_p, ctypes.c_void_p]),
    "dib_nms_batched": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "dib_box_match": (ctypes.c_int, [ctypes.c_void_p, _c_int_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                     ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "dib_box_encode_matched": (ctypes.c_int, [ctypes.c_void_p, _c_int_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_void_p]),
    "dib_box_decode": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                      ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]),
    "dib_bias_act_transpose": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_longlong,
                                              ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "dib_topk_levels": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, _c_int_p, _c_int_p, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_void_p]),
    "dib_det_candidates": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_float] * 9 +
                           [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "dib_box_pool": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, _c_int_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                    ctypes.c_void_p]),
    "dib_box_labels": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, _c_int_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_void_p]),
    "dib_coco_box_iou": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_void_p, ctypes.c_void_p]),
    "dib_bias_act_nhwc": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_void_p]),
    "dib_bias_act_mask_nhwc": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int,
                                              ctypes.c_void_p, ctypes.c_void_p]),
    "dib_relu_mask_backward": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p]),
    "dib_add_relu_mask": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p]),
    "dib_scatter_add_nhwc": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "dib_fpn_topdown_merge_nhwc": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "dib_stem_pool_forward": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "dib_stem_pool_backward": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_void_p]),
    "dib_fold_bn_multi": (ctypes.c_int, [_c_void_pp, _c_void_pp, _c_void_pp, _c_void_pp, _c_void_pp, _c_int_p, _c_int_p, ctypes.c_int,
                                         ctypes.c_float, _c_void_pp, _c_void_pp, _c_void_pp, ctypes.c_void_p]),
    "dib_scale_rows_multi": (ctypes.c_int, [_c_void_pp, _c_void_pp, _c_int_p, _c_int_p, ctypes.c_int, _c_void_pp, ctypes.c_void_p]),
    "dib_post_ops": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_double, ctypes.c_ulonglong, ctypes.c_double, ctypes.c_void_p]),
    "dib_jpeg_roundtrip": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), ctypes.c_void_p]),
    # test hook, not part of the drop-in boundary
    "dib_sparse_blur_generic": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                               ctypes.c_void_p]),
}

EXPORTS = tuple(k for k in _SIGNATURES if k != "dib_sparse_blur_generic")


class DibError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("libdib_hip: %s (code %d)" % (text, code))
        self.code = code


class DibStepTimeout(DibError):
    """DIB_ETIMEOUT: an EARLIER blur step on this device gave up waiting inside its launch (include/dib.h, "Device status"); its
    images are incomplete, nothing was launched by the call that reports it, and the device is on two launches per step from here."""


def lib():
    """Loads libdib_hip.so once.  Raises if it has not been built (python __graft_entry__.py)."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise ImportError("%s not found: build it with `make -C detectinblur_amd/csrc` "
                              "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        if l.dib_abi_version() != 7:
            raise ImportError("libdib_hip.so ABI version mismatch")
        _lib = l
    return _lib


def check(code):
    if code != 0:
        text = lib().dib_last_error().decode("utf-8", "replace")
        raise (DibStepTimeout if code == DIB_ETIMEOUT else DibError)(code, text)


_raw_stream = None


def stream_of(tensor):
    """hipStream_t (as an int) of torch's current stream on the tensor's device.  torch's raw accessor is ~30x cheaper than building
    a torch.cuda.Stream object (8 us of host time per call: measured in round 4), which is most of a small launch."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _raw_stream:
        idx = tensor.device.index
        return _raw_stream(idx if idx is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(tensor.device).cuda_stream


_INT_ARRAYS, _PTR_ARRAYS = {}, {}      # ctypes array TYPES by length (building the type is most of a small array's cost)


def int_array(values):
    n = len(values)
    t = _INT_ARRAYS.get(n)
    if t is None:
        t = _INT_ARRAYS[n] = ctypes.c_int * n
    return t(*values)


def ptr_array(values):
    n = len(values)
    t = _PTR_ARRAYS.get(n)
    if t is None:
        t = _PTR_ARRAYS[n] = ctypes.c_void_p * n
    return t(*values)
