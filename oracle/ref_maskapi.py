"""ctypes loader for oracle/_ref/libmaskapi.so -- the reference's own cocoapi/common/maskApi.c compiled by
oracle/Makefile.  TEST INFRASTRUCTURE ONLY (tests/ and oracle/gen_goldens.py)."""
import ctypes
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libmaskapi.so")


def available():
    return os.path.isfile(_PATH)


def bb_iou(dt, gt, iscrowd=None):
    """maskApi.c bbIou: dt [m, 4], gt [n, 4] float64 xywh -> float64 [m, n] (pycocotools' layout)."""
    lib = ctypes.CDLL(_PATH)
    dt = np.ascontiguousarray(dt, dtype=np.float64).reshape(-1, 4)
    gt = np.ascontiguousarray(gt, dtype=np.float64).reshape(-1, 4)
    m, n = dt.shape[0], gt.shape[0]
    out = np.zeros((n, m), dtype=np.float64)
    crowd = None if iscrowd is None else np.ascontiguousarray(iscrowd, dtype=np.uint8)
    lib.bbIou.restype = None
    lib.bbIou.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ulong, ctypes.c_ulong, ctypes.c_void_p, ctypes.c_void_p]
    lib.bbIou(dt.ctypes.data, gt.ctypes.data, m, n, crowd.ctypes.data if crowd is not None else None, out.ctypes.data)
    return out.T


class _RLE(ctypes.Structure):
    _fields_ = [("h", ctypes.c_ulong), ("w", ctypes.c_ulong), ("m", ctypes.c_ulong), ("cnts", ctypes.POINTER(ctypes.c_uint))]


def poly_mask(xy, h, w):
    """maskApi.c rleFrPoly (:162-218) + rleDecode (:43-47) of ONE polygon (x0, y0, x1, y1, ...) -> uint8 [h, w] (the reference
    decodes into a column-major buffer: pycocotools hands it to numpy as a Fortran-ordered [h, w] array)."""
    lib = ctypes.CDLL(_PATH)
    xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1)
    k = int(len(xy) / 2)                                  # pycocotools _mask.pyx:266
    R = _RLE()
    lib.rleFrPoly.restype = None
    lib.rleFrPoly.argtypes = [ctypes.POINTER(_RLE), ctypes.c_void_p, ctypes.c_ulong, ctypes.c_ulong, ctypes.c_ulong]
    lib.rleFrPoly(ctypes.byref(R), xy.ctypes.data, k, h, w)
    buf = np.zeros(h * w, dtype=np.uint8)
    lib.rleDecode.restype = None
    lib.rleDecode.argtypes = [ctypes.POINTER(_RLE), ctypes.c_void_p, ctypes.c_ulong]
    lib.rleDecode(ctypes.byref(R), buf.ctypes.data, 1)
    runs = np.array([R.cnts[i] for i in range(R.m)], dtype=np.uint32)
    lib.rleFree.restype = None
    lib.rleFree.argtypes = [ctypes.POINTER(_RLE)]
    lib.rleFree(ctypes.byref(R))
    return buf.reshape(w, h).T.copy(), runs
