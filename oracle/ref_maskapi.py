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
