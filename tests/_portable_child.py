"""Child process of tests/test_portable_build_gpu.py: a fixed set of blur launches through whichever build of the device library
DIB_HIP_LIB names, printing one SHA-256 per case.  Every hand-written loop of csrc/dib_blur.hip gets a case: the quad shape's
bit-exact / FMA16 / FP32 loops (full and half tiles, standard and large window, both canvases), the vertical-run loop, the
256-wide shape's loop, the step's single launch (its in-launch wait)."""
import ctypes
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from detectinblur_amd import _lib, blur_ops

dev = torch.device("cuda", 0)
rs = np.random.RandomState(11)


def psf(K, n, spread):
    a = np.zeros((K, K), np.float64)
    r = np.clip(rs.randint(-spread, spread + 1, n) + K // 2 - 1, 0, K - 1)
    c = np.clip(rs.randint(-spread, spread + 1, n) + K // 2 - 1, 0, K - 1)
    a[r, c] = rs.random_sample(n) + 0.01
    return torch.from_numpy((a / a.sum()).astype(np.float16)).to(dev)


def images(sizes):
    return [torch.rand(3, h, w, generator=torch.Generator().manual_seed(100 + i)).half().to(dev) for i, (h, w) in enumerate(sizes)]


def digest(outs):
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for o in outs:
        h.update(o.cpu().numpy().tobytes())
    return h.hexdigest()


out = {}
sizes = [(480, 640), (333, 500), (200, 181), (64, 40), (800, 1333)]          # ragged; 1333 and 181 leave half tiles at the right edge
imgs = images(sizes)
idx = list(range(len(sizes)))
p128 = [psf(128, n, s) for n, s in ((40, 6), (90, 14), (7, 2), (300, 30), (56, 9))]
t = blur_ops.compact_psfs(p128, normalize=True, vruns=True)
for name, mode in (("bitexact", _lib.DIB_ACC_BITEXACT), ("fma16", _lib.DIB_ACC_FMA16), ("fp32", _lib.DIB_ACC_FP32), ("fast16", _lib.DIB_ACC_FAST16)):
    out["quad128_" + name] = digest(blur_ops.sparse_blur(list(imgs), idx, t, mode))
uni = images([(256, 384)] * 4)                                               # a uniform batch: the 2-D grid
for name, mode in (("bitexact", _lib.DIB_ACC_BITEXACT), ("fast16", _lib.DIB_ACC_FAST16)):
    out["uniform_" + name] = digest(blur_ops.sparse_blur(list(uni), [0, 1, 2, 3], t, mode))
tl = blur_ops.compact_psfs(p128[:2], normalize=True, large_window=True)     # the large LDS window
for name, mode in (("bitexact", _lib.DIB_ACC_BITEXACT), ("fma16", _lib.DIB_ACC_FMA16)):
    out["large_" + name] = digest(blur_ops.sparse_blur(list(imgs[:2]), [0, 1], tl, mode))
p256 = [psf(256, n, s) for n, s in ((60, 20), (150, 60))]
t256 = blur_ops.compact_psfs(p256, normalize=True)
for name, mode in (("bitexact", _lib.DIB_ACC_BITEXACT), ("fma16", _lib.DIB_ACC_FMA16), ("fp32", _lib.DIB_ACC_FP32)):
    out["quad256_" + name] = digest(blur_ops.sparse_blur(list(imgs[:2]), [0, 1], t256, mode))
l = _lib.lib()
l.dib_debug_set_shape.argtypes = [ctypes.c_int]; l.dib_debug_set_shape.restype = None
l.dib_debug_set_shape(1)                                                     # the 256-wide shape (tap_loop_r8)
for name, mode in (("bitexact", _lib.DIB_ACC_BITEXACT), ("fma16", _lib.DIB_ACC_FMA16)):
    out["wide_" + name] = digest(blur_ops.sparse_blur(list(imgs), idx, t, mode))
l.dib_debug_set_shape(0)
for name, mode in (("bitexact", _lib.DIB_ACC_BITEXACT), ("fma16", _lib.DIB_ACC_FMA16)):          # the step's single launch
    out["step_" + name] = digest(blur_ops.blur_step(list(imgs), idx, p128, normalize=True, acc_mode=mode))
f32 = [x.float() for x in imgs[:3]]                                          # fp32 images: the generic kernel (no hand-written loop)
out["f32_images"] = digest(blur_ops.sparse_blur(f32, [0, 1, 2], t, _lib.DIB_ACC_BITEXACT))
print("DIGESTS " + json.dumps(out))
