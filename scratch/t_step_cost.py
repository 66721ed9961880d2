"""Where a blur step's time goes: host cost of the Python path (tiny images: the GPU is never the limit), and the full-size
step through (a) the two separate calls, (b) dib_blur_step default ordering, (c) PSFS_COMPLETE, (d) the raw C call in a loop
with prebuilt arguments (no list handling, no allocation)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from detectinblur_amd import _lib, blur_ops
from detectinblur_amd.models import blur_functions as BF

dev = torch.device("cuda", 0)
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev)
small = [im[:, :70, :70].contiguous() for im in images]


def loop(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    h = time.perf_counter() - t0
    torch.cuda.synchronize()
    return h / n * 1e6, (time.perf_counter() - t0) / n * 1e6


def two_call(imgs):
    def f():
        b = list(imgs)
        tabs = blur_ops.compact_psfs(psfs, True)
        idx = sorted(range(8), key=lambda k: -dicts[k]["psf_taps"])
        blur_ops.sparse_blur([b[k] for k in idx], idx, tabs)
    return f


def step(imgs, complete):
    def f():
        b = list(imgs)
        BF.blur_image_list(b, dicts, psfs, psfs_complete=complete)
    return f


for name, fn in (("two-call", two_call), ("step default", lambda im: step(im, False)), ("step complete", lambda im: step(im, True))):
    for _ in range(3):
        hs, ws = loop(fn(small), 2000)
    for _ in range(3):
        hf, wf = loop(fn(images), 2000)
    print("%-14s small: host %.1f us wall %.1f us | full: host-enqueue %.1f us wall %.1f us" % (name, hs, ws, hf, wf))

# raw C loop
l = _lib.lib()
idx = sorted(range(8), key=lambda k: -dicts[k]["psf_taps"])
ordered = [images[k] for k in idx]
out = torch.empty((8,) + tuple(images[0].shape), dtype=torch.float16, device=dev)
args = (_lib.ptr_array([p.data_ptr() for p in psfs]), 0, 8, 128, 1, _lib.ptr_array([t.data_ptr() for t in ordered]),
        _lib.ptr_array([out[k].data_ptr() for k in range(8)]), _lib.int_array([3] * 8), _lib.int_array([800] * 8), _lib.int_array([1333] * 8),
        _lib.int_array(idx), 8, 0, 0)
st = torch.cuda.current_stream().cuda_stream
tabs = blur_ops.TapTables(128, 8, dev)
for flags, tb, name in ((0, None, "C default"), (1, None, "C complete"), (0, tabs.buf.data_ptr(), "C caller tables (serial)")):
    f = lambda: l.dib_blur_step(*args, tb, flags, st)
    for _ in range(3):
        h, w = loop(f, 3000)
    print("%-26s host-enqueue %.1f us wall %.1f us" % (name, h, w))
f = lambda: l.dib_sparse_blur(*args[5:13], tabs.buf.data_ptr(), 8, 128, 0, st)
blur_ops.compact_psfs(psfs, True)
l.dib_psf_compact_list(args[0], 0, 8, 128, 1, tabs.buf.data_ptr(), st)
for _ in range(3):
    h, w = loop(f, 3000)
print("%-26s host-enqueue %.1f us wall %.1f us" % ("C blur only", h, w))
