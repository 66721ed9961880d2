"""A/B of the tiled blur's tile shapes inside ONE process (boxes differ by several per cent):
0 = 128 x 32 "quad" tiles (8-byte LDS elements, default), 1 = 256 x 32 tiles."""
import sys, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from detectinblur_amd import blur_ops, _lib
dev = torch.device("cuda", 0)
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev)
tables = blur_ops.compact_psfs(psfs, normalize=True)
idx = sorted(range(8), key=lambda k: -dicts[k]["psf_taps"])
ordered = [images[k] for k in idx]
sets = [ordered] + [[torch.rand(3, 800, 1333, generator=torch.Generator().manual_seed(977 * s + i)).half().to(dev) for i in range(8)] for s in range(1, 6)]
l = _lib.lib(); l.dib_debug_set_shape.argtypes = [ctypes.c_int]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timeit(fn, reps=100):
    for k in range(300): fn(k)
    torch.cuda.synchronize(); e0.record()
    for k in range(reps): fn(k)
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) * 1000 / reps
ring = [None] * 6
def cold(k, mode):
    j = k % 6; ring[j] = None; ring[j] = blur_ops.sparse_blur(list(sets[j]), idx, tables, mode)
for rep in range(3):
    for shape in (1, 0):
        l.dib_debug_set_shape(shape)
        w0 = timeit(lambda k: blur_ops.sparse_blur(list(ordered), idx, tables, 0))
        w2 = timeit(lambda k: blur_ops.sparse_blur(list(ordered), idx, tables, 2))
        c0 = timeit(lambda k: cold(k, 0), 102)
        print("rep %d shape %d: bit-exact warm %.2f us  cold %.2f us   fma16 warm %.2f us" % (rep, shape, w0, c0, w2), flush=True)
