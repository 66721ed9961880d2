"""Default kernel: launch time by how many trailing tiles run one tile per workgroup (the rest run two).
   python scratch/t_pair.py"""
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
l = _lib.lib(); l.dib_debug_set_single_tail.argtypes = [ctypes.c_int]; l.dib_debug_set_single_tail.restype = None
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timeit(mode):
    for _ in range(300): blur_ops.sparse_blur(list(ordered), idx, tables, mode)
    torch.cuda.synchronize(); e0.record()
    for _ in range(200): blur_ops.sparse_blur(list(ordered), idx, tables, mode)
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) * 5
l.dib_debug_set_single_tail(1 << 30)
ref = [o.clone() for o in blur_ops.sparse_blur(list(ordered), idx, tables, 0)]
for rep in range(3):
    for tail in (1 << 30, 0, 900, 1700, 2700, 3400, 4200):
        l.dib_debug_set_single_tail(tail)
        out = blur_ops.sparse_blur(list(ordered), idx, tables, 0)
        ok = all(torch.equal(a, b) for a, b in zip(out, ref))
        print("rep %d single tail %10d: bit-exact %.2f us  fma16 %.2f us  identical %s" % (rep, tail, timeit(0), timeit(2), ok), flush=True)
