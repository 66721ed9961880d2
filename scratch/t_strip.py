"""Narrow shape: tiles per workgroup (vertical strips, pipelined) -- time and bit-identity, one process."""
import sys, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from detectinblur_amd import blur_ops, _lib
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, torch.device("cuda", 0))
tables = blur_ops.compact_psfs(psfs, normalize=True)
idx = sorted(range(8), key=lambda k: -dicts[k]["psf_taps"])
ordered = [images[k] for k in idx]
l = _lib.lib(); l.dib_debug_set_strip.argtypes = [ctypes.c_int]; l.dib_debug_set_shape.argtypes = [ctypes.c_int]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
l.dib_debug_set_shape(1)
ref = blur_ops.sparse_blur(list(ordered), idx, tables, 0)
l.dib_debug_set_shape(0)
for rep in range(2):
  for strip in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 5, 7, 9, 13]:
    l.dib_debug_set_strip(strip)
    for mode in (0, 2):
        outs = blur_ops.sparse_blur(list(ordered), idx, tables, mode)
        if mode == 0:
            same = all(torch.equal(a, b) for a, b in zip(ref, outs))
        for _ in range(300): blur_ops.sparse_blur(list(ordered), idx, tables, mode)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(100): blur_ops.sparse_blur(list(ordered), idx, tables, mode)
        e1.record(); e1.synchronize()
        print("rep %d strip %2d mode %d: %.2f us   identical to the 256-wide shape: %s" % (rep, strip, mode, e0.elapsed_time(e1) * 10, same), flush=True)
