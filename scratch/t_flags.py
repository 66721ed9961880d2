import sys, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from detectinblur_amd import blur_ops, _lib
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, torch.device("cuda", 0))
tables = blur_ops.compact_psfs(psfs, normalize=True)
idx = sorted(range(8), key=lambda k: -dicts[k]["psf_taps"])
ordered = [images[k] for k in idx]
l = _lib.lib(); l.dib_debug_set_flags.argtypes = [ctypes.c_int]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for mode in (0,):
    for flags in (0, 7, 8, 16):
        l.dib_debug_set_flags(flags)
        for _ in range(300): blur_ops.sparse_blur(list(ordered), idx, tables, mode)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(100): blur_ops.sparse_blur(list(ordered), idx, tables, mode)
        e1.record(); e1.synchronize()
        print("mode %d flags %d (1=no loads 2=no stores 4=no taps): %.2f us" % (mode, flags, e0.elapsed_time(e1) * 10))
l.dib_debug_set_flags(0)
