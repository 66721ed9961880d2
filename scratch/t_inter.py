import sys, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from detectinblur_amd import blur_ops, _lib
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, torch.device("cuda", 0))
tables = blur_ops.compact_psfs(psfs, normalize=True)
idx = sorted(range(8), key=lambda k: -dicts[k]["psf_taps"])
ordered = [images[k] for k in idx]
l = _lib.lib(); l.dib_debug_set_waves.argtypes = [ctypes.c_int]; l.dib_debug_set_interleave.argtypes = [ctypes.c_int]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ref = None
for waves in (4, 8, 2):
  for inter in (0, 1):
    l.dib_debug_set_waves(waves); l.dib_debug_set_interleave(inter)
    for mode in (0, 2):
        outs = blur_ops.sparse_blur(list(ordered), idx, tables, mode)
        if mode == 0:
            if ref is None: ref = [o.clone() for o in outs]
            same = all(torch.equal(a, b) for a, b in zip(ref, outs))
        for _ in range(300): blur_ops.sparse_blur(list(ordered), idx, tables, mode)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(100): blur_ops.sparse_blur(list(ordered), idx, tables, mode)
        e1.record(); e1.synchronize()
        print("shape %d interleave %d mode %d: %.2f us   identical: %s" % (waves, inter, mode, e0.elapsed_time(e1) * 10, same), flush=True)
