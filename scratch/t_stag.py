import sys, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from detectinblur_amd import blur_ops, _lib
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, torch.device("cuda", 0))
tables = blur_ops.compact_psfs(psfs, normalize=True)
idx = sorted(range(8), key=lambda k: -dicts[k]["psf_taps"])
ordered = [images[k] for k in idx]
l = _lib.lib(); l.dib_debug_set_stagger.argtypes = [ctypes.c_int]; l.dib_debug_set_waves.argtypes = [ctypes.c_int]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for waves in (4, 8):
  l.dib_debug_set_waves(waves)
  for which in (1, 0):
    for units in (0, 1, 2, 3, 4, 6):     # s_sleep 64 = 4096 cycles ~ 1.8 us
        stag = (units << 1) | which
        l.dib_debug_set_stagger(stag)
        for _ in range(300): blur_ops.sparse_blur(list(ordered), idx, tables, 0)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(100): blur_ops.sparse_blur(list(ordered), idx, tables, 0)
        e1.record(); e1.synchronize()
        print("waves %d stagger by %s, %d x 1.8us steps: %.2f us" % (waves, "m&3" if which else "(m>>5)&3", units, e0.elapsed_time(e1) * 10), flush=True)
