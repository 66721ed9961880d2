import sys, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from detectinblur_amd import blur_ops, _lib
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, torch.device("cuda", 0))
tables = blur_ops.compact_psfs(psfs, normalize=True)
idx = sorted(range(8), key=lambda k: -dicts[k]["psf_taps"])
ordered = [images[k] for k in idx]
l = _lib.lib(); l.dib_debug_set_flags.argtypes = [ctypes.c_int]; l.dib_debug_set_waves.argtypes = [ctypes.c_int]; l.dib_debug_set_waves(2)
flags = int(sys.argv[1]); mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
l.dib_debug_set_flags(flags)
for _ in range(400): blur_ops.sparse_blur(list(ordered), idx, tables, mode)
torch.cuda.synchronize()
