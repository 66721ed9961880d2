import sys, os, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from detectinblur_amd import blur_ops, _lib
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, torch.device("cuda", 0))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
tables = blur_ops.compact_psfs(psfs, normalize=True)
idx = list(range(8))
for _ in range(n):
    tables = blur_ops.compact_psfs(psfs, normalize=True)
    blur_ops.sparse_blur(list(images), idx, tables)
torch.cuda.synchronize()
