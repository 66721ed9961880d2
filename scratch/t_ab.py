"""A/B of two builds of libdib_hip.so in one gpurun call: DIB_LIB=<path> python scratch/t_ab.py"""
import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
from detectinblur_amd import _lib
if os.environ.get("DIB_LIB"):
    _lib.LIB_PATH = os.environ["DIB_LIB"]
import bench
from detectinblur_amd import blur_ops
dev = torch.device("cuda", 0)
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev)
tables = blur_ops.compact_psfs(psfs, normalize=True)
idx = sorted(range(8), key=lambda k: -dicts[k]["psf_taps"])
ordered = [images[k] for k in idx]
MODE = int(os.environ.get("DIB_MODE", "0"))   # 0 bit-exact, 2 FMA16
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
res = []
for rep in range(5):
    for _ in range(300): blur_ops.sparse_blur(list(ordered), idx, tables, MODE)
    torch.cuda.synchronize(); e0.record()
    for _ in range(200): blur_ops.sparse_blur(list(ordered), idx, tables, MODE)
    e1.record(); e1.synchronize(); res.append(e0.elapsed_time(e1) * 5)
print(os.environ.get("DIB_LIB", "default"), " ".join("%.2f" % r for r in res), "us; median %.2f" % sorted(res)[2])
