"""Like t_ab.py, plus the cold figure (6 input batches + 6 live output blocks visited round-robin): DIB_LIB=<path> python scratch/t_ab_cold.py"""
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
sets = [ordered] + [[torch.rand(3, 800, 1333, generator=torch.Generator().manual_seed(977 * s + i)).half().to(dev) for i in range(8)] for s in range(1, 6)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ring = [None] * 6
def cold(k):
    j = k % 6; ring[j] = None; ring[j] = blur_ops.sparse_blur(list(sets[j]), idx, tables, 0)
def warm(k): blur_ops.sparse_blur(list(ordered), idx, tables, 0)
def timeit(fn, reps):
    for k in range(300): fn(k)
    torch.cuda.synchronize(); e0.record()
    for k in range(reps): fn(k)
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) * 1000 / reps
w = sorted(timeit(warm, 200) for _ in range(5)); c = sorted(timeit(cold, 204) for _ in range(5))
print(os.environ.get("DIB_LIB", "default"), "warm median %.2f (%.2f..%.2f)  cold median %.2f (%.2f..%.2f) us" % (w[2], w[0], w[4], c[2], c[0], c[4]))
