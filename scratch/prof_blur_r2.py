"""Blur launches of the BASELINE batch for the PMC passes: python scratch/prof_blur_r2.py N [cold]
warm = one resident batch re-blurred; cold = 6 input batches + 6 live output blocks visited round-robin."""
import sys
sys.path.insert(0, '.')
import torch
import bench
from detectinblur_amd import blur_ops
dev = torch.device("cuda", 0)
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
cold = len(sys.argv) > 2 and sys.argv[2] == "cold"
tables = blur_ops.compact_psfs(psfs, normalize=True)
idx = sorted(range(8), key=lambda k: -dicts[k]["psf_taps"])
ordered = [images[k] for k in idx]
sets = [ordered]
if cold:
    sets += [[torch.rand(3, 800, 1333, generator=torch.Generator().manual_seed(977 * s + i)).half().to(dev) for i in range(8)] for s in range(1, 6)]
ring = [None] * len(sets)
for k in range(n):
    j = k % len(sets)
    ring[j] = None
    ring[j] = blur_ops.sparse_blur(list(sets[j]), idx, tables)
torch.cuda.synchronize()
