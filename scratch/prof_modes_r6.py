"""A few launches of the blur alone on the BASELINE batch in one accumulation mode, for rocprofv3 --pmc passes:
    python3 scratch/prof_modes_r6.py <bitexact|fma16|fast16> [launches]"""
import sys
sys.path.insert(0, '.')
import torch
import bench
from detectinblur_amd import blur_ops
dev = torch.device("cuda", 0)
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev)
mode = {"bitexact": 0, "fma16": 2, "fast16": 3}[sys.argv[1]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
tables = blur_ops.compact_psfs(psfs, normalize=True, vruns=True)
idx = sorted(range(8), key=lambda k: -dicts[k]["psf_taps"])
ordered = [images[k] for k in idx]
for _ in range(n):
    out = blur_ops.sparse_blur(list(ordered), idx, tables, mode)
torch.cuda.synchronize()
