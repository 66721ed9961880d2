"""Steps of the BASELINE batch (one launch each: blur_step_f16_kernel = tap compaction + blur) for the PMC passes:
python scratch/prof_step_r5.py N [cold]      warm = one resident batch re-blurred; cold = 6 input batches visited round-robin."""
import sys
sys.path.insert(0, '.')
import torch
import bench
from detectinblur_amd.models import blur_functions as BF
dev = torch.device("cuda", 0)
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
cold = len(sys.argv) > 2 and sys.argv[2] == "cold"
sets = [images]
if cold:
    sets += [[torch.rand(3, 800, 1333, generator=torch.Generator().manual_seed(977 * s + i)).half().to(dev) for i in range(8)] for s in range(1, 6)]
ring = [None] * len(sets)
for k in range(n):
    j = k % len(sets)
    ring[j] = None
    batch = list(sets[j])
    BF.blur_image_list(batch, dicts, psfs, psfs_complete=True)
    ring[j] = batch
torch.cuda.synchronize()
