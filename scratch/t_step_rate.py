"""Steps per second of the eager blur step (what the soaks run): BASELINE batch and the ragged one, a few thousand steps each, wall time per step.
    DIB_HIP_LIB=... python scratch/t_step_rate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from detectinblur_amd.models import blur_functions as BF
dev = torch.device("cuda", 0)
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev)
ragged = [torch.rand(3, h, w, generator=torch.Generator().manual_seed(31 + i)).half().to(dev) for i, (h, w) in enumerate(bench.COCO_NATIVE_SIZES)]
for name, imgs in (("baseline", images), ("ragged", ragged), ("baseline", images)):
    for _ in range(200):
        BF.blur_image_list(list(imgs), dicts, psfs, psfs_complete=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 3000
    for _ in range(n):
        BF.blur_image_list(list(imgs), dicts, psfs, psfs_complete=True)
    torch.cuda.synchronize()
    print("%s %-9s %.1f us per step" % (os.environ.get("DIB_HIP_LIB", "product"), name, (time.perf_counter() - t0) / n * 1e6), flush=True)
