import sys, time
sys.path.insert(0, '.')
import torch, bench
from detectinblur_amd import blur_ops
from detectinblur_amd.models import blur_functions as BF
dev = torch.device("cuda", 0)
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev)
host_pinned = [im.cpu().pin_memory() for im in images]
host_pageable = [im.cpu() for im in images]
psf_host = [p.cpu().pin_memory() for p in psfs]
def step(host, non_blocking):
    pass  # (round 3: the table cache is gone)
    batch = [h.to(dev, non_blocking=non_blocking) for h in host]
    ps = [p.to(dev, non_blocking=non_blocking) for p in psf_host]
    BF.blur_image_list(batch, dicts, ps)
    return batch
for name, host, nb in (("pinned, non_blocking", host_pinned, True), ("pageable (reference engine.py:80)", host_pageable, False)):
    for _ in range(20): step(host, nb)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): step(host, nb)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print("%-36s %.3f ms per batch of 8 -> %.0f images/s (H2D of 51.2 MB fp16 + PSFs + compaction + blur)" % (name, el / 100 * 1e3, 800 / el))
