"""Fixed cost of a blur launch: kernel time for 8 / 16 / 24 / 32 images of the BASELINE kind in ONE launch."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench as B
from detectinblur_amd import blur_ops
dev = torch.device("cuda", 0)
host = B.make_psfs_host(0)
images, dicts, psfs, _, _ = B.make_workload(0, dev, host)
for mult in (1, 2, 3, 4):
    imgs = [im for _ in range(mult) for im in images]
    ps = [p for _ in range(mult) for p in psfs]
    tabs = blur_ops.compact_psfs(ps, normalize=True)
    taps = [tabs.header(i)[0] for i in range(len(ps))]
    idx = sorted(range(len(imgs)), key=lambda k: -taps[k])
    ordered = [imgs[k] for k in idx]
    for _ in range(20):
        blur_ops.sparse_blur(list(ordered), idx, tabs)
    ms = sorted(B.kernel_time_ms(lambda k: blur_ops.sparse_blur(list(ordered), idx, tabs), 100) for _ in range(5))[2]
    n = len(imgs)
    print("%2d images per launch: %.1f us = %.2f us per image, %.3f of the HBM roofline" % (n, ms * 1e3, ms * 1e3 / n, B.ALGO_BYTES_PER_IMAGE * n / (ms * 1e-3) / 1e9 / 8000), flush=True)
