"""The post-blur corruption chain on one 3 x 800 x 1333 fp16 image: the fused HIP passes vs the stock torch ops on the GPU."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from detectinblur_amd import blur_ops, transforms as T
from detectinblur_amd.models import blur_functions as BF
from detectinblur_amd.models.jpeg import DiffJPEG
dev = torch.device("cuda", 0)
x = torch.rand(3, 800, 1333, device=dev).half()
F = torch.nn.functional


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def eager_noise_block():
    o = torch.clamp(x + (torch.randn_like(x) * 0.05), 0, 1)
    o = F.interpolate(o.unsqueeze(0), scale_factor=(0.8, 0.8), mode="nearest").squeeze()
    return F.interpolate(o.unsqueeze(0), size=(800, 1333), mode="nearest").squeeze()


nb = 2 * x.numel() * 2
t_f = timeit(lambda: blur_ops.post_ops(x, 0.0025, 0.8))
t_e = timeit(eager_noise_block)
print("noise + clamp + block: fused %.1f us (%.0f GB/s of %d B read + written), eager torch %.1f us (7 launches)" % (t_f, nb / t_f / 1e3, nb, t_e))
t_f = timeit(lambda: blur_ops.post_ops(x, None, 0.8))
print("block only:            fused %.1f us (%.0f GB/s)" % (t_f, nb / t_f / 1e3))
m = DiffJPEG(height=100, width=100, differentiable=False, quality=10).to(dev)
m.setQuality(60)
f = np.float32(m.factor)
ly, lc = m.luma.cpu().numpy() * f, m.chroma.cpu().numpy() * f
t_f = timeit(lambda: blur_ops.jpeg_roundtrip(x, ly, lc))


def eager_jpeg():
    p = F.pad(x.unsqueeze(0), (5, 6, 8, 8), mode="reflect")
    m.setRes(p.shape[2], p.shape[3])
    return m(p.float())[:, :, 8:-8, 5:-6].half()


t_e = timeit(eager_jpeg, 20)
print("JPEG round trip:       fused %.1f us (one launch), module-by-module torch %.1f us (~30 launches)" % (t_f, t_e))
