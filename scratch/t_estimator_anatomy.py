"""The blur estimator (ResNet-18) on one 3 x 800 x 1312 crop: kernel time by group (convolutions vs batch-norm / ReLU / adds)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch import nn
from torch.profiler import ProfilerActivity, profile
from detectinblur_amd.models.blur_estimator import resnet18
est = resnet18(); est.fc = nn.Linear(512, 4); est = est.cuda().eval()
x = torch.rand(1, 3, 800, 1312, device="cuda")
with torch.no_grad():
    for fmt in (torch.contiguous_format, torch.channels_last):
        xx = x.contiguous(memory_format=fmt)
        m = est.to(memory_format=fmt)
        for _ in range(3): m(xx)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): m(xx)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 20 * 1e3
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(5): m(xx)
            torch.cuda.synchronize()
        tot = {}
        for ev in prof.events():
            if ev.device_type != torch.autograd.DeviceType.CUDA: continue
            n = ev.name
            g = "conv/gemm" if any(t in n.lower() for t in ("igemm", "conv", "cijk", "winograd", "sp3", "gemm", "im2d")) else ("batchnorm" if "atch" in n or "bn" in n.lower() else "elementwise/other")
            tot.setdefault(g, [0.0, 0]); tot[g][0] += ev.device_time * 1e-3 / 5; tot[g][1] += 1 / 5
        print("%s: eager wall %.2f ms; kernels: %s" % (fmt, wall, {k: ("%.3f ms" % v[0], "%d launches" % v[1]) for k, v in tot.items()}), flush=True)
