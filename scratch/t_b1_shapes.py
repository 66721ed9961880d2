"""Inference at batch 1 (the evaluation loop): forward time of the trunk's convolutions by shape -- MIOpen channels-last vs
GEMM on the NHWC view (1x1) vs MIOpen planar (3x3)."""
import torch
import torch.nn.functional as F
dev = torch.device("cuda", 0)


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


one = [(64, 64, 200, 336, 3), (64, 256, 200, 336, 4), (256, 64, 200, 336, 2), (256, 128, 200, 336, 1), (128, 512, 100, 168, 4), (512, 128, 100, 168, 3),
       (512, 256, 100, 168, 1), (256, 1024, 50, 84, 6), (1024, 256, 50, 84, 5), (1024, 512, 50, 84, 1), (512, 2048, 25, 42, 3), (2048, 512, 25, 42, 2),
       (256, 256, 200, 336, 1), (512, 256, 100, 168, 1), (1024, 256, 50, 84, 1), (2048, 256, 25, 42, 1)]
tot = [0, 0]
with torch.no_grad():
    for cin, cout, H, W, mult in one:
        x = torch.randn(1, cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(cout, cin, 1, 1, device=dev) * 0.05
        a = timeit(lambda: F.conv2d(x, w)); b = timeit(lambda: F.linear(x.permute(0, 2, 3, 1), w.view(cout, cin)))
        tot[0] += a * mult; tot[1] += min(a, b) * mult
        print("1x1 %4d->%4d %3dx%3d x%d  conv %.1f us  gemm %.1f us" % (cin, cout, H, W, mult, a, b), flush=True)
    print("1x1 per image: conv %.0f us, best-of %.0f us" % tuple(tot))
    tot = [0, 0]
    for cin, cout, H, W, s, mult in [(64, 64, 200, 336, 1, 3), (128, 128, 200, 336, 2, 1), (128, 128, 100, 168, 1, 3), (256, 256, 100, 168, 2, 1), (256, 256, 50, 84, 1, 7),
                                     (512, 512, 50, 84, 2, 1), (512, 512, 25, 42, 1, 2), (256, 256, 200, 336, 1, 2), (256, 256, 100, 168, 1, 2), (256, 256, 25, 42, 1, 2)]:
        x = torch.randn(1, cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, 3, 3, device=dev) * 0.02).contiguous(memory_format=torch.channels_last)
        xp, wp = x.contiguous(), w.contiguous()
        a = timeit(lambda: F.conv2d(x, w, stride=s, padding=1)); b = timeit(lambda: F.conv2d(xp, wp, stride=s, padding=1))
        tot[0] += a * mult; tot[1] += min(a, b) * mult
        print("3x3 %4d->%4d %3dx%3d s%d x%d  NHWC %.1f us  NCHW %.1f us" % (cin, cout, H, W, s, mult, a, b), flush=True)
    print("3x3 per image: NHWC %.0f us, best-of %.0f us" % tuple(tot))
