"""Bias gradient of a few-channel convolution output (RPN predictors: 15 channels at 200 x 336, b = 8): at::sum over (0, 2, 3)
for channels-last and planar gradients against a two-stage sum."""
import torch, time
dev = torch.device("cuda", 0)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for C in (3, 12, 15, 256):
    for (H, W) in ((200, 336), (100, 168)):
        g_cl = torch.randn(8, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        g_pl = g_cl.contiguous()
        a = t(lambda: g_cl.sum((0, 2, 3)))
        b = t(lambda: g_pl.sum((0, 2, 3)))
        c = t(lambda: g_cl.sum(3).sum((0, 2)))
        d = t(lambda: g_cl.permute(0, 2, 3, 1).reshape(-1, C).sum(0))
        ones = torch.ones(1, 8 * H * W, device=dev)
        e = t(lambda: ones @ g_cl.permute(0, 2, 3, 1).reshape(-1, C))
        ref = g_cl.double().sum((0, 2, 3))
        err = float((g_cl.sum(3).sum((0, 2)).double() - ref).abs().max())
        print("C=%3d %dx%d: sum(0,2,3) channels-last %7.1f us, planar %7.1f us; sum(3).sum(0,2) %7.1f us; [M,C].sum(0) %7.1f us; ones@G %7.1f us  (two-stage err %.2e)"
              % (C, H, W, a, b, c, d, e, err), flush=True)
