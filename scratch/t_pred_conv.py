"""The RPN predictor (1x1, 256 -> 16 channels, batch 1) per pyramid level: MIOpen's conv2d vs a GEMM on the NHWC view -- time and
run-to-run bit reproducibility (20 runs on the same input)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import detectinblur_amd  # noqa: F401
torch.manual_seed(0)
w = torch.randn(16, 256, 1, 1, device="cuda") * 0.01
b = torch.randn(16, device="cuda")


def bench(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


with torch.no_grad():
    for N in (1, 8):
        for (H, W) in ((200, 336), (100, 168), (50, 84), (25, 42), (13, 21)):
            x = torch.randn(N, 256, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
            conv = lambda: F.conv2d(x, w, b)
            lin = lambda: F.linear(x.permute(0, 2, 3, 1), w.view(16, 256), b).permute(0, 3, 1, 2)
            mm = lambda: torch.addmm(b, x.permute(0, 2, 3, 1).reshape(-1, 256), w.view(16, 256).t())
            res = {}
            for name, fn in (("conv2d", conv), ("linear", lin), ("addmm", mm)):
                ref = fn().clone()
                nd = sum(0 if torch.equal(fn(), ref) else 1 for _ in range(20))
                res[name] = (bench(fn), nd)
            d = float((conv() - lin()).abs().max())
            print("N=%d %3dx%3d  " % (N, H, W) + "  ".join("%s %.1f us (%d/20 differ)" % (k, v[0], v[1]) for k, v in res.items()) + "  |conv-linear| %.2g" % d)
