"""ATen's sum over (0, 2, 3) of channels-last fp32 gradients at the FPN / RPN sizes of the train step (bias gradients)."""
import torch
dev = torch.device("cuda", 0)
def t(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for (H, W) in ((200, 336), (100, 168), (50, 84), (25, 42), (13, 21)):
    g = torch.randn(8, 256, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    a = t(lambda: g.sum(dim=(0, 2, 3)))
    b = t(lambda: g.permute(0, 2, 3, 1).reshape(-1, 256).sum(0))
    print("8 x 256 x %d x %d (%.0f MB): sum(0,2,3) %.1f us = %.2f TB/s; [M,C].sum(0) %.1f us" % (H, W, g.numel() * 4 / 1e6, a, g.numel() * 4 / a / 1e6, b), flush=True)
