"""1x1 stride-1 convolutions of the R50-FPN trunk at b=8, 800x1344, channels-last fp32: MIOpen conv2d vs the same
contraction as F.linear on the NHWC view (hipBLASLt), forward + backward (data + weight gradients), ms per call."""
import sys, time
import torch
import torch.nn.functional as F
dev = torch.device("cuda", 0)
shapes = [(64, 64, 200, 336, 3), (64, 256, 200, 336, 3), (256, 64, 200, 336, 2), (256, 128, 200, 336, 1), (128, 512, 100, 168, 4),
          (512, 128, 100, 168, 3), (512, 256, 100, 168, 1), (256, 1024, 50, 84, 6), (1024, 256, 50, 84, 5), (1024, 512, 50, 84, 1),
          (512, 2048, 25, 42, 3), (2048, 512, 25, 42, 2), (256, 256, 200, 336, 1), (512, 256, 100, 168, 1), (1024, 256, 50, 84, 1),
          (2048, 256, 25, 42, 1)]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n


tot = [0.0, 0.0]
for cin, cout, H, W, mult in shapes:
    x = torch.randn(8, cin, H, W, device=dev).to(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(cout, cin, 1, 1, device=dev) * 0.05).requires_grad_(True)
    g = torch.randn(8, cout, H, W, device=dev).to(memory_format=torch.channels_last)

    def conv():
        y = F.conv2d(x, w)
        y.backward(g)
        x.grad = None; w.grad = None

    def lin():
        y = F.linear(x.permute(0, 2, 3, 1), w.view(cout, cin)).permute(0, 3, 1, 2)
        y.backward(g)
        x.grad = None; w.grad = None

    y1 = F.conv2d(x, w); y2 = F.linear(x.permute(0, 2, 3, 1), w.view(cout, cin)).permute(0, 3, 1, 2)
    err = (y1 - y2).abs().max().item()
    a, b = timeit(conv), timeit(lin)
    tot[0] += a * mult; tot[1] += b * mult
    fl = 3 * 2 * 8 * H * W * cin * cout / 1e9
    print("%4d->%4d %3dx%3d x%d  conv2d %.3f ms (%.0f TF)  linear %.3f ms (%.0f TF)  maxdiff %.1e  cl=%s" % (cin, cout, H, W, mult, a, fl / a, b, fl / b, err, y2.is_contiguous(memory_format=torch.channels_last)), flush=True)
print("per step (weighted by block count): conv2d %.2f ms, linear %.2f ms" % tuple(tot))
