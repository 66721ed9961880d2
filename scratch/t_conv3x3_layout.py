"""The big 3x3 stride-1 convolutions of the step (FPN output convs, RPN head conv: 256 -> 256, and the trunk's conv2s) in
channels-last (MIOpen implicit GEMM) vs planar NCHW (where MIOpen's fp32 Winograd kernels apply): forward, backward-data,
backward-weights, ms per call, b = 8."""
import torch
import torch.nn.functional as F
dev = torch.device("cuda", 0)
cb = torch.ops.aten.convolution_backward
shapes = [(256, 256, 200, 336), (256, 256, 100, 168), (64, 64, 200, 336), (128, 128, 100, 168), (256, 256, 50, 84), (512, 512, 25, 42)]


def timeit(fn, n=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n


for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    print("cudnn.benchmark (MIOpen Find) =", bench)
    for cin, cout, H, W in shapes:
        row = []
        for fmt in (torch.channels_last, torch.contiguous_format):
            x = torch.randn(8, cin, H, W, device=dev).contiguous(memory_format=fmt)
            w = (torch.randn(cout, cin, 3, 3, device=dev) * 0.02).contiguous(memory_format=fmt)
            g = torch.randn(8, cout, H, W, device=dev).contiguous(memory_format=fmt)
            f = timeit(lambda: F.conv2d(x, w, padding=1))
            b = timeit(lambda: cb(g, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False]))
            ww = timeit(lambda: cb(g, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))
            row.append("%s fwd %.3f bwd %.3f wrw %.3f" % ("NHWC" if fmt == torch.channels_last else "NCHW", f, b, ww))
        print("  %3d->%3d %3dx%3d  " % (cin, cout, H, W) + "  |  ".join(row), flush=True)
