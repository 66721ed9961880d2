"""Per-direction timing of the trunk's 1x1 stride-1 convolutions (b=8, channels-last fp32): MIOpen forward / backward-data /
backward-weights vs the same contractions as hipBLASLt GEMMs on the NHWC view."""
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


tot = {}
cb = torch.ops.aten.convolution_backward
for cin, cout, H, W, mult in shapes:
    x = torch.randn(8, cin, H, W, device=dev).to(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 1, 1, device=dev) * 0.05
    g = torch.randn(8, cout, H, W, device=dev).to(memory_format=torch.channels_last)
    x2, g2, w2 = x.permute(0, 2, 3, 1).reshape(-1, cin), g.permute(0, 2, 3, 1).reshape(-1, cout), w.view(cout, cin)
    r = {"fwd conv": timeit(lambda: F.conv2d(x, w)), "fwd gemm": timeit(lambda: torch.mm(x2, w2.t())),
         "bwd conv": timeit(lambda: cb(g, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])),
         "bwd gemm": timeit(lambda: torch.mm(g2, w2)),
         "wrw conv": timeit(lambda: cb(g, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])),
         "wrw gemm": timeit(lambda: torch.mm(g2.t(), x2))}
    for k, v in r.items():
        tot[k] = tot.get(k, 0.0) + v * mult
    print("%4d->%4d %3dx%3d x%d  " % (cin, cout, H, W, mult) + "  ".join("%s %.3f" % kv for kv in r.items()), flush=True)
print("per step: " + "  ".join("%s %.2f" % kv for kv in tot.items()))
