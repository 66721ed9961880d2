"""RoIAlign backward (channels-last, 4 levels, 8 x 512 RoIs at 800 x 1333): does the ORDER of the RoIs matter (atomics landing near
each other in time hit the L2 / Infinity Cache instead of HBM)?  Random order vs sorted by (level, image, y, x)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from detectinblur_amd.models import detector_ops as ops
g = torch.Generator().manual_seed(0)
N, H, W = 8, 800, 1344
feats = [torch.randn(N, 256, H // s, W // s, generator=g).cuda().contiguous(memory_format=torch.channels_last).requires_grad_() for s in (4, 8, 16, 32)]
K = 512
boxes = []
for i in range(N):
    wh = torch.exp(torch.rand(K, 2, generator=g) * 3.5 + 2.5)              # sides 12 .. 400
    xy = torch.rand(K, 2, generator=g) * torch.tensor([W - 1.0, H - 1.0])
    b = torch.cat((xy - wh / 2, xy + wh / 2), 1).clamp(min=0)
    b[:, 2].clamp_(max=W - 1.0); b[:, 3].clamp_(max=H - 1.0)
    boxes.append(b.cuda())
pool = ops.MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
features = {str(i): f for i, f in enumerate(feats)}
def run(bx):
    out = pool(features, bx, [(H, W)] * N)
    go = torch.ones_like(out)
    for f in feats: f.grad = None
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out.backward(go); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
def sort_boxes(bx):
    out = []
    for b in bx:
        area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        lvl = torch.floor(4 + torch.log2(torch.sqrt(area) / 224) + 1e-6).clamp(2, 5)
        cy, cx = (b[:, 1] + b[:, 3]) / 2, (b[:, 0] + b[:, 2]) / 2
        key = lvl * 1e8 + torch.floor(cy / 32) * 1e4 + cx
        out.append(b[torch.argsort(key)])
    return out
for name, bx in (("random order", boxes), ("sorted by level, row band, x", sort_boxes(boxes))):
    ts = [run(bx) for _ in range(6)][2:]
    print("%-32s backward (zero fills + kernel) %.3f ms" % (name, sum(ts) / len(ts)), flush=True)
