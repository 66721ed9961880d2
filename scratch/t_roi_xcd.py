"""RoIAlign backward: RoIs in image order (workgroup k of image k // 512: every image's gradient lines are hit from all 8 XCDs) vs
interleaved (RoI k belongs to image k % 8: with round-robin dispatch every image is handled by ONE XCD, its lines stay in one L2)."""
import math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from detectinblur_amd.models import detector_ops as ops
g = torch.Generator().manual_seed(0)
N, H, W, K = 8, 800, 1344, 512
feats = [torch.randn(N, 256, H // s, W // s, generator=g).cuda().contiguous(memory_format=torch.channels_last).requires_grad_() for s in (4, 8, 16, 32)]
rois = []
for i in range(N):
    wh = torch.exp(torch.rand(K, 2, generator=g) * 3.5 + 2.5)
    xy = torch.rand(K, 2, generator=g) * torch.tensor([W - 1.0, H - 1.0])
    b = torch.cat((xy - wh / 2, xy + wh / 2), 1).clamp(min=0)
    b[:, 2].clamp_(max=W - 1.0); b[:, 3].clamp_(max=H - 1.0)
    rois.append(torch.cat((torch.full((K, 1), float(i)), b), 1))
by_image = torch.cat(rois).cuda()
interleaved = torch.stack(rois, 1).reshape(-1, 5).cuda()             # row k: image k % 8
scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
def run(r):
    area = (r[:, 3] - r[:, 1]) * (r[:, 4] - r[:, 2])
    lvl = (torch.floor(4 + torch.log2(torch.sqrt(area) / 224) + 1e-6).clamp(2, 5) - 2).to(torch.int32)
    out = ops._RoIAlignNHWC.apply(r, lvl, scales, 7, 2, False, *feats)
    go = torch.ones_like(out)
    for f in feats: f.grad = None
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out.backward(go); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1), [f.grad.clone() for f in feats]
for name, r in (("image order (as the step has it)", by_image), ("interleaved: one XCD per image", interleaved)):
    ts = []
    for _ in range(6):
        t, grads = run(r); ts.append(t)
    print("%-36s backward (zero fills + kernel) %.3f ms" % (name, sum(ts[2:]) / 4), flush=True)
