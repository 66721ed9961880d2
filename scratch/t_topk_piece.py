"""RPN proposal filter (rpn._filter, 800 x 1333) against the piece size of the split top-k (detector_ops.TOPK_PIECE)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from detectinblur_amd.models import detector_ops as ops
from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
torch.manual_seed(0)
rpn = fasterrcnn_resnet50_fpn(pretrained=False, pretrained_backbone=False, num_classes=91).rpn
counts = [201600, 50400, 12600, 3150, 819]
A = sum(counts)
def timeit(fn, reps=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for N, training in ((1, False), (8, True)):
    rpn.train(training)
    g = torch.Generator().manual_seed(N)
    obj = torch.randn(N, A, generator=g).cuda()
    xy = torch.rand(N, A, 2, generator=g) * torch.tensor([1300.0, 780.0])
    props = torch.cat((xy, xy + torch.rand(N, A, 2, generator=g) * 200), dim=2).cuda()
    sizes = torch.tensor([[1333.0, 800.0]] * N).cuda()
    ks = [min(rpn._n(rpn._pre), n) for n in counts]
    K = max(ks)
    ref = None
    for piece in (1 << 30, 32768, 16384, 12288, 8192):
        ops.TOPK_PIECE = piece
        out = ops.topk_levels_split_hip(obj, counts, ks, K, props, sizes, 1e-3)
        if ref is None: ref = out
        same = all(torch.equal(a, b) for a, b in zip(out, ref))
        t_sel = timeit(lambda: ops.topk_levels_split_hip(obj, counts, ks, K, props, sizes, 1e-3))
        t_all = timeit(lambda: rpn._filter(props, obj.reshape(-1, 1), sizes, counts))
        print("N=%d piece %10d: select %.1f us, whole filter %.1f us (event-timed loops), identical to one piece per level: %s" % (N, piece, t_sel, t_all, same), flush=True)
