"""dib_topk_levels vs the per-level torch.topk chain at the RPN's sizes (800 x 1333): wall time per call."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from detectinblur_amd.models import detector_ops as ops
from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
torch.manual_seed(0)
rpn = fasterrcnn_resnet50_fpn(pretrained=False, pretrained_backbone=False, num_classes=91).rpn
counts = [201600, 50400, 12600, 3150, 819]
A = sum(counts)
def timeit(name, fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%-60s host %.0f us, wall %.0f us per call" % (name, (t1 - t0) / reps * 1e6, (t2 - t0) / reps * 1e6), flush=True)
for N, training in ((8, True), (1, False)):
    rpn.train(training)
    g = torch.Generator().manual_seed(N)
    obj = torch.randn(N, A, generator=g).cuda()
    xy = torch.rand(N, A, 2, generator=g) * torch.tensor([1300.0, 780.0])
    props = torch.cat((xy, xy + torch.rand(N, A, 2, generator=g) * 200), dim=2).cuda()
    sizes = torch.tensor([[1333.0, 800.0]] * N).cuda()
    K = max(min(rpn._n(rpn._pre), n) for n in counts)
    ks = [min(rpn._n(rpn._pre), n) for n in counts]
    timeit("N=%d K=%d: dib_topk_levels (select + gather + clip)" % (N, K), lambda: ops.topk_levels_hip(obj, counts, ks, K, props, sizes, 1e-3))
    timeit("N=%d K=%d: tensor form (5 x topk + gather + clip)" % (N, K), lambda: rpn._select_levels(props, obj, sizes, counts, K))
    for flag in (True, False):
        ops.HIP_BOXES = flag
        timeit("N=%d: whole _filter, hip_boxes=%d" % (N, flag), lambda: rpn._filter(props, obj.reshape(-1, 1), sizes, counts))
    ops.HIP_BOXES = True
