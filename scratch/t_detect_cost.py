"""Per-call device + host cost of the box bookkeeping kernels vs the tensor expressions, RPN / RoI sizes of the 800 x 1333 step."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from detectinblur_amd.models import detector_ops as ops
from tests.test_detect_gpu import _boxes
g = torch.Generator().manual_seed(0)
N, A = 2, 242991
anchors = _boxes(g, A, 1333.0, 800.0).cuda()
gts = [_boxes(g, 7, 1333.0, 800.0).cuda(), _boxes(g, 12, 1333.0, 800.0).cuda()]
deltas = (torch.randn(N * A, 4, generator=g) * 0.3).cuda()
matcher = ops.Matcher(0.7, 0.3, True)
coder = ops.BoxCoder((1.0, 1.0, 1.0, 1.0))
gt_cat, offs = ops.cat_boxes(gts)

def timeit(name, fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-44s host %.1f us/call, wall %.1f us/call" % (name, (t1 - t0) / reps * 1e6, (t2 - t0) / reps * 1e6), flush=True)

m = ops.match_boxes_hip(matcher, gt_cat, offs, anchors, True)
timeit("rpn match hip", lambda: ops.match_boxes_hip(matcher, gt_cat, offs, anchors, True))
def t_match():
    gt, valid = ops.pad_boxes(gts)
    return ops.match_batched(matcher, ops.box_iou_batched(gt, anchors), valid)
timeit("rpn match torch (pad + iou + matcher)", t_match)
timeit("rpn encode hip", lambda: ops.encode_matched_hip(coder, gt_cat, offs, m, anchors, True))
gt, valid = ops.pad_boxes(gts)
def t_enc():
    matched = gt.gather(1, m.clamp(min=0)[..., None].expand(-1, -1, 4))
    return coder.encode(matched.reshape(-1, 4), torch.cat([anchors] * N))
timeit("rpn gather + encode torch", t_enc)
timeit("rpn decode hip", lambda: ops.decode_boxes_hip(coder, deltas, anchors))
timeit("rpn decode torch", lambda: coder.decode(deltas, torch.cat([anchors] * N)))
timeit("rpn labels (clamp + 1).float()", lambda: (m.clamp(max=0) + 1).to(torch.float32))
P = 2000
props = torch.stack([_boxes(g, P, 1333.0, 800.0) for _ in range(N)]).cuda()
ok = torch.ones((N, P), dtype=torch.bool).cuda()
labs = [torch.randint(1, 91, (b.shape[0],), generator=g).cuda() for b in gts]
lab_cat = torch.cat(labs)
rm = ops.Matcher(0.5, 0.5, False)
timeit("roi pool hip", lambda: ops.pool_boxes_hip(props, gt_cat, offs, 12))
cands = ops.pool_boxes_hip(props, gt_cat, offs, 12)
timeit("roi match hip", lambda: ops.match_boxes_hip(rm, gt_cat, offs, cands, False))
mm = ops.match_boxes_hip(rm, gt_cat, offs, cands, False)
timeit("roi labels hip", lambda: ops.pool_labels_hip(mm, lab_cat, offs, ok, P))
timeit("cat_boxes", lambda: ops.cat_boxes(gts))
timeit("torch.empty", lambda: torch.empty((N, P), dtype=torch.int64, device="cuda"))
timeit("current_stream().cuda_stream", lambda: torch.cuda.current_stream().cuda_stream)
