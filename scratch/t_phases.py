"""Host-side time per phase of one training step (perf_counter, no added syncs) + launches per phase
(counted through torch's profiler-free hook: we count aten ops dispatched via TorchDispatchMode)."""
import sys, time, collections
sys.path.insert(0, '.')
import torch, bench
from detectinblur_amd import utils
from detectinblur_amd.models import blur_functions as BF, rpn as RPN, roi_heads as RH, net_transforms as NT, backbone as BB, generalized_rcnn as GR, detector_ops as DO
from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
acc = collections.defaultdict(float); stack = []
def wrap(cls, name, label=None):
    fn = getattr(cls, name); label = label or (cls.__name__ + "." + name)
    def w(*a, **k):
        t0 = time.perf_counter(); stack.append(0.0)
        try: return fn(*a, **k)
        finally:
            dt = time.perf_counter() - t0; child = stack.pop()
            acc[label] += dt - child
            if stack: stack[-1] += dt
    setattr(cls, name, w)
for c, n in ((RPN.RegionProposalNetwork, "filter_proposals"), (RPN.RegionProposalNetwork, "assign_targets"), (RPN.RegionProposalNetwork, "compute_loss"),
             (RPN.RegionProposalNetwork, "forward"), (RPN.AnchorGenerator, "forward"), (RPN.RPNHead, "forward"),
             (RH.RoIHeads, "select_training_samples"), (RH.RoIHeads, "forward"), (DO.MultiScaleRoIAlign, "forward"),
             (NT.GeneralizedRCNNTransform, "forward"), (BB.BackboneWithFPN, "forward"), (GR.GeneralizedRCNN, "forward")):
    wrap(c, n)
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev)
H, W, BATCH = bench.H, bench.W, bench.BATCH
torch.manual_seed(1337)
model = fasterrcnn_resnet50_fpn(num_classes=91, pretrained=False, pretrained_backbone=False).to(dev)
opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.0004, momentum=0.9, weight_decay=1e-4)
g = torch.Generator().manual_seed(4242)
targets = []
for _ in range(BATCH):
    x1 = torch.rand(8, generator=g) * (W - 34); y1 = torch.rand(8, generator=g) * (H - 34)
    w = 32 + torch.rand(8, generator=g) * 368; h = 32 + torch.rand(8, generator=g) * 368
    boxes = torch.stack([x1, y1, torch.clamp(x1 + w, max=W), torch.clamp(y1 + h, max=H)], 1)
    targets.append({"boxes": boxes.to(dev), "labels": torch.randint(1, 91, (8,), generator=g).to(dev)})
means, stds = utils.get_norm_params(dicts, False)
model.train()
def step():
    t0 = time.perf_counter()
    batch = list(images)
    tg = [{k: v.clone() for k, v in t.items()} for t in targets]
    BF.blur_image_list(batch, dicts, psfs)
    tg = utils.expand_targets(tg, dicts, psfs, batch)
    batch = [b.float() for b in batch]
    t1 = time.perf_counter(); acc["blur+expand+float"] += t1 - t0
    losses = sum(model(batch, tg, newMeans=means, newSTDs=stds).values())
    t2 = time.perf_counter()
    opt.zero_grad(); losses.backward()
    t3 = time.perf_counter(); acc["backward (host issue)"] += t3 - t2
    opt.step()
    acc["optimizer"] += time.perf_counter() - t3
    return losses
for _ in range(3): step()
torch.cuda.synchronize(); acc.clear()
N = 6; t0 = time.perf_counter()
for _ in range(N): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host %.1f ms/step, with final sync %.1f ms/step" % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-45s %7.2f ms/step (self)" % (k, v / N * 1e3))
