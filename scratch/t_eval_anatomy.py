"""GPU anatomy of the per-image evaluation loop (configs[4], batch 1, in-memory batches: no loader): wall per image vs the sum of
kernel time (torch.profiler), and the kernels by group (trunk graph, estimator graph, RoI heads, blur / transform, copies)."""
import contextlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch import nn
from detectinblur_amd import engine, utils
from detectinblur_amd.coco_utils import SyntheticCocoDetection
from detectinblur_amd.models.blur_estimator import resnet18
from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
from detectinblur_amd.train import get_transform
from torch.profiler import ProfilerActivity, profile

dev = torch.device("cuda", 0)
torch.manual_seed(1337)
ens = [fasterrcnn_resnet50_fpn(num_classes=91, pretrained=False, pretrained_backbone=False).to(dev).eval() for _ in range(4)]
est = resnet18(); est.fc = nn.Linear(512, 4); est = est.to(dev).eval()
with contextlib.redirect_stdout(sys.stderr):
    tf = get_transform(False, blur=True, blur_type=0.001, blur_ratio=1, blur_exposure=0.5)
N = 40
ds = SyntheticCocoDetection(num_images=N, size=(800, 1333), transforms=tf)


class L(list):
    dataset = ds


batches = L(utils.collate_fn([ds[i]]) for i in range(N))
kw = dict(blurring_images=True, gpu_blur=True, expand_target_boxes=True, use_ensemble=True, ensemble_models=ens, blur_estimator=est, LEHE=True)
with contextlib.redirect_stdout(sys.stderr):
    engine.evaluate(None, L(batches[:16]), dev, **kw)          # warm-up: every detector's graph captured
    torch.cuda.synchronize(); t0 = time.perf_counter()
    engine.evaluate(None, batches, dev, **kw)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / N * 1e3
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        engine.evaluate(None, L(batches[:20]), dev, **kw)
        torch.cuda.synchronize()
tot = {}
iv = []
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CUDA:
        continue
    n = ev.name
    g = ("conv/gemm" if any(t in n.lower() for t in ("igemm", "conv", "cijk", "winograd", "sp3", "gemm")) else
         "dib" if "dib::" in n else "memcpy" if "memcpy" in n.lower() or "copyBuffer" in n else "sort/topk" if any(t in n for t in ("topk", "Sort", "sort", "rocprim")) else "elementwise/other")
    tot.setdefault(g, [0.0, 0])
    tot[g][0] += ev.device_time * 1e-3 / 20
    tot[g][1] += 1 / 20
    iv.append((ev.time_range.start, ev.time_range.start + ev.device_time))
iv.sort()
busy, (s0, e0) = 0.0, iv[0]
for s, e in iv[1:]:
    if s <= e0:
        e0 = max(e0, e)
    else:
        busy += e0 - s0; s0, e0 = s, e
busy += e0 - s0
print("wall per image %.2f ms (unprofiled); GPU busy per image %.2f ms (union of kernel intervals, profiled run)" % (wall, busy * 1e-3 / 20))
for g, (ms, cnt) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print("  %-20s %6.2f ms  %6.0f launches per image" % (g, ms, cnt))
