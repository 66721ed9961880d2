"""Which convolutions of the train step are slow for their FLOPs: torch.profiler (record_shapes) over 3 steps, grouped by
(op, input shapes): device time per step and achieved TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch.profiler import ProfilerActivity, profile
import bench as B
dev = torch.device("cuda", 0)
host = B.make_psfs_host(0)
images, dicts, psfs, _, _ = B.make_workload(0, dev, host)
import types
steps = {}
orig = B.train_step_bench
# reuse bench's step by running its warm-up, then profile via a tiny hook: monkeypatch time.perf_counter is overkill -- rebuild the step
from detectinblur_amd import blur_ops, engine, utils
from detectinblur_amd.models import blur_functions as BF
from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
torch.manual_seed(1337)
model = fasterrcnn_resnet50_fpn(num_classes=91, pretrained=False, pretrained_backbone=False).to(dev).train()
opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.0004, momentum=0.9, weight_decay=1e-4)
g = torch.Generator().manual_seed(4242)
targets = []
for _ in range(8):
    x1 = torch.rand(8, generator=g) * (1333 - 34); y1 = torch.rand(8, generator=g) * (800 - 34)
    w = 32 + torch.rand(8, generator=g) * 368; h = 32 + torch.rand(8, generator=g) * 368
    boxes = torch.stack([x1, y1, torch.clamp(x1 + w, max=1333), torch.clamp(y1 + h, max=800)], 1)
    targets.append({"boxes": boxes.to(dev), "labels": torch.randint(1, 91, (8,), generator=g).to(dev)})
means, stds = utils.get_norm_params(dicts, False)


def step():
    batch = list(images)
    tg = [{k: v.clone() for k, v in t.items()} for t in targets]
    BF.blur_image_list(batch, dicts, psfs)
    batch = engine._to_float(batch, model, dev)
    loss = sum(model(batch, tg, newMeans=means, newSTDs=stds).values())
    opt.zero_grad(); loss.backward(); opt.step()


for _ in range(4):
    step()
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(N):
        step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if any(t in e.key for t in ("convolution", "addmm", "mm", "linear")) and e.device_time_total > 0 and "aten::" in e.key and e.key in ("aten::miopen_convolution", "aten::convolution_backward", "aten::addmm", "aten::mm"):
        rows.append((e.device_time_total / N / 1e3, e.count // N, e.key, str(e.input_shapes)[:150]))
for ms, cnt, key, shp in sorted(rows, reverse=True)[:45]:
    print("%7.3f ms/step  x%-3d %-28s %s" % (ms, cnt, key, shp))
