"""FLOPs of one fasterrcnn_resnet50_fpn training step at BASELINE shapes (b = 8, 3 x 800 x 1333, fp32), counted by
torch.utils.flop_counter on the real step (forward + backward; convolutions and matmuls), for the achieved-TFLOP/s
line of profiles/r2_train_step_conv.txt:  python scratch/train_flops.py  -> JSON on stdout."""
import json, sys
sys.path.insert(0, '.')
import numpy as np, torch
from torch.utils.flop_counter import FlopCounterMode
import bench
from detectinblur_amd import engine, utils
from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
dev = torch.device("cuda", 0)
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev)
torch.manual_seed(1337)
model = fasterrcnn_resnet50_fpn(num_classes=91, pretrained=False, pretrained_backbone=False).to(dev).train()
g = torch.Generator().manual_seed(4242)
targets = []
for _ in range(8):
    x1 = torch.rand(8, generator=g) * (1333 - 34); y1 = torch.rand(8, generator=g) * (800 - 34)
    w = 32 + torch.rand(8, generator=g) * 368; h = 32 + torch.rand(8, generator=g) * 368
    boxes = torch.stack([x1, y1, torch.clamp(x1 + w, max=1333), torch.clamp(y1 + h, max=800)], 1)
    targets.append({"boxes": boxes.to(dev), "labels": torch.randint(1, 91, (8,), generator=g).to(dev)})
means, stds = utils.get_norm_params(dicts, False)
def step():
    batch = engine._to_float(list(images), model, dev)
    loss = sum(model(batch, [dict(t) for t in targets], newMeans=means, newSTDs=stds).values())
    model.zero_grad(); loss.backward()
for _ in range(2): step()
with FlopCounterMode(display=False) as fc:
    step()
tot = fc.get_flop_counts()["Global"]
conv = sum(v for k, v in tot.items() if "conv" in str(k))
mm = sum(v for k, v in tot.items() if "mm" in str(k))
print(json.dumps({"total_flops_per_step": int(sum(tot.values())), "conv_flops_per_step": int(conv), "matmul_flops_per_step": int(mm),
                  "by_op": {str(k): int(v) for k, v in tot.items()}}))
