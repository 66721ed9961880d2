"""The detector's train step (b = 8, fp32) at one padded batch shape: python scratch/t_shape_steps.py H W [steps] -> first-step seconds, steady ms."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detectinblur_amd import kernel_choices, utils
from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
kernel_choices.use_shipped_kernel_choices()
H, W = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = fasterrcnn_resnet50_fpn(num_classes=91, pretrained=False, pretrained_backbone=False).to(dev).train()
opt = utils.make_sgd([p for p in model.parameters() if p.requires_grad], 0.0004, 0.9, 1e-4)
g = torch.Generator().manual_seed(1)
imgs = [torch.rand(3, min(H, 1333), min(W, 1333), generator=g).to(dev) for _ in range(8)]
tg = [{"boxes": torch.tensor([[10.0, 20.0, 300.0, 400.0], [200.0, 100.0, 700.0, 600.0]], device=dev), "labels": torch.tensor([3, 7], device=dev)} for _ in range(8)]
ts = []
for it in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = sum(model(list(imgs), [dict(t) for t in tg]).values())
    opt.zero_grad(); loss.backward(); opt.step()
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(json.dumps({"shape": [H, W], "first_step_s": round(ts[0], 2), "steady_ms": round(1e3 * sorted(ts[2:])[len(ts[2:]) // 2], 2) if steps > 2 else None}))
