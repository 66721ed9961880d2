"""Same-box A/B of the train step (bench.train_step_bench's step) with the bias gradients through dib_channel_sum_nhwc vs torch's sum."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from detectinblur_amd import kernel_choices, utils
from detectinblur_amd.models import backbone as B
from detectinblur_amd.models import blur_functions as BF
from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
kernel_choices.use_shipped_kernel_choices()
dev = torch.device("cuda", 0)
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev)
torch.manual_seed(1337)
model = fasterrcnn_resnet50_fpn(num_classes=91, pretrained=False, pretrained_backbone=False).to(dev).train()
opt = utils.make_sgd([p for p in model.parameters() if p.requires_grad], 0.0004, 0.9, 1e-4)
g = torch.Generator().manual_seed(4242)
targets = []
for _ in range(8):
    x1 = torch.rand(8, generator=g) * (1333 - 34); y1 = torch.rand(8, generator=g) * (800 - 34)
    w = 32 + torch.rand(8, generator=g) * 368; h = 32 + torch.rand(8, generator=g) * 368
    targets.append({"boxes": torch.stack([x1, y1, torch.clamp(x1 + w, max=1333), torch.clamp(y1 + h, max=800)], 1).to(dev), "labels": torch.randint(1, 91, (8,), generator=g).to(dev)})
means, stds = utils.get_norm_params(dicts, False)
def step():
    batch = list(images)
    BF.blur_image_list(batch, dicts, psfs)
    loss = sum(model([b.float() for b in batch], [{k: v.clone() for k, v in t.items()} for t in targets], newMeans=means, newSTDs=stds).values())
    opt.zero_grad(); loss.backward(); opt.step()
for _ in range(6): step()
for rnd in range(3):
    for flag in (True, False):
        B.CHANNEL_SUM = flag
        for _ in range(2): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(12): step()
        torch.cuda.synchronize()
        print("round %d: bias gradients through %s: %.2f ms per step" % (rnd, "dib_channel_sum_nhwc" if flag else "torch.sum", (time.perf_counter() - t0) / 12 * 1e3), flush=True)
