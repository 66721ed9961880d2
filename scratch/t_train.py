"""Train-step A/B: variants given on the command line, e.g.  base fold cl fold+cl"""
import sys, time, os
sys.path.insert(0, '.')
import torch, bench
from detectinblur_amd import utils
from detectinblur_amd.models import blur_functions as BF, backbone as BB
from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev)
H, W, BATCH = bench.H, bench.W, bench.BATCH
def run(variant, steps=6, warmup=3):
    opts = set(variant.split("+"))
    BB.FOLD_FROZEN_BN = "fold" in opts
    torch.backends.cudnn.benchmark = "bm" in opts
    torch.manual_seed(1337)
    model = fasterrcnn_resnet50_fpn(num_classes=91, pretrained=False, pretrained_backbone=False).to(dev)
    if "cl" in opts:
        model = model.to(memory_format=torch.channels_last)
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
        batch = list(images)
        tg = [{k: v.clone() for k, v in t.items()} for t in targets]
        BF.blur_image_list(batch, dicts, psfs)
        tg = utils.expand_targets(tg, dicts, psfs, batch)
        batch = [b.float() for b in batch]
        losses = sum(model(batch, tg, newMeans=means, newSTDs=stds).values())
        opt.zero_grad(); losses.backward(); opt.step()
        return losses
    t0 = time.perf_counter()
    for _ in range(warmup): l = step()
    torch.cuda.synchronize(); tw = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(steps): l = step()
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print("%-12s %.1f ms/step  %.1f img/s  loss %.4f  (warmup %.0f s)" % (variant, el / steps * 1e3, BATCH * steps / el, l.item(), tw), flush=True)
for v in sys.argv[1:]:
    run(v)
