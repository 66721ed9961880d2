"""DDP / RCCL smoke on one GPU: world_size 1 under torchrun, model wrapped in DistributedDataParallel."""
import os, sys, time
sys.path.insert(0, '.')
import torch, torch.distributed as dist
import bench
from detectinblur_amd import utils
from detectinblur_amd.models import blur_functions as BF
from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local); dev = torch.device("cuda", local)
dist.init_process_group("nccl", init_method="env://", device_id=dev)
images, dicts, psfs, psfs_host, _ = bench.make_workload(rank, dev)
torch.manual_seed(1337)
model = fasterrcnn_resnet50_fpn(num_classes=91, pretrained=False, pretrained_backbone=False).to(dev)
kw = eval(os.environ.get("DDP_KW", "{}"))
ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local], **kw)
opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.0004, momentum=0.9, weight_decay=1e-4)
g = torch.Generator().manual_seed(4242)
H, W = bench.H, bench.W
targets = []
for _ in range(8):
    x1 = torch.rand(8, generator=g) * (W - 34); y1 = torch.rand(8, generator=g) * (H - 34)
    w = 32 + torch.rand(8, generator=g) * 368; h = 32 + torch.rand(8, generator=g) * 368
    targets.append({"boxes": torch.stack([x1, y1, torch.clamp(x1 + w, max=W), torch.clamp(y1 + h, max=H)], 1).to(dev),
                    "labels": torch.randint(1, 91, (8,), generator=g).to(dev)})
means, stds = utils.get_norm_params(dicts, False)
ddp.train()
def step():
    batch = list(images); tg = [{k: v.clone() for k, v in t.items()} for t in targets]
    BF.blur_image_list(batch, dicts, psfs); tg = utils.expand_targets(tg, dicts, psfs, batch)
    losses = sum(ddp([b.float() for b in batch], tg, newMeans=means, newSTDs=stds).values())
    opt.zero_grad(); losses.backward(); opt.step(); return losses
for _ in range(3): l = step()
dist.barrier(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(6): l = step()
dist.barrier(); torch.cuda.synchronize()
print(str(kw), "ddp world=%d: %.1f ms/step, loss %.4f finite=%s" % (world, (time.perf_counter() - t0) / 6 * 1e3, l.item(), bool(torch.isfinite(l))))
dist.destroy_process_group()
