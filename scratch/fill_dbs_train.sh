#!/bin/bash
# Extends the shipped kernel-choice data to the padded BATCH shapes real COCO training reaches at b = 8 (min side 800, max side 1333,
# batch padded to a multiple of 32; aspect-ratio groups keep a batch's images alike): without a find-db record MIOpen times every
# solver of every convolution the first time it meets a shape -- 67-75 s per batch shape (scratch/t_new_shape.py) -- and the FAST
# find mode that skips it runs the step at 1.7 s instead of 92 ms.  Normal find, in place; TunableOp with tuning on.
#   gpurun --timeout 3300 -- bash scratch/fill_dbs_train.sh [first] [last]     then copy gpurun_out/train_* into the package
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export DIB_MIOPEN_DB_INPLACE=1 DIB_NO_TUNABLEOP=1
cp detectinblur_amd/tunableop/tunableop_results.csv gpurun_out/train_tunableop0.csv
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=gpurun_out/train_tunableop.csv
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=15 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=3
wc -l detectinblur_amd/miopen_db/*.ufdb.txt gpurun_out/train_tunableop0.csv
timeout ${GRID_SECONDS:-3000} python3 - "$@" <<'PY' 2>&1 | grep -E "shape|done" | tail -40
import sys, time; sys.path.insert(0, '.')
import torch
from detectinblur_amd import kernel_choices, utils
from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
kernel_choices.use_shipped_kernel_choices()
sides = list(range(800, 1345, 32))
shapes = [(800, w) for w in sides] + [(h, 800) for h in sides[1:]] + [(768, 1344), (1344, 768), (736, 1344), (1344, 736), (1088, 1088), (1024, 1024)]
first, last = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, len(shapes))
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = fasterrcnn_resnet50_fpn(num_classes=91, pretrained=False, pretrained_backbone=False).to(dev).train()
opt = utils.make_sgd([p for p in model.parameters() if p.requires_grad], 0.0004, 0.9, 1e-4)
t0 = time.time()
for k, (H, W) in enumerate(shapes[first:last]):
    hh, ww = min(H, 1333), min(W, 1333)              # the largest image of such a batch: the transform pads the batch to (H, W)
    g = torch.Generator().manual_seed(k)
    imgs = [torch.rand(3, hh, ww, generator=g).to(dev) for _ in range(8)]
    tg = [{"boxes": torch.tensor([[10.0, 20.0, 300.0, 400.0], [200.0, 100.0, 700.0, 600.0]], device=dev), "labels": torch.tensor([3, 7], device=dev)} for _ in range(8)]
    ts = []
    for it in range(3):
        torch.cuda.synchronize(); t1 = time.perf_counter()
        loss = sum(model(list(imgs), [dict(t) for t in tg]).values())
        opt.zero_grad(); loss.backward(); opt.step()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t1)
    print("shape %d x %d (%d of %d): first step %.1f s, third %.1f ms; %.0f s so far" % (H, W, first + k + 1, len(shapes), ts[0], ts[2] * 1e3, time.time() - t0), flush=True)
print("done")
PY
wc -l detectinblur_amd/miopen_db/*.ufdb.txt gpurun_out/train_tunableop*.csv
cp detectinblur_amd/miopen_db/*.ufdb.txt gpurun_out/train_miopen.ufdb.txt
cp detectinblur_amd/miopen_db/*.udb.txt gpurun_out/train_miopen.udb.txt
