#!/bin/bash
# Extends the shipped kernel-choice data to the grid of batch-1 input sizes real COCO images reach after the detector's transform
# (min side 800, max side 1333, padded to a multiple of 32): MIOpen find-db (normal find, in place) and TunableOp (tuning on).
#   gpurun --timeout 3300 -- bash scratch/fill_dbs_grid.sh [first] [last]     then copy gpurun_out/grid_* into the package
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cp detectinblur_amd/tunableop/tunableop_results.csv gpurun_out/grid_tunableop0.csv
export DIB_MIOPEN_DB_INPLACE=1 DIB_NO_TUNABLEOP=1 DIB_NO_GRAPHS=1
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=gpurun_out/grid_tunableop.csv
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=15 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=3
wc -l detectinblur_amd/miopen_db/*.ufdb.txt gpurun_out/grid_tunableop0.csv
timeout ${GRID_SECONDS:-2700} python3 - "$@" <<'PY' 2>&1 | grep -E "size|done" | tail -5
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from torch import nn
from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
from detectinblur_amd.models.blur_estimator import resnet18
from detectinblur_amd.models import net_transforms
sides = list(range(800, 1345, 32))
sizes = [(800, w) for w in sides] + [(h, 800) for h in sides[1:]] + [(h, 1344) for h in range(512, 800, 32)] + [(1344, w) for w in range(512, 800, 32)]
first, last = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, len(sizes))
m = fasterrcnn_resnet50_fpn(num_classes=91, pretrained=False, pretrained_backbone=False).cuda().eval()
est = resnet18(); est.fc = nn.Linear(512, 4); est = est.cuda().eval()
batcher = net_transforms.GeneralizedRCNNTransform(800, 1333, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], crop_images=True)
mean, std = np.tile([0.485, 0.456, 0.406], (1, 1)), np.tile([0.229, 0.224, 0.225], (1, 1))
t0 = time.time()
with torch.no_grad():
    for k, (h, w) in enumerate(sizes[first:last]):
        hh, ww = min(h, 1333), min(w, 1333)                       # the transform keeps sizes whose sides already satisfy 800 / 1333
        img = torch.rand(3, hh, ww, device="cuda")
        m([img], newMeans=mean, newSTDs=std)
        b, _ = batcher([img], None)
        est(b.tensors)
        torch.cuda.synchronize()
        print("size %d x %d (%d of %d) at %.0f s" % (hh, ww, first + k + 1, len(sizes), time.time() - t0), flush=True)
print("done")
PY
wc -l detectinblur_amd/miopen_db/*.ufdb.txt gpurun_out/grid_tunableop0.csv
cp detectinblur_amd/miopen_db/*.ufdb.txt gpurun_out/grid_miopen.ufdb.txt
