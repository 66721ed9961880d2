#!/bin/bash
# Regenerate detectinblur_amd/tunableop/tunableop_results.csv: PyTorch TunableOp's recorded GEMM solution per shape for the bench,
# the evaluation drivers (b = 1 at 800 x 1333 / 800 x 1088, estimator included) and the train step (b = 8).
#   gpurun -- bash scratch/fill_tunableop.sh      (writes gpurun_out/tunableop_results.csv; copy into the package)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/tunableop_fill*.csv
export DIB_NO_TUNABLEOP=1 PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=gpurun_out/tunableop_fill.csv
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=15 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=3
DIB_NO_GRAPHS=1 timeout 1200 python3 scratch/t_eval_anatomy.py > /dev/null 2>&1; echo "eval loop: rc $?"; wc -l gpurun_out/tunableop_fill0.csv
timeout 1200 python3 - <<'PY' 2>&1 | tail -2
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
m = fasterrcnn_resnet50_fpn(num_classes=91, pretrained=False, pretrained_backbone=False).cuda().eval()
with torch.no_grad():
    for hw in ((800, 1333), (800, 1088), (1333, 800), (800, 1066), (800, 1200)):
        img = torch.rand(3, *hw, device="cuda")
        m([img], newMeans=np.tile([0.485, 0.456, 0.406], (1, 1)), newSTDs=np.tile([0.229, 0.224, 0.225], (1, 1)))
print("inference sizes done")
PY
wc -l gpurun_out/tunableop_fill0.csv
timeout 2400 python3 scratch/train_only.py 2 2>&1 | tail -1
wc -l gpurun_out/tunableop_fill0.csv
cp gpurun_out/tunableop_fill0.csv gpurun_out/tunableop_results.csv
