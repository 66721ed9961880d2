#!/bin/bash
# MIOpen solver tuning (MIOPEN_FIND_ENFORCE=SEARCH) of the evaluation path's convolutions (batch 1: detector at the drivers' sizes,
# the estimator on its crop).   gpurun --timeout 3300 -- bash scratch/tune_miopen_eval.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
D=$GRAFT_REPO_ROOT/gpurun_out/miopen_tune_eval; rm -rf $D; mkdir -p $D; cp detectinblur_amd/miopen_db/*.txt $D/
echo "== before"; WARM=2 N=30 python3 scratch/t_graph_speed.py 2>&1 | grep -E "trunk graph replay"; python3 scratch/t_eval_anatomy.py 2>/dev/null | head -1
t0=$(date +%s)
MIOPEN_USER_DB_PATH=$D MIOPEN_FIND_ENFORCE=3 DIB_NO_GRAPHS=1 timeout ${TUNE_SECONDS:-1500} python3 - <<'PY' > gpurun_out/miopen_tune_eval.log 2>&1
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from torch import nn
from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
from detectinblur_amd.models.blur_estimator import resnet18
from detectinblur_amd.models import net_transforms
m = fasterrcnn_resnet50_fpn(num_classes=91, pretrained=False, pretrained_backbone=False).cuda().eval()
est = resnet18(); est.fc = nn.Linear(512, 4); est = est.cuda().eval()
batcher = net_transforms.GeneralizedRCNNTransform(800, 1333, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], crop_images=True)
with torch.no_grad():
    for hw in ((800, 1333), (800, 1088), (1333, 800), (800, 1066), (800, 1200)):
        img = torch.rand(3, *hw, device="cuda")
        m([img], newMeans=np.tile([0.485, 0.456, 0.406], (1, 1)), newSTDs=np.tile([0.229, 0.224, 0.225], (1, 1)))
        b, _ = batcher([img], None)
        est(b.tensors)
        print("tuned", hw, flush=True)
PY
echo "tuning rc $? after $(( $(date +%s) - t0 )) s"; tail -2 gpurun_out/miopen_tune_eval.log; wc -l $D/*.txt
echo "== after"; MIOPEN_USER_DB_PATH=$D WARM=2 N=30 python3 scratch/t_graph_speed.py 2>&1 | grep -E "trunk graph replay"; MIOPEN_USER_DB_PATH=$D python3 scratch/t_eval_anatomy.py 2>/dev/null | head -1
