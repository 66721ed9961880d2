"""Graphed inference at b = 1, 800 x 1333 (what the evaluation sweep runs per image), a dozen times: for scratch/prof_trunk.sh."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
torch.manual_seed(0)
m = fasterrcnn_resnet50_fpn(num_classes=91, pretrained=False, pretrained_backbone=False).cuda().eval()
img = torch.rand(3, 800, 1333, device="cuda")
means, stds = np.tile([0.485, 0.456, 0.406], (1, 1)), np.tile([0.229, 0.224, 0.225], (1, 1))
m.graph_inference = True
with torch.no_grad():
    for k in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m([img], newMeans=means, newSTDs=stds)
        torch.cuda.synchronize()
        print("forward %d: %.2f ms" % (k, (time.perf_counter() - t0) * 1e3), file=sys.stderr, flush=True)
