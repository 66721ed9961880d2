"""RoIHeads.postprocess_detections at the evaluation size (1000 RoIs, 91 classes): kernels vs tensor path, wall time per image."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from detectinblur_amd.models import detector_ops as ops
from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
from tests.test_detect_gpu import _head_outputs
torch.manual_seed(0)
heads = fasterrcnn_resnet50_fpn(pretrained=False, pretrained_backbone=False, num_classes=91).roi_heads
for spread in (0.5, 1.5, 4.0):
    logits, deltas, rois = _head_outputs(1, 1000, spread=spread)
    for flag in (True, False):
        ops.HIP_BOXES = flag
        f = lambda: heads.postprocess_detections(logits, deltas, [rois], [(800, 1333)])
        for _ in range(3): o = f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): f()
        torch.cuda.synchronize()
        print("logit spread %.1f hip=%d: %.0f us per image (%d detections)" % (spread, flag, (time.perf_counter() - t0) / 30 * 1e6, o[0]["scores"].numel()), flush=True)
ops.HIP_BOXES = True
