"""Same-box A/B of the detector train step (bench.py's resident train_step), round 4: box bookkeeping as HIP launches
(csrc/dib_detect.hip, ops.HIP_BOXES) vs the batched tensor expressions."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench as B
from detectinblur_amd.models import detector_ops as ops
dev = torch.device("cuda", 0)
host = B.make_psfs_host(0)
images, dicts, psfs, _, _ = B.make_workload(0, dev, host)
for rnd in range(3):
    for flag in (True, False):
        ops.HIP_BOXES = flag
        tr, ddp, opt = B.train_step_bench(images, dicts, psfs, dev, None, 1, 0, 12, 4, account=False)
        print("round %d hip_boxes=%d: %.2f ms/step" % (rnd, flag, tr["ms_per_step"]), flush=True)
        del ddp, opt
        torch.cuda.empty_cache()
