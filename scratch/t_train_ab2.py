"""Same-box A/B of the detector train step: FPN top-down merge (FUSE_TOPDOWN) and fused RPN head (FUSE_HEAD) on / off."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench as B
from detectinblur_amd.models import backbone as BB
from detectinblur_amd.models import rpn as RR
dev = torch.device("cuda", 0)
host = B.make_psfs_host(0)
images, dicts, psfs, _, _ = B.make_workload(0, dev, host)
for rnd in range(2):
    for td, head in ((True, True), (False, True), (True, False), (False, False)):
        BB.FUSE_TOPDOWN, RR.FUSE_HEAD = td, head
        tr, ddp, opt = B.train_step_bench(images, dicts, psfs, dev, None, 1, 0, 10, 3, account=False)
        print("round %d fuse_topdown=%d fuse_head=%d: %.2f ms/step" % (rnd, td, head, tr["ms_per_step"]), flush=True)
        del ddp, opt
        torch.cuda.empty_cache()
