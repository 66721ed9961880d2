"""Same-box A/B of the detector train step: fused stem (FUSE_STEM_POOL) on / off, and all of this session's fusions off
(FPN top-down merge, RPN head, stem pool) against all on."""
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
    for stem, rest in ((True, True), (False, True), (False, False)):
        BB.FUSE_STEM_POOL, BB.FUSE_TOPDOWN, RR.FUSE_HEAD = stem, rest, rest
        tr, ddp, opt = B.train_step_bench(images, dicts, psfs, dev, None, 1, 0, 10, 3, account=False)
        print("round %d stem_pool=%d topdown+head=%d: %.2f ms/step" % (rnd, stem, rest, tr["ms_per_step"]), flush=True)
        del ddp, opt
        torch.cuda.empty_cache()
