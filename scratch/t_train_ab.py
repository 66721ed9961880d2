"""Same-box A/B of the detector train step (bench.py's resident train_step): LINEAR_1X1 and RELU_MASK on / off."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench as B
from detectinblur_amd.models import backbone as BB
dev = torch.device("cuda", 0)
host = B.make_psfs_host(0)
images, dicts, psfs, _, _ = B.make_workload(0, dev, host)
for rnd in range(2):
    for lin, mask, entry, planar in ((True, True, True, True), (True, True, False, True), (False, False, False, False)):
        BB.LINEAR_1X1, BB.RELU_MASK, BB.BLOCK_ENTRY, BB.NCHW_SMALL_3X3 = lin, mask, entry, planar
        tr, ddp, opt = B.train_step_bench(images, dicts, psfs, dev, None, 1, 0, 10, 3, account=False)   # rank 1: no flop accounting
        print("round %d linear_1x1=%d relu_mask=%d block_entry=%d planar_3x3=%d: %.2f ms/step" % (rnd, lin, mask, entry, planar, tr["ms_per_step"]), flush=True)
        del ddp, opt
        torch.cuda.empty_cache()
