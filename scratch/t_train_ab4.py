"""Same-box A/B of the detector train step (bench.py's resident train_step), round 4: RPN target assignment + RoI sampling for
all images at once vs the per-image loops."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench as B
from detectinblur_amd.models import rpn as R, roi_heads as RH, backbone as BB
dev = torch.device("cuda", 0)
host = B.make_psfs_host(0)
images, dicts, psfs, _, _ = B.make_workload(0, dev, host)
per_image = R.RegionProposalNetwork.assign_targets_per_image
batched = R.RegionProposalNetwork.assign_targets
for rnd in range(2):
    for flag, fold in ((True, True), (True, False), (False, False)):
        RH.RoIHeads.batched = flag
        BB.FOLD_ALL = fold
        R.RegionProposalNetwork.assign_targets = batched if flag else per_image
        tr, ddp, opt = B.train_step_bench(images, dicts, psfs, dev, None, 1, 0, 12, 4, account=False)
        print("round %d batched_targets=%d fold_all=%d: %.2f ms/step" % (rnd, flag, fold, tr["ms_per_step"]), flush=True)
        del ddp, opt
        torch.cuda.empty_cache()
