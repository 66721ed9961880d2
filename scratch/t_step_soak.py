"""Soak of the blur step's single launch (in-launch compaction + hand-off): N steps with a PSF set that changes every step (cycling through
64 random sets of 8, rasterised trajectories of all blur types), the BASELINE batch and a ragged one alternating; every 1,000th
step compared bit for bit with compaction + blur as two launches.  A hand-off that ever failed to make progress would trap
(hipErrorLaunchFailure) within a second."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from detectinblur_amd import _lib, blur_ops
from detectinblur_amd.models import blur_functions as BF
dev = torch.device("cuda", 0)
n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev)
ragged = [torch.rand(3, h, w, generator=torch.Generator().manual_seed(31 + i)).half().to(dev) for i, (h, w) in enumerate(bench.COCO_NATIVE_SIZES)]
rs = np.random.RandomState(5)
sets = []
for s in range(64):
    ps = []
    for k in range(8):
        a = np.zeros((128, 128), np.float64)
        n = int(rs.randint(3, 200)); sp = int(rs.choice([2, 6, 14, 30, 62]))
        a[np.clip(rs.randint(-sp, sp + 1, n) + 63, 0, 127), np.clip(rs.randint(-sp, sp + 1, n) + 63, 0, 127)] = rs.random_sample(n) + 0.01
        ps.append(torch.from_numpy((a / a.sum()).astype(np.float16)).to(dev))
    sets.append(ps)
sets.append(list(psfs))
l = _lib.lib()
import ctypes
l.dib_debug_set_step_fused.argtypes = [ctypes.c_int]; l.dib_debug_set_step_fused.restype = None
t0 = time.time(); checked = 0
for it in range(n_steps):
    ps = sets[it % len(sets)]
    batch = list(images if it % 2 == 0 else ragged)
    BF.blur_image_list(batch, dicts, ps, psfs_complete=True)
    if it % 1000 == 0:
        l.dib_debug_set_step_fused(0)
        ref = list(images if it % 2 == 0 else ragged)
        BF.blur_image_list(ref, dicts, ps, psfs_complete=True)
        l.dib_debug_set_step_fused(1)
        assert all(torch.equal(a, b) for a, b in zip(batch, ref)), it
        checked += 1
torch.cuda.synchronize()
print("%d steps in %.0f s (%d compared bit for bit with the two-launch path): no trap, no difference" % (n_steps, time.time() - t0, checked))
