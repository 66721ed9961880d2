"""The large LDS window vs the standard one: blur kernel time for one 3 x 800 x 1333 image (b = 1) and for the batch of 8, over
the sweep's 15 (blur type, exposure) cells with on-the-fly PSFs (seeded), and the BASELINE batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from detectinblur_amd import blur_ops
from detectinblur_amd.motion_blur.generate_PSF import PSF
from detectinblur_amd.motion_blur.generate_trajectory import Trajectory
dev = torch.device("cuda", 0)
img = [torch.rand(3, 800, 1333, generator=torch.Generator().manual_seed(7 + i)).half().to(dev) for i in range(8)]
np.random.seed(11)
print("%-8s %5s %9s %9s | %8s %8s | %8s %8s" % ("cell", "taps", "rows x c", "segs s/L", "b1 std", "b1 large", "b8 std", "b8 large"))
for pi, expl in enumerate((0.005, 0.001, 0.00005), 1):
    for ei, frac in enumerate((1 / 25, 1 / 10, 1 / 5, 1 / 2, 1.0)):
        psfs = []
        for k in range(8):
            tr = Trajectory(canvas=256, max_len=96, expl=expl).fit().fit()
            p = PSF(canvas=256, trajectory=tr, fraction=[frac]); p.fit(); p.centerPSF()
            psfs.append(torch.HalfTensor(np.ascontiguousarray(p.PSFs[0][64:192, 64:192])).to(dev))
        r, c = np.nonzero(psfs[0].cpu().numpy())
        res = {}
        for large in (False, True):
            t1 = blur_ops.compact_psfs(psfs[:1], True, large)
            t8 = blur_ops.compact_psfs(psfs, True, large)
            for _ in range(10):
                blur_ops.sparse_blur(img[:1], [0], t1); blur_ops.sparse_blur(img, list(range(8)), t8)
            res[large] = (sorted(bench.kernel_time_ms(lambda k: blur_ops.sparse_blur(img[:1], [0], t1), 100) for _ in range(3))[1] * 1e3,
                          sorted(bench.kernel_time_ms(lambda k: blur_ops.sparse_blur(img, list(range(8)), t8), 50) for _ in range(3))[1] * 1e3,
                          len(t1.segments(0)))
        print("P%dE%d     %5d %4dx%-4d %4d/%-4d | %8.1f %8.1f | %8.1f %8.1f" % (pi, ei, len(r), r.max() - r.min() + 1, c.max() - c.min() + 1, res[False][2], res[True][2],
                                                                     res[False][0], res[True][0], res[False][1], res[True][1]), flush=True)
