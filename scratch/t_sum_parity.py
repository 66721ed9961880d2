"""How often does the exactly-summed-then-rounded fp16 PSF sum of the compaction kernel differ from what the reference
computes, `psf_GPU.sum()` on a Half CUDA tensor (fp32 accumulation in torch's reduction order, one final rounding)?"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from detectinblur_amd import blur_ops
from detectinblur_amd.motion_blur.generate_PSF import PSF
from detectinblur_amd.motion_blur.generate_trajectory import Trajectory
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
np.random.seed(7)
rs = np.random.RandomState(8)
fr = [1 / 18, 1 / 10, 1 / 5, 1 / 2, 1]
psfs = []
for i in range(N):
    tr = Trajectory(canvas=256, max_len=96, expl=[0.005, 0.001, 0.00005][rs.randint(3)]).fit().fit()
    p = PSF(canvas=256, trajectory=tr, fraction=[fr[rs.randint(5)]]); p.fit(); p.centerPSF()
    psfs.append(torch.HalfTensor(np.ascontiguousarray(p.PSFs[0][64:192, 64:192])))
stack = torch.stack(psfs).cuda()
ref = torch.stack([stack[i].sum() for i in range(N)]).cpu().view(torch.int16).numpy()
mine = np.zeros(N, np.int16)
for b0 in range(0, N, 32):
    t = blur_ops.compact_psfs(stack[b0:b0 + 32].contiguous(), normalize=True)
    for k in range(min(32, N - b0)):
        mine[b0 + k] = np.int16(t.buf[k * t.words + 6].item() & 0xffff if (t.buf[k * t.words + 6].item() & 0xffff) < 32768 else (t.buf[k * t.words + 6].item() & 0xffff) - 65536)
diff = int((ref != mine).sum())
print("PSFs: %d   sums that differ from torch's Half .sum(): %d (%.3f %%)" % (N, diff, 100.0 * diff / N))
if diff:
    idx = np.nonzero(ref != mine)[0][:5]
    for i in idx:
        print("  psf %d: torch %s  kernel %s  fp64 %.10f" % (i, np.array([ref[i]]).view(np.float16)[0], np.array([mine[i]]).view(np.float16)[0], float(stack[i].double().sum())))
