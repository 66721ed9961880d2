import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import numpy as np, torch
import gen_goldens as GG
from detectinblur_amd.models.warper import Warper, squint_matrices
g = np.load("tests/golden/warper.npz")
x, feat, th, l1, l2 = GG.warper_inputs()
w = Warper()
got = w(x.cuda(), th.cuda(), l1.cuda(), l2.cuda()).cpu().numpy()
d = np.abs(got - g["warp_image"])
print("max", d.max(), "mean", d.mean(), "frac>1e-2", (d > 1e-2).mean())
mc = squint_matrices(th, l1, l2, 56, 40); mg = squint_matrices(th.cuda(), l1.cuda(), l2.cuda(), 56, 40).cpu()
print("matrix diff", (mc.float() - mg.float()).abs().max().item()); print(mc[0], mg[0])
