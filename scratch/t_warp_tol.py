import sys; sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
import numpy as np, torch
import gen_goldens as GG
from detectinblur_amd.models.warper import Warper
g=np.load('tests/golden/warper.npz')
x, feat, th, l1, l2 = GG.warper_inputs()
w=Warper()
xs=GG.warper_smooth_input()
got=w(xs.cuda(), th.cuda(), l1.cuda(), l2.cuda()).cpu().numpy()
d=np.abs(got-g['warp_smooth']); print("smooth: max %.4g mean %.4g" % (d.max(), d.mean()))
got=w(x.cuda(), th.cuda(), l1.cuda(), l2.cuda()).cpu().numpy()
d=np.abs(got-g['warp_image']); print("noise: max %.4g mean %.4g" % (d.max(), d.mean()))
got=w(xs.cuda(), th.cuda(), l1.cuda(), l2.cuda()).cpu().numpy(); ref=g['warp_smooth']
d=np.abs(got-ref)
print("smooth percentiles 50/99/99.9/max:", [float(np.percentile(d,q)) for q in (50,99,99.9,100)])
sh=np.abs(np.roll(ref,1,axis=3)-ref)
print("one-pixel shift: mean %.4g  99th %.4g" % (sh.mean(), np.percentile(sh,99)))
inside=(ref>0)&(np.roll(ref,1,3)>0)&(np.roll(ref,-1,3)>0)&(np.roll(ref,1,2)>0)&(np.roll(ref,-1,2)>0)
print("interior (all 4 neighbours inside the warped image): max %.4g mean %.4g  n=%d of %d" % (d[inside].max(), d[inside].mean(), inside.sum(), inside.size))
