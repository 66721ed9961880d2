import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from detectinblur_amd.models import detector_ops as ops
from test_detector_ops import _rois
rs = np.random.RandomState(3)
shape=(2,16,50,68); scale=0.25
feat = torch.tensor(rs.randn(*shape), dtype=torch.float32); rois=_rois(rs,37,2,50,68,scale)
ref = ops.roi_align_torch(feat, rois, scale, 7, 2)
out = ops.roi_align(feat.cuda(), rois.cuda(), scale, 7, 2).cpu()
d = (out-ref).abs()
print("max diff", d.max().item(), "count>1e-5", (d>1e-5).sum().item())
idx = (d>1e-5).nonzero()
for k,c,ph,pw in idx[:6].tolist(): print(k,c,ph,pw, out[k,c,ph,pw].item(), ref[k,c,ph,pw].item(), rois[k].tolist())
