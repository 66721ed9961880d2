import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import numpy as np, torch
import dib_oracle as O, golden_inputs as GI
from detectinblur_amd.models import blur_functions as BF
from detectinblur_amd import blur_ops
for name in sys.argv[1:]:
    case = [c for c in GI.blur_cases() if c["name"] == name][0]
    img = GI.make_image(case); psf = GI.make_case_psf(case)
    out = BF.manual_blur(torch.from_numpy(img).cuda(), torch.from_numpy(psf).cuda()).cpu().numpy()
    want = O.manual_blur(img, psf)
    d = np.argwhere(out.view(np.uint16) != want.view(np.uint16))
    print(name, "shape", out.shape, "mismatches", len(d))
    if len(d):
        print(" channels", np.unique(d[:,0]), "rows", d[:,1].min(), d[:,1].max(), "cols", np.unique(d[:,2])[:40])
        rr, cc, ww = O.taps_of(psf)
        print(" taps", len(rr), "r", rr.min(), rr.max(), "c", cc.min(), cc.max())
        for (c,y,x) in d[:5]: print("  ", c,y,x, out[c,y,x], want[c,y,x])
