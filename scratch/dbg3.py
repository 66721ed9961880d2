import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import numpy as np, torch
from detectinblur_amd.models import blur_functions as BF
g = torch.Generator().manual_seed(1337)
img = torch.rand(3, 800, 1333, generator=g).half().cuda()
psf = torch.zeros(128, 128, dtype=torch.float16, device="cuda"); psf[63, 63] = 1
for rep in range(3):
    out = BF.manual_blur(img, psf)
    d = (out != img).nonzero().cpu().numpy()
    print("rep", rep, "mismatches", len(d))
    if len(d):
        print(" ch", np.unique(d[:,0]), "rows", d[:,1].min(), d[:,1].max(), np.unique(d[:,1]//32)[:20], "cols", d[:,2].min(), d[:,2].max(), np.unique(d[:,2]//256))
        tiles = np.unique(np.stack([d[:,0], d[:,1]//32, d[:,2]//256],1), axis=0)
        print(" bad tiles", len(tiles), tiles[:10].tolist())
        c,y,x = d[0]; print(" first", c,y,x, float(out[c,y,x]), float(img[c,y,x]))
