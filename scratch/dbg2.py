import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import numpy as np, torch
from detectinblur_amd.models import blur_functions as BF
H, W = 70, 300
xs = torch.arange(W, dtype=torch.float16).view(1,1,W).expand(1,H,W).contiguous().cuda()
ys = torch.arange(H, dtype=torch.float16).view(1,H,1).expand(1,H,W).contiguous().cuda()
for (r,c) in ((63,63),(60,66)):
    psf = torch.zeros(128,128,dtype=torch.float16,device='cuda'); psf[r,c]=1
    ox = BF.manual_blur(xs, psf).cpu().numpy(); oy = BF.manual_blur(ys, psf).cpu().numpy()
    print("tap",r,c)
    print(" src col for x=0..9:", ox[5,:10]); print(" x=60..70:", ox[5,60:71]); print(" x=124..132:", ox[5,124:133]); print(" x=188..196", ox[5,188:197]); print(" x=252..262", ox[5,252:263])
    print(" src row for y=0..9 @x=3:", oy[:10,3], "@x=70:", oy[:10,70], "@x=200", oy[:10,200])
