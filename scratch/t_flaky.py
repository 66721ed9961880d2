import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from detectinblur_amd.models import backbone as B
torch.manual_seed(0)
m = B.ResNet50Body().cuda().to(memory_format=torch.channels_last)
for mod in m.modules():
    if isinstance(mod, B.FrozenBatchNorm2d):
        mod.weight.uniform_(0.5, 1.5); mod.bias.uniform_(-.2, .2); mod.running_mean.uniform_(-.2, .2); mod.running_var.uniform_(0.5, 1.5)
x = torch.randn(2, 3, 96, 128, device="cuda").contiguous(memory_format=torch.channels_last)


def run(fuse):
    B.FUSE_EPILOGUE = fuse
    for p in m.parameters():
        p.grad = None
    ys = m(x)
    sum(y.square().mean() for y in ys).backward()
    return [y.detach().clone() for y in ys], m.layer2[0].conv1.weight.grad.clone(), m.conv1.weight.grad.clone()


ref = run(False)
for it in range(40):
    for fuse in (True, False):
        r = run(fuse)
        yd = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(r[0], ref[0]))
        g1 = float((r[1] - ref[1]).norm() / ref[1].norm()); g2 = float((r[2] - ref[2]).norm() / ref[2].norm())
        flag = " <<<<" if max(yd, g1, g2) > 1e-3 else ""
        if flag or it < 3:
            print("it %2d fuse=%d  out rel %.2e  grad(layer2.0.conv1) rel-norm %.2e  grad(conv1) rel-norm %.2e%s" % (it, fuse, yd, g1, g2, flag), flush=True)
print("done")
