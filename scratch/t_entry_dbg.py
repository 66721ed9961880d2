import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from detectinblur_amd.models import backbone as B
torch.manual_seed(2)
body = B.ResNet50Body().cuda().to(memory_format=torch.channels_last)
for mod in body.modules():
    if isinstance(mod, B.FrozenBatchNorm2d):
        mod.weight.uniform_(0.5, 1.5); mod.bias.uniform_(-.2, .2); mod.running_mean.uniform_(-.2, .2); mod.running_var.uniform_(0.5, 1.5)
x0 = torch.randn(2, 512, 20, 28, device="cuda").contiguous(memory_format=torch.channels_last)


def run(flag):
    B.BLOCK_ENTRY = flag
    grads = {}
    x = x0.clone().requires_grad_(True)
    y = x
    outs = []
    for i, blk in enumerate(body.layer3):
        y = blk(y)
        outs.append(y)
        y.register_hook(lambda g, i=i: grads.__setitem__(i, g.detach().clone()))
    for p in body.parameters():
        p.grad = None
    y.square().mean().backward()
    return grads, [o.detach().clone() for o in outs], x.grad.clone(), [p.grad.clone() for p in body.layer3.parameters()]


for f in (True, False):
    run(f)          # warm
seq = [run(f) for f in (True, False, True, False)]
print("forward equal T1/F1/T2/F2:", [torch.equal(seq[0][1][5], s[1][5]) for s in seq])
for name, a, b in (("T1 vs T2", seq[0], seq[2]), ("F1 vs F2", seq[1], seq[3]), ("T1 vs F1", seq[0], seq[1])):
    print(name, "x.grad rel %.2e" % float((a[2] - b[2]).abs().max() / b[2].abs().max()),
          "weights rel max %.2e" % max(float((p - q).abs().max() / q.abs().max()) for p, q in zip(a[3], b[3])))
T, F = seq[0], seq[1]
for i in range(5):     # grads at y3.i: fused arrives masked; plain arrives unmasked -> mask it with the plain run's output
    g_f = F[0][i] * (F[1][i] > 0)
    print("y3.%d: masked plain vs fused max|d| %.3e of %.3e; mask mismatches %d" % (i, float((T[0][i] - g_f).abs().max()), float(g_f.abs().max()), int(((T[1][i] > 0) != (F[1][i] > 0)).sum())))
