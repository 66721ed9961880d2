"""torch.optim.SGD(fused=True) vs the default (foreach) on the detector's trainable parameters: time per step and the difference."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
torch.manual_seed(0)
dev = torch.device("cuda", 0)
models = [fasterrcnn_resnet50_fpn(pretrained=False, pretrained_backbone=False, num_classes=91).to(dev) for _ in range(2)]
models[1].load_state_dict(models[0].state_dict())
opts = []
for m, kw in zip(models, ({}, {"fused": True})):
    ps = [p for p in m.parameters() if p.requires_grad]
    opts.append((ps, torch.optim.SGD(ps, lr=0.02, momentum=0.9, weight_decay=1e-4, **kw)))
g = torch.Generator(device=dev).manual_seed(1)
for step in range(4):
    grads = [torch.randn_like(p) * 0.01 for p in opts[0][0]]
    for ps, opt in opts:
        for p, gr in zip(ps, grads):
            p.grad = gr.clone(memory_format=torch.preserve_format)
        opt.step()
    worst = max(float(((a - b).abs().max() / a.abs().max().clamp(min=1e-12))) for a, b in zip(opts[0][0], opts[1][0]))
    exact = sum(int(torch.equal(a, b)) for a, b in zip(opts[0][0], opts[1][0]))
    print("step %d: %d / %d tensors bit-equal, worst |diff| / max|w| = %.2e" % (step, exact, len(opts[0][0]), worst), flush=True)
for name, (ps, opt) in zip(("foreach (default)", "fused"), opts):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        opt.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-18s host %.1f us, wall %.1f us per step" % (name, (t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6), flush=True)
