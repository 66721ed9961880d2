"""Eager trunk vs captured trunk graph at b = 1, 800 x 1333, in a fresh process (is the captured graph made of the kernels the eager
path settles on?).  WARM=n: n eager forward passes before graph_inference is switched on."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
torch.manual_seed(0)
m = fasterrcnn_resnet50_fpn(num_classes=91, pretrained=False, pretrained_backbone=False).cuda().eval()
img = torch.rand(3, 800, 1333, device="cuda")
means, stds = np.tile([0.485, 0.456, 0.406], (1, 1)), np.tile([0.229, 0.224, 0.225], (1, 1))


def t(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    f = lambda: m([img], newMeans=means, newSTDs=stds)
    for k in range(int(os.environ.get("WARM", "0"))):
        print("eager forward %d: %.1f ms" % (k, t(f, 1)))
    m.graph_inference = True
    print("first graphed forward (warm-up + capture + replay): %.1f ms" % t(f, 1))
    print("graphed forward: %.2f ms" % t(f, int(os.environ.get("N", "20"))))
    g = list(m._trunk_graphs.graphs.values())[0]
    x = g.static_in.clone()
    print("trunk graph replay: %.2f ms" % t(lambda: g(x), int(os.environ.get("N", "20"))))
    m.graph_inference = False
    print("eager forward: %.2f ms" % t(f, 5))
    imgs, _ = m.transform([img], None, means, stds)
    print("eager trunk: %.2f ms" % t(lambda: m._trunk(imgs.tensors), 5))
