"""How long the evaluation loop's host waits for the GPU per image (time inside torch.cuda.synchronize + the first blocking read):
near zero = the host is the limiter, near the GPU time = the GPU is."""
import contextlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch import nn
from detectinblur_amd import engine, utils
from detectinblur_amd.coco_utils import SyntheticCocoDetection
from detectinblur_amd.models.blur_estimator import resnet18
from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
from detectinblur_amd.train import get_transform
dev = torch.device("cuda", 0)
torch.manual_seed(1337)
ens = [fasterrcnn_resnet50_fpn(num_classes=91, pretrained=False, pretrained_backbone=False).to(dev).eval() for _ in range(4)]
est = resnet18(); est.fc = nn.Linear(512, 4); est = est.to(dev).eval()
with contextlib.redirect_stdout(sys.stderr):
    tf = get_transform(False, blur=True, blur_type=0.001, blur_ratio=1, blur_exposure=0.5)
N = 60
ds = SyntheticCocoDetection(num_images=N, size=(800, 1333), transforms=tf)
class L(list):
    dataset = ds
batches = L(utils.collate_fn([ds[i]]) for i in range(N))
for b in batches:                      # pinned, as the drivers' loaders deliver them
    b[0][0].data = b[0][0].pin_memory()
kw = dict(blurring_images=True, gpu_blur=True, expand_target_boxes=True, use_ensemble=True, ensemble_models=ens, blur_estimator=est, LEHE=True)
waited = [0.0, 0]
real = torch.cuda.synchronize
def timed(*a, **k):
    t0 = time.perf_counter(); real(*a, **k); waited[0] += time.perf_counter() - t0; waited[1] += 1
with contextlib.redirect_stdout(sys.stderr):
    engine.evaluate(None, L(batches[:16]), dev, **kw)
    torch.cuda.synchronize = timed
    real(); t0 = time.perf_counter()
    engine.evaluate(None, batches, dev, **kw)
    real(); wall = time.perf_counter() - t0
    torch.cuda.synchronize = real
print("wall per image %.2f ms; inside the loop's synchronize: %.2f ms per image (%d calls)" % (wall / N * 1e3, waited[0] / N * 1e3, waited[1]))
