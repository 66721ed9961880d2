"""cProfile of the main thread of engine.evaluate (configs[4], batch 1, in-memory batches): where the host's 11 ms per image go."""
import contextlib, cProfile, os, pstats, sys, time
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
N = 40
ds = SyntheticCocoDetection(num_images=N, size=(800, 1333), transforms=tf)
class L(list):
    dataset = ds
batches = L(utils.collate_fn([ds[i]]) for i in range(N))
kw = dict(blurring_images=True, gpu_blur=True, expand_target_boxes=True, use_ensemble=True, ensemble_models=ens, blur_estimator=est, LEHE=True)
with contextlib.redirect_stdout(sys.stderr):
    engine.evaluate(None, L(batches[:16]), dev, **kw)
    pr = cProfile.Profile()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pr.enable()
    engine.evaluate(None, batches, dev, **kw)
    pr.disable()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / N * 1e3
print("wall per image under cProfile: %.2f ms" % wall)
st = pstats.Stats(pr); st.sort_stats("cumulative")
import io
buf = io.StringIO(); st.stream = buf; st.print_stats(45); print(buf.getvalue()[:9000])
