"""Host-side phases of the per-image evaluation loop (configs[4], batch 1, in-memory batches), each closed by a device synchronisation:
where the ~3 ms per image go in which the GPU waits for the host."""
import contextlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch import nn
from detectinblur_amd import engine, utils
from detectinblur_amd.coco_utils import SyntheticCocoDetection
from detectinblur_amd.models import blur_functions, net_transforms
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
batches = [utils.collate_fn([ds[i]]) for i in range(N)]
for m in ens:
    m.graph_inference = True
batcher = net_transforms.GeneralizedRCNNTransform(800, 1333, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], crop_images=True)
acc = {}
def phase(name, t0):
    torch.cuda.synchronize()
    t = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t - t0)
    return t
with torch.no_grad():
    for rep in range(2):
        acc.clear()
        for images_CPU, targets_CPU, blur_dicts in batches:
            torch.cuda.synchronize(); t = time.perf_counter()
            images_GPU, targets_GPU, psfs_GPU, thetas, l1, l2, tables = engine._to_device(images_CPU, targets_CPU, blur_dicts, dev, True, want_tables=True)
            t = phase("1 _to_device (H2D image, PSF block, tap tables)", t)
            blur_functions.blur_image_list(images_GPU, blur_dicts, psfs_GPU=psfs_GPU, tables=tables)
            t = phase("2 blur", t)
            targets_GPU = utils.expand_targets(targets_GPU, blur_dicts, psfs_GPU, images_GPU, tables=engine._tables_128(tables))
            boxes = [utils.convert_to_xywh(tg["boxes"]).cpu().numpy().tolist() for tg in targets_GPU]
            ids = [int(tg["image_id"].item()) for tg in targets_GPU]
            t = phase("3 expand_targets + boxes to host", t)
            images_GPU = engine._to_float(images_GPU, ens[0], dev)
            means, stds = utils.get_norm_params(blur_dicts, False)
            t = phase("4 _to_float + norm params", t)
            batched, _ = batcher(images_GPU, None)
            e = engine._estimate(est, batched.tensors, True)
            k = engine.get_network_index_to_use_blur_estimator_LEHE(e, [0, 1, 2, 3])
            t = phase("5 estimator + route", t)
            out = ens[k](images_GPU, thetas=thetas, lambda1s=l1, lambda2s=l2, newMeans=means, newSTDs=stds)
            t = phase("6 detector forward", t)
            out = [{kk: v.to("cpu") for kk, v in o.items()} for o in out]
            gt = [utils.convert_to_xywh(tg["boxes"]).cpu() for tg in targets_GPU]
            t = phase("7 outputs + gt to host", t)
    tot = sum(acc.values())
    for k2 in sorted(acc):
        print("%-52s %7.3f ms per image" % (k2, acc[k2] / N * 1e3))
    print("%-52s %7.3f ms per image (every phase synchronised)" % ("sum", tot / N * 1e3))
