"""Evaluation driver with the reference's hot-path flags (reference evaluate.py:378-468): one detector,
or the 4-detector ensemble routed per image by a ResNet-18 blur estimator (`--use_ensemble`, `--LEHE`)
or by the ground-truth blur_dict, over the blur sweep P in {0.005, 0.001, 0.00005} x E in
{1/25, 1/10, 1/5, 1/2, 1} (reference evaluate.py:299-370).  Image-parallel across ranks with a
DistributedSampler, batch size 1 per rank as in the reference.  Random-initialised models stand in for
checkpoints when no paths are given (synthetic throughput runs)."""
import argparse

import torch
import torch.utils.data
from torch import nn

from . import utils
from .coco_utils import get_coco
from .engine import evaluate
from .models.blur_estimator import resnet18
from .models.faster_rcnn import fasterrcnn_resnet50_fpn
from .train import _seed_worker, get_transform, seed_everything

SWEEP_PARAMS = [0.005, 0.001, 0.00005]
SWEEP_FRACTIONS = [1 / 25, 1 / 10, 1 / 5, 1 / 2, 1]


def build_parser():
    p = argparse.ArgumentParser(description="detectInBlur hot path on MI355X: evaluation")
    p.add_argument("--data_path", default=None)
    p.add_argument("--synthetic", action="store_true")
    p.add_argument("--synthetic_images", default=32, type=int)
    p.add_argument("--synthetic_size", default=[800, 1333], nargs=2, type=int)
    p.add_argument("-j", "--workers", default=0, type=int)
    p.add_argument("--device", default="cuda")
    p.add_argument("--model_path", default=None)
    p.add_argument("--use_ensemble", action="store_true")
    p.add_argument("--ensemble_model_paths", default=None, nargs="+")
    p.add_argument("--blur_estimator_path", default=None)
    p.add_argument("--use_blur_estimator", action="store_true")
    p.add_argument("--LEHE", action="store_true")
    p.add_argument("--blur_eval", action="store_true", help="parsed for compatibility; the sweep always blurs (reference quirk)")
    p.add_argument("--gpu_blur", action="store_true")
    p.add_argument("--expand_target_boxes", action="store_true")
    p.add_argument("--use_custom_image_norm", action="store_true")
    p.add_argument("--add_noise", action="store_true")
    p.add_argument("--noise_level", default=0.001, type=float)
    p.add_argument("--add_block", action="store_true")
    p.add_argument("--add_jpeg_artefacts", action="store_true", help="Add jpeg compression artifacts.")
    p.add_argument("--warp_in_model", action="store_true", help="Warp and dewarp images before and after backbone.")
    p.add_argument("--early_stop", type=int, default=None)
    p.add_argument("--world-size", default=1, type=int)
    p.add_argument("--dist-url", default="env://")
    return p


def _load(model, path):
    if path:
        model.load_state_dict(torch.load(path, map_location="cpu", weights_only=False)["model"])
    return model


def main(args):
    utils.init_distributed_mode(args)
    seed_everything(args.distributed)
    device = torch.device(args.device if torch.cuda.is_available() or args.device == "cpu" else "cpu")
    synthetic = dict(num_images=args.synthetic_images, size=tuple(args.synthetic_size)) if args.synthetic else None

    def detector(path=None):
        m = _load(fasterrcnn_resnet50_fpn(num_classes=91, pretrained=False, pretrained_backbone=False,
                                          warp_internally=args.warp_in_model), path).to(device)
        if not args.distributed:
            return m
        return torch.nn.parallel.DistributedDataParallel(m, device_ids=[args.gpu] if device.type == "cuda" else None,
                                                         broadcast_buffers=False)

    ensemble, estimator, model = None, None, None
    if args.use_ensemble:                                               # reference evaluate.py:159-205
        paths = args.ensemble_model_paths or [None] * 4
        ensemble = [detector(p) for p in paths]
        if args.use_blur_estimator or args.blur_estimator_path:
            estimator = resnet18()
            estimator.fc = nn.Linear(512, 4 if args.LEHE else 16)
            estimator = _load(estimator, args.blur_estimator_path).to(device)
    else:
        model = detector(args.model_path)

    results = {}
    for p_i, param in enumerate(SWEEP_PARAMS):
        for f_i, fraction in enumerate(SWEEP_FRACTIONS):
            tf = get_transform(False, blur=True, blur_type=param, blur_ratio=1, blur_exposure=fraction)
            ds, _ = get_coco(args.data_path, "val", tf, synthetic=synthetic)
            sampler = torch.utils.data.distributed.DistributedSampler(ds, shuffle=False) if args.distributed else torch.utils.data.SequentialSampler(ds)
            loader = torch.utils.data.DataLoader(ds, batch_size=1, sampler=sampler, num_workers=args.workers, collate_fn=utils.collate_fn,
                                                 pin_memory=device.type == "cuda", worker_init_fn=_seed_worker)
            out = evaluate(model, loader, device=device, distributed_mode=args.distributed, early_stop=args.early_stop,
                           blurring_images=True, gpu_blur=args.gpu_blur, expand_target_boxes=args.expand_target_boxes,
                           use_custom_image_norm=args.use_custom_image_norm, use_ensemble=args.use_ensemble, ensemble_models=ensemble,
                           blur_estimator=estimator, LEHE=args.LEHE, add_noise=args.add_noise, noise_level=args.noise_level,
                           add_block=args.add_block, add_jpeg_artifact=args.add_jpeg_artefacts)
            results["P%dE%d" % (p_i + 1, f_i)] = out
            print("P%d E%d: %d images, routes %s" % (p_i + 1, f_i, len(out["detections"]), out["routes"][:8]))
    return results


if __name__ == "__main__":
    main(build_parser().parse_args())
