"""Evaluation driver with the reference's hot-path flags (reference evaluate.py:378-468): one detector,
or the 4-detector ensemble routed per image by a ResNet-18 blur estimator (`--use_ensemble`, `--LEHE`)
or by the ground-truth blur_dict, over the blur sweep P in {0.005, 0.001, 0.00005} x E in
{1/25, 1/10, 1/5, 1/2, 1} (reference evaluate.py:299-370).  Image-parallel across ranks with a
DistributedSampler, batch size 1 per rank as in the reference.  `--vanilla_eval` scores clean images once
(reference :226-262).  Every pass returns the reference's CocoEvaluator surface and logs its statistics to
TensorBoard under the reference's tags.  Random-initialised models stand in for checkpoints when no paths are
given (synthetic throughput runs).  The reference README's command lines parse unchanged."""
import argparse

import torch
import torch.utils.data
from torch import nn

from . import utils
from .coco_utils import get_coco
from .engine import evaluate
from .models.blur_estimator import resnet18
from .models.faster_rcnn import fasterrcnn_resnet50_fpn
from .train import _seed_worker, add_shared_flags, detector_size_kwargs, get_transform, log_coco_stats, reject_out_of_scope, seed_everything

SWEEP_PARAMS = [0.005, 0.001, 0.00005]
SWEEP_FRACTIONS = [1 / 25, 1 / 10, 1 / 5, 1 / 2, 1]


def build_parser():
    """reference evaluate.py:378-466, flag for flag (shared ones in train.add_shared_flags)."""
    p = add_shared_flags(argparse.ArgumentParser(description="detectInBlur hot path on MI355X: evaluation"))
    p.set_defaults(synthetic_images=32)
    p.add_argument("--use_ensemble", action="store_true", help="Use blur network system ensemble.")
    p.add_argument("--ensemble_model_paths", default=None, nargs="+", help="Ensemble model paths (one quoted string or several).")
    p.add_argument("--blur_estimator_path", default=None, help="Blur estimator model path.")
    p.add_argument("--use_blur_estimator", action="store_true", help="(this repo) route with a random-init estimator when no path is given")
    p.add_argument("--vanilla_eval", action="store_true", help="Vanilla eval on clean COCO images.")
    p.add_argument("--blur_eval", action="store_true", help="Blur during evaluation (the sweep always blurs, as in the reference).")
    p.add_argument("--LEHE", action="store_true", help="System with low and high exposure networks.")
    p.add_argument("--dilate_psf", action="store_true", help="Dilate PSF to simulate defocus with motion blur.")
    p.add_argument("--model_path", default=None, help="(this repo) alias of --resume")
    # outside the built path: accepted, refused when set
    p.add_argument("--blurred_dataset", action="store_true", help="(not built) real-blur datasets")
    p.add_argument("--expand_synth_boxes", action="store_true", help="(not built) real-blur datasets")
    p.add_argument("--mode_one_norm", action="store_true", help="(not built) batch-norm remedy")
    return p


def _load(model, path):
    if path:
        print("Loading from " + path)
        model.load_state_dict(torch.load(path, map_location="cpu", weights_only=False)["model"])
    return model


def main(args):
    from . import kernel_choices
    kernel_choices.use_shipped_kernel_choices()      # shipped MIOpen / TunableOp choices, private copy per process (kernel_choices.py)
    reject_out_of_scope(args)
    mp_ctx = utils.loader_context() if args.workers > 0 else None      # before anything touches the GPU (see utils.loader_context)
    utils.init_distributed_mode(args)
    print(args)
    seed_everything(args.distributed)
    device = torch.device(args.device if torch.cuda.is_available() or args.device == "cpu" else "cpu")
    writer = None
    if utils.is_main_process() and args.tensorboard_path:               # reference evaluate.py:143-151
        from .tb_writer import make_writer
        writer = make_writer(args.tensorboard_path)
    synthetic = dict(num_images=args.synthetic_images, size=tuple(args.synthetic_size), as_tensor=not args.cpu_blur) if args.synthetic else None
    if not (args.gpu_blur or args.cpu_blur or args.vanilla_eval):
        # README: "You'll need to specify --blur_eval and hardware --gpu_blur or --cpu_blur"; without either the
        # reference scores sharp images against the sweep's labels without saying so
        print("Warning: neither --gpu_blur nor --cpu_blur: the sweep will score UNBLURRED images.")

    def detector(path=None):
        m = _load(fasterrcnn_resnet50_fpn(num_classes=91, pretrained=args.pretrained, pretrained_backbone=False,
                                          warp_internally=args.warp_in_model, **detector_size_kwargs(args)), path).to(device)
        if not args.distributed:
            return m
        return torch.nn.parallel.DistributedDataParallel(m, device_ids=[args.gpu] if device.type == "cuda" else None,
                                                         broadcast_buffers=False)

    ensemble, estimator, model = None, None, None
    if args.use_ensemble:                                               # reference evaluate.py:159-205
        # the README passes the four paths as ONE quoted string (evaluate.py:161 splits element 0)
        paths = [q for p_ in (args.ensemble_model_paths or []) for q in p_.split()] or [None] * 4
        ensemble = [detector(p_) for p_ in paths]
        if args.use_blur_estimator or args.blur_estimator_path:
            estimator = resnet18()
            estimator.fc = nn.Linear(512, 4 if args.LEHE else 16)
            estimator = _load(estimator, args.blur_estimator_path).to(device)
    else:
        model = detector(args.resume or args.model_path)

    def loader_for(tf):
        ds, _ = get_coco(args.data_path, "val", tf, synthetic=synthetic, with_masks=getattr(args, "with_masks", False))
        sampler = torch.utils.data.distributed.DistributedSampler(ds) if args.distributed else torch.utils.data.SequentialSampler(ds)
        return torch.utils.data.DataLoader(ds, batch_size=1, sampler=sampler, num_workers=args.workers, collate_fn=utils.collate_fn,
                                           pin_memory=device.type == "cuda", worker_init_fn=_seed_worker, multiprocessing_context=mp_ctx)

    ens_kw = dict(use_ensemble=args.use_ensemble, ensemble_models=ensemble, blur_estimator=estimator, LEHE=args.LEHE)
    results = {}
    if args.vanilla_eval:                                               # reference evaluate.py:226-262
        ce = evaluate(model, loader_for(get_transform(False, blur=False)), device=device, vanilla_eval=True,
                      distributed_mode=args.distributed, use_custom_image_norm=args.use_custom_image_norm,
                      early_stop=args.early_stop, **ens_kw)
        log_coco_stats(writer, "Clean", ce, 0)
        if writer is not None:
            writer.close()
        return {"Clean": ce}

    # reference evaluate.py:272-370: params[0] / fractions[0] are legacy entries the loops skip, so the tags are
    # P1..P3 over E1..E5
    for param_index, param in enumerate(SWEEP_PARAMS, start=1):
        for fraction_index, fraction in enumerate(SWEEP_FRACTIONS, start=1):
            print("################################## P" + str(param_index) + " and E" + str(fraction_index) + " ###################################")
            tf = get_transform(False, blur=True, blur_type=param, blur_ratio=1, blur_exposure=fraction,
                               use_stored_psfs=args.use_stored_psfs, cpu_blur=args.cpu_blur,
                               stored_psf_directory=args.stored_psf_directory, dont_center_psf=args.dont_center_psf,
                               stored_psf_count=args.stored_psf_count, dilate_psf=args.dilate_psf)
            out = evaluate(model, loader_for(tf), device=device, distributed_mode=args.distributed, early_stop=args.early_stop,
                           blurring_images=True, gpu_blur=args.gpu_blur, expand_target_boxes=args.expand_target_boxes,
                           use_custom_image_norm=args.use_custom_image_norm, add_noise=args.add_noise, noise_level=args.noise_level,
                           add_block=args.add_block, add_jpeg_artifact=args.add_jpeg_artefacts, **ens_kw)
            results["P%dE%d" % (param_index, fraction_index - 1)] = out
            if utils.is_main_process():
                log_coco_stats(writer, "P" + str(param_index), out, fraction_index)
            print("P%d E%d: %d images, routes %s" % (param_index, fraction_index - 1, len(out["detections"]), out["routes"][:8]))
    if writer is not None:
        writer.close()
    return results


if __name__ == "__main__":
    main(build_parser().parse_args())
