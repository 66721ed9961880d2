"""Training driver with the reference's hot-path flags (reference train.py:393-483):

    python -m torch.distributed.run --nproc-per-node=8 --master-addr 127.0.0.1 -m detectinblur_amd.train \
        --synthetic --blur_train --gpu_blur --param_index 1 --low_exposure --expand_target_boxes -b 8

One process per GPU, DistributedDataParallel over RCCL (backend "nccl" on ROCm), DistributedSampler,
SGD(lr 0.04, momentum 0.9, wd 1e-4) + MultiStepLR, seeds rank*1337, per-epoch checkpoints
{'model','optimizer','lr_scheduler','args','epoch'} and --resume / --start_from_weights
(reference train.py:89-391), TensorBoard scalars under --tensorboard_path.  The reference's README command
lines parse unchanged; flags of subsystems outside the built path (AugMix, deblur-first, custom BN, real-blur
datasets) are accepted and refused with a clear message only when set.
"""
import argparse
import datetime
import os
import random
import time

import numpy as np
import torch
import torch.utils.data

from . import transforms as T
from . import utils
from .coco_utils import get_coco
from .engine import evaluate, train_one_epoch
from .models.faster_rcnn import fasterrcnn_resnet50_fpn


def get_transform(train, blur=False, blur_type=None, blur_ratio=0.5, use_stored_psfs=False, cpu_blur=False,
                  stored_psf_directory=None, dont_center_psf=False, low_exposure=False, high_exposure=False,
                  blur_exposure=None, stored_psf_count=T.STORED_PSF_COUNT, LEHE_blur_seg=False, dilate_psf=False):
    """reference train.py:48-86: [BlurImage] -> ToTensor -> [RandomHorizontalFlip(0.5) when training]."""
    tf = []
    if blur:
        tf.append(T.BlurImage(prob=blur_ratio, blur_type=blur_type, blur_exposure=blur_exposure, use_stored_psfs=use_stored_psfs,
                              stored_psf_directory=stored_psf_directory, blur_image_in_transform=cpu_blur,
                              dont_center_psf=dont_center_psf, low_exposure=low_exposure, high_exposure=high_exposure,
                              stored_psf_count=stored_psf_count, LEHE_blur_seg=LEHE_blur_seg, dilate_psf=dilate_psf))
    tf.append(T.ToTensor())
    if train:
        tf.append(T.RandomHorizontalFlip(0.5))
    return T.Compose(tf)


def seed_everything(distributed):
    """reference train.py:93-107 (rank 0 of a distributed run seeds with 0 -- reproduced, not fixed)."""
    s = torch.distributed.get_rank() * 1337 if distributed else 1337
    np.random.seed(s)
    random.seed(s)
    torch.manual_seed(s)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(1337)


# Flags of subsystems SURVEY.md section 2 places outside the built path: they parse (the reference's command
# lines are accepted unchanged) and raise only when actually set.
_OUT_OF_SCOPE = {"deblur_first": "the deblur-first pipeline (DeepDeblur)", "non_pos_aug_mix": "AugMix",
                 "include_pos_aug_mix": "AugMix", "aug_mix_target_expand": "AugMix",
                 "unfrozen_batch_norm": "trainable batch-norm conversion", "mode_one_norm": "the custom BatchNorm remedy",
                 "blurred_dataset": "real-blur datasets (GOPRO / REDS)", "expand_synth_boxes": "real-blur datasets (GOPRO / REDS)"}


def reject_out_of_scope(args):
    for name, what in _OUT_OF_SCOPE.items():
        if getattr(args, name, False):
            raise SystemExit("--%s: %s is outside the built hot path (SURVEY.md section 2)" % (name, what))
    if "coco" not in args.dataset:
        raise SystemExit("--dataset %s: only COCO (and --synthetic) is built; real-blur datasets are outside the hot path" % args.dataset)
    if "fasterrcnn_resnet50_fpn" not in args.model:
        raise SystemExit("--model %s: only fasterrcnn_resnet50_fpn is built (SURVEY.md section 2)" % args.model)


def add_shared_flags(p):
    """Flags train.py and evaluate.py share (reference train.py:399-478, evaluate.py:384-466), same names,
    defaults and help texts' meaning; `--synthetic*` / `--stored_psf_count` / `--min_size` / `--max_size` are this repo's additions."""
    p.add_argument("--dataset", default="coco", help="dataset")
    p.add_argument("--data_path", default=None, help="COCO root (train2017/, val2017/, annotations/)")
    p.add_argument("--synthetic", action="store_true", help="COCO-shaped synthetic data (no dataset on disk needed)")
    p.add_argument("--synthetic_images", default=64, type=int)
    p.add_argument("--with_masks", action="store_true", help="rasterise the objects' segmentation masks into the targets as the reference's loader does "
                   "(Faster R-CNN never reads them: off by default, which spares the loader N full-resolution masks per image and the model their resize)")
    p.add_argument("--synthetic_size", default=[800, 1333], nargs=2, type=int)
    p.add_argument("--min_size", default=None, type=int, help="(this repo) FasterRCNN(min_size=), reference default 800 (models/faster_rcnn.py:148)")
    p.add_argument("--max_size", default=None, type=int, help="(this repo) FasterRCNN(max_size=), reference default 1333")
    p.add_argument("--use_stored_psfs", action="store_true", help="Use stored PSFs when blurring in the data loader.")
    p.add_argument("--stored_psf_directory", default=None, help="Stored PSFs path.")
    p.add_argument("--stored_psf_count", default=T.STORED_PSF_COUNT, type=int)
    p.add_argument("-j", "--workers", default=0, type=int, metavar="N", help="number of data loading workers (default: 0)")
    p.add_argument("--model", default="fasterrcnn_resnet50_fpn", help="model")
    p.add_argument("--trainable_backbone_blocks", default=3, type=int, help="Resnet backbone blocks to train.")
    p.add_argument("--pretrained", action="store_true", help="Use pre-trained models from the modelzoo (a locally cached file).")
    p.add_argument("--device", default="cuda", help="device")
    p.add_argument("--resume", default=None, help="resume from checkpoint")
    p.add_argument("--early_stop", type=int, default=None, help="early stop for eval")
    p.add_argument("--tensorboard_path", default="debug", help="directory of the TensorBoard event file")
    p.add_argument("--output_dir", default="debug", help="Output directory for weights.")
    p.add_argument("--image_output_dir", default="debug", help="Output directory for images.")
    p.add_argument("--cpu_blur", action="store_true", help="CPU blurring in the Fourier domain, in the data loader's workers.")
    p.add_argument("--gpu_blur", action="store_true", help="GPU blurring, on the GPU in the training thread.")
    p.add_argument("--param_index", default=None, help="Type of blur. Options are 1, 2, and 3.")
    p.add_argument("--high_exposure", action="store_true", help="Train and evaluate with high exposure blur.")
    p.add_argument("--low_exposure", action="store_true", help="Train and evaluate with low exposure blur.")
    p.add_argument("--expand_target_boxes", action="store_true", help="Expand target boxes according to blur kernel shifts.")
    p.add_argument("--dont_center_psf", action="store_true", help="Don't center PSFs (on-the-fly PSFs only).")
    p.add_argument("--add_noise", action="store_true", help="Add noise after blurring.")
    p.add_argument("--noise_level", default=0.001, type=float, help="Noise level.")
    p.add_argument("--add_block", action="store_true", help="Add block artifacts after blurring.")
    p.add_argument("--add_jpeg_artefacts", action="store_true", help="Add jpeg compression artifacts.")
    p.add_argument("--warp_in_model", action="store_true", help="Warp and dewarp images before and after backbone.")
    p.add_argument("--use_custom_image_norm", action="store_true", help="Use blur specific normalization on input to network.")
    # outside the built path: accepted, refused when set (reject_out_of_scope)
    p.add_argument("--deblur_first", action="store_true", help="(not built) deblur before detecting")
    p.add_argument("--deblurer_model_location", default=None, help="(not built)")
    p.add_argument("--non_pos_aug_mix", action="store_true", help="(not built) AugMix")
    p.add_argument("--include_pos_aug_mix", action="store_true", help="(not built) AugMix")
    p.add_argument("--aug_mix_target_expand", action="store_true", help="(not built) AugMix")
    p.add_argument("--unfrozen_batch_norm", action="store_true", help="(not built)")
    p.add_argument("--world-size", default=1, type=int, help="number of distributed processes")
    p.add_argument("--dist-url", default="env://", help="url used to set up distributed training")
    return p


def detector_size_kwargs(args):
    """--min_size / --max_size as FasterRCNN keyword arguments (absent: the reference's 800 / 1333)."""
    return {k: getattr(args, k) for k in ("min_size", "max_size") if getattr(args, k, None) is not None}


def build_parser():
    p = add_shared_flags(argparse.ArgumentParser(description="detectInBlur hot path on MI355X: training"))
    p.add_argument("--aspect-ratio-group-factor", default=3, type=int)
    p.add_argument("-b", "--batch_size", default=8, type=int, help="images per gpu, the total batch size is $NGPU x batch_size")
    p.add_argument("--lr", default=0.04, type=float, help="initial learning rate")
    p.add_argument("--lr-step-size", default=8, type=int, help="decrease lr every step-size epochs (unused, as in the reference)")
    p.add_argument("--lr-steps", default=[16, 22], nargs="+", type=int, help="epochs at which the lr drops")
    p.add_argument("--lr-gamma", default=0.1, type=float, help="decrease lr by a factor of lr-gamma")
    p.add_argument("--epochs", default=37, type=int, metavar="N", help="number of total epochs to run")
    p.add_argument("--momentum", default=0.9, type=float, metavar="M", help="momentum")
    p.add_argument("--weight_decay", default=1e-4, type=float, metavar="W", help="weight decay (default: 1e-4)")
    p.add_argument("--foreach_sgd", action="store_true", help="torch's default (foreach) SGD kernels instead of the fused ones (utils.make_sgd)")
    p.add_argument("--start_from_weights", default=None, help="start training from provided weights")
    p.add_argument("--start_epoch", default=0, type=int, help="Custom start epoch.")
    p.add_argument("--eval_first", action="store_true", help="Evaluate first before training.")
    p.add_argument("--print_freq", default=20, type=int, help="print frequency")
    p.add_argument("--blur_train", action="store_true", help="Blur during training.")
    return p


_TB_STATS = (("AccuraciesSweep", 0), ("Accuracies", 1), ("AccuraciesSmall", 3), ("AccuraciesMedium", 4), ("AccuraciesLarge", 5),
             ("recallSmall", 9), ("recallMedium", 10), ("recallLarge", 11), ("recall", 12))


def log_coco_stats(writer, prefix, coco_evaluator, step):
    """reference train.py:350-387 / evaluate.py:249-259.  The reference reads stats[12] for "recall", one past
    COCOeval's 12 numbers; AR@100 (stats[8]) is logged under that tag here."""
    if writer is None:
        return
    stats = coco_evaluator.coco_eval["bbox"].stats
    for tag, k in _TB_STATS:
        writer.add_scalar(prefix + "/" + tag, float(stats[k] if k < len(stats) else stats[8]), step)


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
    if utils.is_main_process() and args.tensorboard_path:               # reference train.py:109-120
        from .tb_writer import make_writer
        writer = make_writer(args.tensorboard_path)

    if args.use_stored_psfs:                                            # reference train.py:127-137
        blur_type = None if args.param_index is None else int(args.param_index)
    else:
        blur_type = None if args.param_index is None else [0.01, 0.005, 0.001, 0.00005][int(args.param_index)]
    blur_ratio = 0.75 if args.low_exposure else (1 if args.high_exposure else 0.9)   # :139-144
    synthetic = dict(num_images=args.synthetic_images, size=tuple(args.synthetic_size), as_tensor=not args.cpu_blur) if args.synthetic else None
    common = dict(use_stored_psfs=args.use_stored_psfs, cpu_blur=args.cpu_blur,
                  stored_psf_directory=args.stored_psf_directory, dont_center_psf=args.dont_center_psf,
                  high_exposure=args.high_exposure, low_exposure=args.low_exposure, stored_psf_count=args.stored_psf_count)
    dataset, num_classes = get_coco(args.data_path, "train", get_transform(True, blur=args.blur_train, blur_type=blur_type,
                                                                           blur_ratio=blur_ratio, **common), synthetic=synthetic, with_masks=args.with_masks)
    dataset_test, _ = get_coco(args.data_path, "val", get_transform(False, blur=False), synthetic=synthetic, with_masks=args.with_masks)
    eval_blur_type = blur_type if (args.high_exposure and not args.low_exposure) else None        # :164-169
    dataset_test_blur, _ = get_coco(args.data_path, "val", get_transform(False, blur=True, blur_ratio=1, blur_type=eval_blur_type,
                                                                         **common), synthetic=synthetic, with_masks=args.with_masks)

    if args.distributed:
        train_sampler = torch.utils.data.distributed.DistributedSampler(dataset)
        test_sampler = torch.utils.data.distributed.DistributedSampler(dataset_test)
        test_sampler_blur = torch.utils.data.distributed.DistributedSampler(dataset_test_blur)
    else:
        train_sampler = torch.utils.data.RandomSampler(dataset)
        test_sampler = torch.utils.data.SequentialSampler(dataset_test)
        test_sampler_blur = torch.utils.data.SequentialSampler(dataset_test_blur)
    if args.aspect_ratio_group_factor >= 0:                             # :192-198
        from .group_by_aspect_ratio import GroupedBatchSampler, create_aspect_ratio_groups
        batch_sampler = GroupedBatchSampler(train_sampler, create_aspect_ratio_groups(dataset, k=args.aspect_ratio_group_factor),
                                            args.batch_size)
    else:
        batch_sampler = torch.utils.data.BatchSampler(train_sampler, args.batch_size, drop_last=True)
    pin = device.type == "cuda"
    data_loader = torch.utils.data.DataLoader(dataset, batch_sampler=batch_sampler, num_workers=args.workers, collate_fn=utils.collate_fn,
                                              pin_memory=pin, worker_init_fn=_seed_worker, multiprocessing_context=mp_ctx)
    mk_test = lambda ds, sm: torch.utils.data.DataLoader(ds, batch_size=1, sampler=sm, num_workers=args.workers,  # noqa: E731
                                                         collate_fn=utils.collate_fn, pin_memory=pin, worker_init_fn=_seed_worker,
                                                         multiprocessing_context=mp_ctx)
    data_loader_test, data_loader_test_blur = mk_test(dataset_test, test_sampler), mk_test(dataset_test_blur, test_sampler_blur)

    print("Creating model")
    model = fasterrcnn_resnet50_fpn(num_classes=num_classes, pretrained=args.pretrained,
                                    pretrained_backbone=False if args.synthetic else "auto",
                                    trainable_backbone_layers=args.trainable_backbone_blocks,
                                    warp_internally=args.warp_in_model, **detector_size_kwargs(args))
    model.to(device)
    model_without_ddp = model
    if args.distributed:
        # the buffers are frozen batch-norm statistics: equal on all ranks after DDP's construction-time sync,
        # so the per-forward re-broadcast (~7 ms per step) is switched off
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[args.gpu] if device.type == "cuda" else None,
                                                          broadcast_buffers=False, gradient_as_bucket_view=True)
        model_without_ddp = model.module
    params = [p for p in model.parameters() if p.requires_grad]
    optimizer = utils.make_sgd(params, args.lr, args.momentum, args.weight_decay, foreach=args.foreach_sgd)
    lr_scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=args.lr_steps, gamma=args.lr_gamma)

    if args.resume:                                                     # reference train.py:251-257
        print("Resuming training from from " + args.resume)
        ck = torch.load(args.resume, map_location="cpu", weights_only=False)
        model_without_ddp.load_state_dict(ck["model"])
        optimizer.load_state_dict(ck["optimizer"])
        utils.restore_sgd_implementation(optimizer, getattr(args, "foreach_sgd", False))
        lr_scheduler.load_state_dict(ck["lr_scheduler"])
        args.start_epoch = ck["epoch"] + 1
    if args.start_from_weights:                                         # :260-263
        print("Using model weights from " + args.start_from_weights)
        model_without_ddp.load_state_dict(torch.load(args.start_from_weights, map_location="cpu", weights_only=False)["model"])

    blur_eval_kw = dict(device=device, early_stop=args.early_stop, distributed_mode=args.distributed, blurring_images=True,
                        gpu_blur=args.gpu_blur, expand_target_boxes=args.expand_target_boxes,
                        use_custom_image_norm=args.use_custom_image_norm, add_noise=args.add_noise, noise_level=args.noise_level,
                        add_block=args.add_block, add_jpeg_artifact=args.add_jpeg_artefacts)
    # the clean pass takes none of the blur options (reference train.py:346-349)
    clean_kw = dict(device=device, distributed_mode=args.distributed, early_stop=args.early_stop, vanilla_eval=True)
    if args.eval_first:                                                 # :272-289
        evaluate(model, data_loader_test_blur, **blur_eval_kw)
        evaluate(model, data_loader_test, **clean_kw)

    print("Starting training.")
    start = time.time()
    for epoch in range(args.start_epoch, args.epochs):
        base = dataset.dataset if isinstance(dataset, torch.utils.data.Subset) else dataset
        base._epoch_number = epoch                                      # :294
        if args.distributed:
            train_sampler.set_epoch(epoch)
        train_one_epoch(model, optimizer, data_loader, device, epoch, args.print_freq, writer, args.distributed, args.blur_train,
                        args.early_stop, args.gpu_blur, args.expand_target_boxes, args.use_custom_image_norm, args.add_noise,
                        args.noise_level, args.add_block, args.add_jpeg_artefacts)
        lr_scheduler.step()
        if args.output_dir:
            utils.mkdir(args.output_dir)
            utils.save_on_master({"model": model_without_ddp.state_dict(), "optimizer": optimizer.state_dict(),
                                  "lr_scheduler": lr_scheduler.state_dict(), "args": args, "epoch": epoch},
                                 os.path.join(args.output_dir, "model_{}.pth".format(epoch)))
        ce = evaluate(model, data_loader_test, **clean_kw)              # :345-361
        if utils.is_main_process():
            log_coco_stats(writer, "Normal", ce, epoch)
        ce = evaluate(model, data_loader_test_blur, **blur_eval_kw)     # :363-387
        if utils.is_main_process():
            log_coco_stats(writer, "Blurred", ce, epoch)
            if writer is not None:
                writer.flush()
    if writer is not None:
        writer.close()
    print("Training time {}".format(str(datetime.timedelta(seconds=int(time.time() - start)))))


def _seed_worker(worker_id):
    """Every DataLoader worker gets its own numpy / random stream (the reference passes no
    worker_init_fn, so with the torch of its era all workers replayed one numpy stream: SURVEY A.9)."""
    s = torch.initial_seed() % 2 ** 31
    np.random.seed(s)
    random.seed(s)


if __name__ == "__main__":
    main(build_parser().parse_args())
