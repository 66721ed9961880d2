"""Training driver with the reference's hot-path flags (reference train.py:393-483):

    python -m torch.distributed.run --nproc-per-node=8 --master-addr 127.0.0.1 -m detectinblur_amd.train \
        --synthetic --blur_train --gpu_blur --param_index 1 --low_exposure --expand_target_boxes -b 8

One process per GPU, DistributedDataParallel over RCCL (backend "nccl" on ROCm), DistributedSampler,
SGD(lr 0.04, momentum 0.9, wd 1e-4) + MultiStepLR, seeds rank*1337, per-epoch checkpoints
{'model','optimizer','lr_scheduler','args','epoch'} and --resume / --start_from_weights
(reference train.py:89-391).  Flags of subsystems outside the built path (AugMix, deblur-first, squint
custom BN, real-blur datasets) are not offered.
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
                  blur_exposure=None, stored_psf_count=T.STORED_PSF_COUNT, LEHE_blur_seg=False):
    """reference train.py:48-86: [BlurImage] -> ToTensor -> [RandomHorizontalFlip(0.5) when training]."""
    tf = []
    if blur:
        tf.append(T.BlurImage(prob=blur_ratio, blur_type=blur_type, blur_exposure=blur_exposure, use_stored_psfs=use_stored_psfs,
                              stored_psf_directory=stored_psf_directory, blur_image_in_transform=cpu_blur,
                              dont_center_psf=dont_center_psf, low_exposure=low_exposure, high_exposure=high_exposure,
                              stored_psf_count=stored_psf_count, LEHE_blur_seg=LEHE_blur_seg))
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


def build_parser():
    p = argparse.ArgumentParser(description="detectInBlur hot path on MI355X: training")
    p.add_argument("--dataset", default="coco")
    p.add_argument("--data_path", default=None)
    p.add_argument("--synthetic", action="store_true", help="COCO-shaped synthetic data (no dataset on disk needed)")
    p.add_argument("--synthetic_images", default=64, type=int)
    p.add_argument("--synthetic_size", default=[800, 1333], nargs=2, type=int)
    p.add_argument("--use_stored_psfs", action="store_true")
    p.add_argument("--stored_psf_directory", default=None)
    p.add_argument("--stored_psf_count", default=T.STORED_PSF_COUNT, type=int)
    p.add_argument("-j", "--workers", default=0, type=int)
    p.add_argument("--model", default="fasterrcnn_resnet50_fpn")
    p.add_argument("--trainable_backbone_blocks", default=3, type=int)
    p.add_argument("--pretrained", action="store_true")
    p.add_argument("--device", default="cuda")
    p.add_argument("-b", "--batch_size", default=8, type=int)
    p.add_argument("--lr", default=0.04, type=float)
    p.add_argument("--lr-steps", default=[16, 22], nargs="+", type=int)
    p.add_argument("--lr-gamma", default=0.1, type=float)
    p.add_argument("--epochs", default=37, type=int)
    p.add_argument("--momentum", default=0.9, type=float)
    p.add_argument("--weight_decay", default=1e-4, type=float)
    p.add_argument("--resume", default=None)
    p.add_argument("--start_from_weights", default=None)
    p.add_argument("--start_epoch", default=0, type=int)
    p.add_argument("--early_stop", type=int, default=None)
    p.add_argument("--eval_first", action="store_true")
    p.add_argument("--output_dir", default="debug")
    p.add_argument("--print_freq", default=20, type=int)
    p.add_argument("--blur_train", action="store_true")
    p.add_argument("--cpu_blur", action="store_true")
    p.add_argument("--gpu_blur", action="store_true")
    p.add_argument("--param_index", default=None)
    p.add_argument("--high_exposure", action="store_true")
    p.add_argument("--low_exposure", action="store_true")
    p.add_argument("--expand_target_boxes", action="store_true")
    p.add_argument("--dont_center_psf", action="store_true")
    p.add_argument("--add_noise", action="store_true")
    p.add_argument("--noise_level", default=0.001, type=float)
    p.add_argument("--add_block", action="store_true")
    p.add_argument("--add_jpeg_artefacts", action="store_true", help="Add jpeg compression artifacts.")
    p.add_argument("--use_custom_image_norm", action="store_true")
    p.add_argument("--warp_in_model", action="store_true", help="Warp and dewarp images before and after backbone.")
    p.add_argument("--world-size", default=1, type=int)
    p.add_argument("--dist-url", default="env://")
    return p


def main(args):
    utils.init_distributed_mode(args)
    print(args)
    seed_everything(args.distributed)
    device = torch.device(args.device if torch.cuda.is_available() or args.device == "cpu" else "cpu")

    if args.use_stored_psfs:                                            # reference train.py:127-137
        blur_type = None if args.param_index is None else int(args.param_index)
    else:
        blur_type = None if args.param_index is None else [0.01, 0.005, 0.001, 0.00005][int(args.param_index)]
    blur_ratio = 0.75 if args.low_exposure else (1 if args.high_exposure else 0.9)   # :139-144
    synthetic = dict(num_images=args.synthetic_images, size=tuple(args.synthetic_size), as_tensor=not args.cpu_blur) if args.synthetic else None
    common = dict(blur_type=blur_type, blur_ratio=blur_ratio, use_stored_psfs=args.use_stored_psfs, cpu_blur=args.cpu_blur,
                  stored_psf_directory=args.stored_psf_directory, dont_center_psf=args.dont_center_psf,
                  high_exposure=args.high_exposure, low_exposure=args.low_exposure, stored_psf_count=args.stored_psf_count)
    dataset, num_classes = get_coco(args.data_path, "train", get_transform(True, blur=args.blur_train, **common), synthetic=synthetic)
    dataset_test, _ = get_coco(args.data_path, "val", get_transform(False, blur=False), synthetic=synthetic)
    dataset_test_blur, _ = get_coco(args.data_path, "val", get_transform(False, blur=True, **common), synthetic=synthetic)

    if args.distributed:
        train_sampler = torch.utils.data.distributed.DistributedSampler(dataset)
        test_sampler = torch.utils.data.distributed.DistributedSampler(dataset_test, shuffle=False)
    else:
        train_sampler = torch.utils.data.RandomSampler(dataset)
        test_sampler = torch.utils.data.SequentialSampler(dataset_test)
    batch_sampler = torch.utils.data.BatchSampler(train_sampler, args.batch_size, drop_last=True)
    pin = device.type == "cuda"
    data_loader = torch.utils.data.DataLoader(dataset, batch_sampler=batch_sampler, num_workers=args.workers, collate_fn=utils.collate_fn,
                                              pin_memory=pin, worker_init_fn=_seed_worker)
    mk_test = lambda ds: torch.utils.data.DataLoader(ds, batch_size=1, sampler=test_sampler, num_workers=args.workers,  # noqa: E731
                                                     collate_fn=utils.collate_fn, pin_memory=pin, worker_init_fn=_seed_worker)
    data_loader_test, data_loader_test_blur = mk_test(dataset_test), mk_test(dataset_test_blur)

    print("Creating model")
    model = fasterrcnn_resnet50_fpn(num_classes=num_classes, pretrained=args.pretrained, pretrained_backbone=args.pretrained,
                                    trainable_backbone_layers=args.trainable_backbone_blocks,
                                    warp_internally=args.warp_in_model)
    model.to(device)
    model_without_ddp = model
    if args.distributed:
        # the buffers are frozen batch-norm statistics: equal on all ranks after DDP's construction-time sync,
        # so the per-forward re-broadcast (~7 ms per step) is switched off
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[args.gpu] if device.type == "cuda" else None,
                                                          broadcast_buffers=False, gradient_as_bucket_view=True)
        model_without_ddp = model.module
    params = [p for p in model.parameters() if p.requires_grad]
    optimizer = torch.optim.SGD(params, lr=args.lr, momentum=args.momentum, weight_decay=args.weight_decay)
    lr_scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=args.lr_steps, gamma=args.lr_gamma)

    if args.resume:                                                     # reference train.py:251-257
        ck = torch.load(args.resume, map_location="cpu", weights_only=False)
        model_without_ddp.load_state_dict(ck["model"])
        optimizer.load_state_dict(ck["optimizer"])
        lr_scheduler.load_state_dict(ck["lr_scheduler"])
        args.start_epoch = ck["epoch"] + 1
    elif args.start_from_weights:                                       # :260-263
        model_without_ddp.load_state_dict(torch.load(args.start_from_weights, map_location="cpu", weights_only=False)["model"])

    # the clean pass never uses the custom statistics (reference train.py:346-349 passes none of the blur
    # options): its blur_dicts have no "blurring" key for get_norm_params to read
    clean_kw = dict(device=device, distributed_mode=args.distributed, early_stop=args.early_stop)
    eval_kw = dict(clean_kw, use_custom_image_norm=args.use_custom_image_norm)
    if args.eval_first:
        evaluate(model, data_loader_test, vanilla_eval=True, **clean_kw)

    print("Start training")
    start = time.time()
    for epoch in range(args.start_epoch, args.epochs):
        if args.distributed:
            train_sampler.set_epoch(epoch)
        train_one_epoch(model, optimizer, data_loader, device, epoch, args.print_freq, None, args.distributed, args.blur_train,
                        args.early_stop, args.gpu_blur, args.expand_target_boxes, args.use_custom_image_norm, args.add_noise,
                        args.noise_level, args.add_block, args.add_jpeg_artefacts)
        lr_scheduler.step()
        if args.output_dir:
            utils.mkdir(args.output_dir)
            utils.save_on_master({"model": model_without_ddp.state_dict(), "optimizer": optimizer.state_dict(),
                                  "lr_scheduler": lr_scheduler.state_dict(), "args": args, "epoch": epoch},
                                 os.path.join(args.output_dir, "model_{}.pth".format(epoch)))
        evaluate(model, data_loader_test, vanilla_eval=True, **clean_kw)
        evaluate(model, data_loader_test_blur, blurring_images=True, gpu_blur=args.gpu_blur,
                 expand_target_boxes=args.expand_target_boxes, add_noise=args.add_noise, noise_level=args.noise_level,
                 add_block=args.add_block, add_jpeg_artifact=args.add_jpeg_artefacts, **eval_kw)
    print("Training time {}".format(str(datetime.timedelta(seconds=int(time.time() - start)))))


def _seed_worker(worker_id):
    """Every DataLoader worker gets its own numpy / random stream (the reference passes no
    worker_init_fn, so with the torch of its era all workers replayed one numpy stream: SURVEY A.9)."""
    s = torch.initial_seed() % 2 ** 31
    np.random.seed(s)
    random.seed(s)


if __name__ == "__main__":
    main(build_parser().parse_args())
