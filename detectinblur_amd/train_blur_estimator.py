"""Blur-estimator training driver -- reference train_blur_estimator.py (flags :511-585, loop :300-480).

ResNet-18 with a 16-way (or, with --LEHE_blur_seg, 4-way) head, trained on COCO images blurred on the
GPU by the same HIP path as the detector.  One process per GPU; `torchrun` / RANK, WORLD_SIZE,
LOCAL_RANK as in train.py.  Flags of subsystems that are out of scope (AugMix, TensorBoard) are not offered.
"""
import argparse
import datetime
import os
import time

import torch
from torch import nn

from . import utils
from .coco_utils import get_coco
from .engine_blur_estimator import evaluate, train_one_epoch
from .models.blur_estimator import resnet18
from .train import _seed_worker, get_transform, seed_everything


def build_parser():
    p = argparse.ArgumentParser(description="detectInBlur hot path on MI355X: blur-estimator training")
    p.add_argument("--dataset", default="coco")
    p.add_argument("--data_path", default=None)
    p.add_argument("--synthetic", action="store_true", help="COCO-shaped synthetic data (no dataset on disk needed)")
    p.add_argument("--synthetic_images", default=64, type=int)
    p.add_argument("--synthetic_size", default=[480, 640], nargs=2, type=int)
    p.add_argument("--use_stored_psfs", action="store_true")
    p.add_argument("--stored_psf_directory", default=None)
    p.add_argument("--stored_psf_count", default=12000, type=int)
    p.add_argument("--crop_images", action="store_true", help="Crop images when batching.")
    p.add_argument("--resize_images", action="store_true")
    p.add_argument("--quantize_image", action="store_true")
    p.add_argument("--device", default="cuda")
    p.add_argument("-b", "--batch_size", default=8, type=int)
    p.add_argument("-j", "--workers", default=0, type=int)
    p.add_argument("--lr", default=0.04, type=float)
    p.add_argument("--lr-steps", default=[16, 22], nargs="+", type=int)
    p.add_argument("--lr-gamma", default=0.1, type=float)
    p.add_argument("--epochs", default=37, type=int)
    p.add_argument("--momentum", default=0.9, type=float)
    p.add_argument("--wd", "--weight-decay", dest="weight_decay", default=1e-4, type=float)
    p.add_argument("--resume", default=None)
    p.add_argument("--start_from_weights", default=None)
    p.add_argument("--start_epoch", default=0, type=int)
    p.add_argument("--early_stop", type=int, default=None)
    p.add_argument("--eval_first", action="store_true")
    p.add_argument("--test_only", action="store_true")
    p.add_argument("--output_dir", default="debug")
    p.add_argument("--print_freq", default=20, type=int)
    p.add_argument("--blur_train", action="store_true")
    p.add_argument("--gpu_blur", action="store_true")
    p.add_argument("--param_index", default=None)
    p.add_argument("--LEHE_blur_seg", action="store_true")
    p.add_argument("--high_exposure", action="store_true")
    p.add_argument("--low_exposure", action="store_true")
    p.add_argument("--dont_center_psf", action="store_true")
    p.add_argument("--add_noise", action="store_true")
    p.add_argument("--noise_level", default=0.001, type=float)
    p.add_argument("--add_block", action="store_true")
    p.add_argument("--add_jpeg_artefacts", action="store_true")
    p.add_argument("--world-size", default=1, type=int)
    p.add_argument("--dist-url", default="env://")
    return p


def main(args):
    from . import kernel_choices
    kernel_choices.use_shipped_kernel_choices()      # shipped MIOpen / TunableOp choices, private copy per process (kernel_choices.py)
    mp_ctx = utils.loader_context() if args.workers > 0 else None      # before anything touches the GPU (see utils.loader_context)
    utils.init_distributed_mode(args)
    print(args)
    seed_everything(args.distributed)
    device = torch.device(args.device if torch.cuda.is_available() or args.device == "cpu" else "cpu")
    if args.use_stored_psfs:
        blur_type = None if args.param_index is None else int(args.param_index)
    else:
        blur_type = None if args.param_index is None else [0.01, 0.005, 0.001, 0.00005][int(args.param_index)]
    blur_ratio = 0.75 if args.low_exposure else (1 if args.high_exposure else 0.9)
    synthetic = dict(num_images=args.synthetic_images, size=tuple(args.synthetic_size)) if args.synthetic else None
    common = dict(blur=True, blur_type=blur_type, blur_ratio=blur_ratio, use_stored_psfs=args.use_stored_psfs,
                  stored_psf_directory=args.stored_psf_directory, dont_center_psf=args.dont_center_psf,
                  low_exposure=args.low_exposure, high_exposure=args.high_exposure, stored_psf_count=args.stored_psf_count,
                  LEHE_blur_seg=args.LEHE_blur_seg)
    dataset, _ = get_coco(args.data_path, "train", get_transform(True, **common), synthetic=synthetic, with_masks=False)
    dataset_test, _ = get_coco(args.data_path, "val", get_transform(False, **common), synthetic=synthetic, with_masks=False)
    if args.distributed:
        train_sampler = torch.utils.data.distributed.DistributedSampler(dataset)
        test_sampler = torch.utils.data.distributed.DistributedSampler(dataset_test, shuffle=False)
    else:
        train_sampler = torch.utils.data.RandomSampler(dataset)
        test_sampler = torch.utils.data.SequentialSampler(dataset_test)
    pin = device.type == "cuda"
    loader = torch.utils.data.DataLoader(dataset, batch_size=args.batch_size, sampler=train_sampler, num_workers=args.workers,
                                         collate_fn=utils.collate_fn, drop_last=True, pin_memory=pin, worker_init_fn=_seed_worker,
                                         multiprocessing_context=mp_ctx)
    loader_test = torch.utils.data.DataLoader(dataset_test, batch_size=1, sampler=test_sampler,      # reference train_blur_estimator.py:206
                                              num_workers=args.workers, collate_fn=utils.collate_fn, pin_memory=pin,
                                              worker_init_fn=_seed_worker, multiprocessing_context=mp_ctx)

    print("Creating model")
    model = resnet18()
    model.fc = nn.Linear(512, 4 if args.LEHE_blur_seg else 16)          # reference evaluate.py:188-194
    model = model.to(device).to(memory_format=torch.channels_last)
    bare = model
    if args.distributed:
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[args.gpu] if device.type == "cuda" else None)
        bare = model.module
    criterion = nn.CrossEntropyLoss().to(device)
    optimizer = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=args.lr, momentum=args.momentum,
                                weight_decay=args.weight_decay)
    scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=args.lr_steps, gamma=args.lr_gamma)
    if args.resume:
        ck = torch.load(args.resume, map_location="cpu", weights_only=False)
        bare.load_state_dict(ck["model"]); optimizer.load_state_dict(ck["optimizer"]); scheduler.load_state_dict(ck["lr_scheduler"])
        args.start_epoch = ck["epoch"] + 1
    elif args.start_from_weights:
        bare.load_state_dict(torch.load(args.start_from_weights, map_location="cpu", weights_only=False)["model"])

    eval_kw = dict(device=device, distributed_mode=args.distributed, blurring_images=True, gpu_blur=args.gpu_blur,
                   LEHE_blur_seg=args.LEHE_blur_seg, resize_images=args.resize_images, quantize_image=args.quantize_image,
                   add_noise=args.add_noise, noise_level=args.noise_level, add_block=args.add_block,
                   add_jpeg_artifact=args.add_jpeg_artefacts, early_stop=args.early_stop)
    if args.eval_first or args.test_only:
        evaluate(model, loader_test, **eval_kw)
        if args.test_only:
            return
    print("Start training")
    start = time.time()
    for epoch in range(args.start_epoch, args.epochs):
        if args.distributed:
            train_sampler.set_epoch(epoch)
        train_one_epoch(model, optimizer, criterion, loader, device, args.print_freq, epoch, args.distributed, None,
                        args.gpu_blur, args.LEHE_blur_seg, args.resize_images, args.quantize_image, args.crop_images,
                        args.add_noise, args.noise_level, args.add_block, args.add_jpeg_artefacts, args.early_stop, args.blur_train)
        scheduler.step()
        if args.output_dir:
            utils.mkdir(args.output_dir)
            utils.save_on_master({"model": bare.state_dict(), "optimizer": optimizer.state_dict(),
                                  "lr_scheduler": scheduler.state_dict(), "args": args, "epoch": epoch},
                                 os.path.join(args.output_dir, "blur_estimator_{}.pth".format(epoch)))
        evaluate(model, loader_test, **eval_kw)
    print("Training time {}".format(str(datetime.timedelta(seconds=int(time.time() - start)))))


if __name__ == "__main__":
    main(build_parser().parse_args())
