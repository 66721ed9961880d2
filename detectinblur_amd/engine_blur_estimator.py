"""Blur-estimator training / evaluation loops -- SURVEY.md section 8f-4, reference
engine_blur_estimator.py:82-130 (labels), :132-298 (train_one_epoch), :300-492 (evaluate).

The estimator is the ResNet-18 classifier that routes an image to one of the specialised detectors
(`evaluate.py --use_ensemble`): 16 classes (sharp + 3 blur types x 5 exposures) or 4 classes with
`LEHE_blur_seg` (low exposure of any type, or high exposure of type 1 / 2 / 3).  It trains on the same
on-GPU motion blur as the detector, so the hot path is the same two HIP launches per batch
(`models/blur_functions.blur_image_list`); the reference's private copy of the roll loop (:27-79) adds an
optional bilinear resize to 800 px around the blur (`resize_images`), kept here as two stock
`interpolate` calls around the same kernel.
"""
import math
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import utils
from .models import blur_functions
from .models.net_transforms import GeneralizedRCNNTransform

IMAGE_MEAN, IMAGE_STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


# ---- labels --------------------------------------------------------------------------------------------

def get_target_from_blur_dict(blur_dicts, target):
    """16-way label: 0 = not blurred, else param_index * 5 + fraction_index + 1 (reference :100-107)."""
    for i, bd in enumerate(blur_dicts):
        target[i] = bd["param_index"] * 5 + bd["fraction_index"] + 1 if bd["blurring"] else 0
    return target


def get_target_from_blur_dict_LEHE(blur_dicts, target):
    """4-way label (reference :109-130): an explicit `blur_est_label` wins; not blurred or exposure index
    below 3 -> 0 (low exposure); otherwise 1 + param_index."""
    for i, bd in enumerate(blur_dicts):
        if "blur_est_label" in bd:
            target[i] = bd["blur_est_label"]
        elif bd["blurring"] and bd["fraction_index"] >= 3:
            target[i] = 1 + bd["param_index"]
        else:
            target[i] = 0
    return target


def accuracy(output, target, topk=(1,)):
    """Top-k accuracies in per cent (reference :82-97)."""
    with torch.no_grad():
        maxk = max(topk)
        pred = output.topk(maxk, 1, True, True)[1].t()
        correct = pred.eq(target.view(1, -1).expand_as(pred))
        return [correct[:k].reshape(-1).float().sum(0, keepdim=True).mul_(100.0 / target.size(0)) for k in topk]


# ---- blur with the optional 800-px round trip ----------------------------------------------------------

def blur_image_list(images_GPU, blur_dicts, psfs_GPU, resize_images=False):
    """reference :69-79.  With resize_images every blurred image is brought to height 800 (aspect kept,
    portrait images transposed first), blurred, and brought back -- bilinear both ways (:33-44, :66-72)."""
    if not resize_images:
        return blur_functions.blur_image_list(images_GPU, blur_dicts, psfs_GPU)
    shapes, work = {}, list(images_GPU)
    for i, (img, bd) in enumerate(zip(images_GPU, blur_dicts)):
        if not bd["blurring"]:
            continue
        h, w = img.shape[-2], img.shape[-1]
        x = img.unsqueeze(0)
        if h > w:
            x = x.permute(0, 1, 3, 2)
            new_w = int(800 * h / w)
        else:
            new_w = int(800 * w / h)
        shapes[i] = (h, w)
        work[i] = F.interpolate(x, size=(800, new_w), mode="bilinear").squeeze(0)
    blur_functions.blur_image_list(work, blur_dicts, psfs_GPU)
    for i, (h, w) in shapes.items():
        # the reference interpolates the blurred (still transposed, for portrait images) tensor straight to
        # (image_height, image_width) without transposing back (:66-72); reproduced as is
        images_GPU[i] = F.interpolate(work[i].unsqueeze(0), size=(h, w), mode="bilinear").squeeze(0).squeeze()
    return None


def _post(images_GPU, add_noise, noise_level, add_block, quantize_image, jpeg_compressor=None):
    for i, img in enumerate(images_GPU):                                   # reference :200-219
        if add_noise:
            noise_var = np.random.uniform(0.0001, noise_level)
            img = torch.clamp(img + torch.randn_like(img) * math.sqrt(noise_var), 0, 1)
        if add_block and np.random.uniform(0, 1) > 0.3:
            shape = img.shape
            s = np.random.uniform(0.6, 1)
            img = F.interpolate(img.unsqueeze(0), scale_factor=(s, s), mode="nearest").squeeze()
            img = F.interpolate(img.unsqueeze(0), size=shape[1:], mode="nearest").squeeze()
        if jpeg_compressor is not None and np.random.uniform(0, 1) > 0.35:
            from .transforms import add_jpeg_artifact_to_image
            img = add_jpeg_artifact_to_image(img, jpeg_compressor, np.random.uniform(20, 90)).to(img.device)
        if quantize_image:
            img = (img * 255).type(torch.uint8).type(torch.half) / 255
        images_GPU[i] = img
    return images_GPU


def _jpeg(device):
    from .models.jpeg import DiffJPEG
    return DiffJPEG(height=100, width=100, differentiable=False, quality=10).to(device)


def _stage(images_CPU, blur_dicts, device, with_psfs):
    images = [im.half().to(device, non_blocking=True) for im in images_CPU]
    psfs = None
    if with_psfs:
        psfs = [torch.HalfTensor(bd["psf"]).to(device, non_blocking=True) for bd in blur_dicts]
    return images, psfs


def _targets(blur_dicts, device, LEHE_blur_seg):
    t = torch.zeros(len(blur_dicts), dtype=torch.long)
    t = (get_target_from_blur_dict_LEHE if LEHE_blur_seg else get_target_from_blur_dict)(blur_dicts, t)
    return t.to(device)


# ---- loops ---------------------------------------------------------------------------------------------

def train_one_epoch(model, optimizer, criterion, data_loader, device, print_freq=500, epoch=0, distributed_mode=False,
                    writer=None, gpu_blur=False, LEHE_blur_seg=False, resize_images=False, quantize_image=False,
                    crop_images=False, add_noise=False, noise_level=0.001, add_block=False, add_jpeg_artifact=False,
                    early_stop=None, blur_train=False):
    jpeg = _jpeg(device) if add_jpeg_artifact else None
    batcher = GeneralizedRCNNTransform(800, 1333, IMAGE_MEAN, IMAGE_STD, crop_images=crop_images)
    model.train()
    logger = utils.MetricLogger(delimiter="  ")
    logger.add_meter("lr", utils.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    lr_scheduler = None
    if epoch == 0:
        lr_scheduler = utils.warmup_lr_scheduler(optimizer, min(1000, len(data_loader) - 1), 1.0 / 1000)
    it = 0
    for images_CPU, targets, blur_dicts in logger.log_every(data_loader, print_freq, "Epoch: [{}]".format(epoch)):
        images, psfs = _stage(images_CPU, blur_dicts, device, blur_train)
        if gpu_blur and blur_train:
            blur_image_list(images, blur_dicts, psfs, resize_images)
        images = _post(images, add_noise, noise_level, add_block, quantize_image, jpeg)
        batch = batcher([im.float() for im in images])[0].tensors
        target = _targets(blur_dicts, device, LEHE_blur_seg)
        loss = criterion(model(batch), target)
        reduced = utils.reduce_dict({"loss": loss})["loss"]
        value = reduced.item()
        if not math.isfinite(value):
            print("Loss is {}, stopping training".format(value))
            sys.exit(1)
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        if lr_scheduler is not None:
            lr_scheduler.step()
        logger.update(loss=reduced, lr=optimizer.param_groups[0]["lr"])
        it += 1
        if early_stop is not None and it > early_stop:
            break
    return logger


@torch.no_grad()
def evaluate(model, data_loader, device, distributed_mode=False, blurring_images=False, gpu_blur=False, LEHE_blur_seg=False,
             send_back_preds_targets=False, add_jpeg_artifact=False, resize_images=False, quantize_image=False,
             add_noise=False, noise_level=0.001, add_block=False, early_stop=None):
    """Top-1 / top-2 accuracy of the estimator; returns (top1, top2[, predictions, targets])."""
    jpeg = _jpeg(device) if add_jpeg_artifact else None
    batcher = GeneralizedRCNNTransform(800, 1333, IMAGE_MEAN, IMAGE_STD)
    model.eval()
    logger = utils.MetricLogger(delimiter="  ")
    seen, hit1, hit2, preds, tgts = 0, 0.0, 0.0, [], []
    for n, (images_CPU, targets, blur_dicts) in enumerate(logger.log_every(data_loader, 100, "Test:")):
        t0 = time.time()
        images, psfs = _stage(images_CPU, blur_dicts, device, blurring_images)
        if gpu_blur and blurring_images:
            blur_image_list(images, blur_dicts, psfs, resize_images)
        images = _post(images, add_noise, noise_level, add_block, quantize_image, jpeg)
        out = model(batcher([im.float() for im in images])[0].tensors)
        target = _targets(blur_dicts, device, LEHE_blur_seg)
        a1, a2 = accuracy(out, target, topk=(1, 2))
        b = target.numel()
        seen += b
        hit1 += float(a1) * b / 100.0
        hit2 += float(a2) * b / 100.0
        if send_back_preds_targets:
            preds.append(out.argmax(1).cpu())
            tgts.append(target.cpu())
        logger.update(model_time=time.time() - t0)
        if early_stop is not None and n >= early_stop:
            break
    stats = torch.tensor([seen, hit1, hit2], dtype=torch.float64, device=device)
    if utils.is_dist_avail_and_initialized():
        torch.distributed.all_reduce(stats)
    seen, hit1, hit2 = stats.tolist()
    top1, top2 = 100.0 * hit1 / max(seen, 1), 100.0 * hit2 / max(seen, 1)
    print("Blur estimator accuracy: top-1 {:.2f} %  top-2 {:.2f} %  ({:d} images)".format(top1, top2, int(seen)))
    if send_back_preds_targets:
        return top1, top2, torch.cat(preds) if preds else torch.empty(0), torch.cat(tgts) if tgts else torch.empty(0)
    return top1, top2
