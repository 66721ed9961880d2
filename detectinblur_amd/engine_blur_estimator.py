"""Blur-estimator training / evaluation loops -- SURVEY.md section 8f-4, reference
engine_blur_estimator.py:82-130 (labels), :132-298 (train_one_epoch), :300-492 (evaluate).

The estimator is the ResNet-18 classifier that routes an image to one of the specialised detectors
(`evaluate.py --use_ensemble`): 16 classes (sharp + 3 blur types x 5 exposures) or 4 classes with
`LEHE_blur_seg` (low exposure of any type, or high exposure of type 1 / 2 / 3).  It trains on the same
on-GPU motion blur as the detector, so the hot path is the same HIP launch per batch
(`models/blur_functions.blur_image_list`); the reference's private copy of the roll loop (:27-79) adds an
optional bilinear resize to 800 px around the blur (`resize_images`), kept here as two stock
`interpolate` calls around the same kernel.  Both loops are pinned against the reference's own functions run in
the build container (oracle/gen_detector_pins.py: `estimator`; tests/test_blur_estimator.py).
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
    """reference :69-79 around its private `manual_blur` (:27-67).  With resize_images every blurred image is first brought to
    height 800 (bilinear, aspect kept; a portrait image is transposed first and stays transposed), blurred, CROPPED to its
    ORIGINAL height x width from the top-left corner -- the reference takes `image_height` / `image_width` before the resize and
    crops the padded result with them (:29-30, :64) -- and that crop is interpolated to the original size (:66-72; a no-op
    resample unless the crop ran into the resized image's edge).  Pinned by tests/golden/detector_pins.json
    (`estimator/*/blur_resize_quant`): the blur is the same HIP launch, on the resized image."""
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
        crop = work[i][..., :h, :w]                     # `output[:, :, 63:63 + image_height, 63:63 + image_width]` of the padded result
        images_GPU[i] = F.interpolate(crop.unsqueeze(0), size=(h, w), mode="bilinear").squeeze()
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
    # the reference starts from a uniformly random label vector (drawn on the host from torch's default generator, :228-236,
    # :404-414) and overwrites every entry; the draw is kept so that a seeded run consumes the generator as the reference does
    t = torch.zeros(len(blur_dicts), requires_grad=False).uniform_(0, 3 if LEHE_blur_seg else 15).long()
    t = (get_target_from_blur_dict_LEHE if LEHE_blur_seg else get_target_from_blur_dict)(blur_dicts, t)
    return t.to(device)


# ---- loops ---------------------------------------------------------------------------------------------

def train_one_epoch(model, optimizer, criterion, data_loader, device, print_freq=500, epoch=0, distributed_mode=False,
                    writer=None, gpu_blur=False, LEHE_blur_seg=False, resize_images=False, quantize_image=False,
                    crop_images=False, add_noise=False, noise_level=0.001, add_block=False, add_jpeg_artifact=False,
                    early_stop=None, blur_train=False):
    """reference :132-298, argument for argument.  Pinned against the reference's own function on a toy classifier
    (oracle/gen_detector_pins.py -> tests/test_blur_estimator.py): weights, losses, label vectors, LR trajectory, scalars."""
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
        targets_dev = [{k: v.to(device) for k, v in t.items()} for t in targets]
        if gpu_blur and blur_train:
            blur_image_list(images, blur_dicts, psfs, resize_images)
        images = _post(images, add_noise, noise_level, add_block, quantize_image, jpeg)
        batch = batcher([im.float() for im in images], targets_dev)[0].tensors
        target = _targets(blur_dicts, device, LEHE_blur_seg)
        loss_dict = {"loss": criterion(model(batch), target)}
        losses = sum(loss for loss in loss_dict.values())
        loss_dict_reduced = utils.reduce_dict(loss_dict)
        losses_reduced = sum(loss for loss in loss_dict_reduced.values())
        value = losses_reduced.item()
        if writer is not None and (not distributed_mode or torch.distributed.get_rank() == 0) and it % print_freq == 0:
            step = it + epoch * len(data_loader)
            for key, item in loss_dict_reduced.items():
                writer.add_scalar("losses/" + key, item, step)
            writer.add_scalar("losses/overallLoss", value, step)
            writer.add_scalar("learningRate", optimizer.param_groups[0]["lr"], step)
        if not math.isfinite(value):
            print("Loss is {}, stopping training".format(value))
            print(loss_dict_reduced)
            sys.exit(1)
        optimizer.zero_grad()
        losses.backward()
        optimizer.step()
        if lr_scheduler is not None:
            lr_scheduler.step()
        logger.update(loss=losses_reduced)
        logger.update(lr=optimizer.param_groups[0]["lr"])
        it += 1
        if early_stop is not None and it > early_stop:
            break
    return logger


@torch.no_grad()
def evaluate(model, data_loader, device, distributed_mode=False, blurring_images=False, gpu_blur=False, LEHE_blur_seg=False,
             send_back_preds_targets=False, add_jpeg_artifact=False, resize_images=False, quantize_image=False,
             add_noise=False, noise_level=0.001, add_block=False, early_stop=None):
    """reference :301-492, argument for argument and return for return: `accuracies` = [top-1, top-2] in per cent over the images
    seen, or (accuracies, targetsAll, predsAll) with `send_back_preds_targets` -- the lists hold `target[0]` and `pred[0]` of
    every batch, as the reference's do (it evaluates with batch size 1, train_blur_estimator.py:206; with larger batches its
    per-class summary fails on the stacked shapes, and so does this one).  Prints the reference's three summary lines."""
    n_threads = torch.get_num_threads()
    jpeg = _jpeg(device) if add_jpeg_artifact else None
    batcher = GeneralizedRCNNTransform(800, 1333, IMAGE_MEAN, IMAGE_STD)
    torch.set_num_threads(1)
    model.eval()
    logger = utils.MetricLogger(delimiter="  ")
    count, total = 0, 0
    correctCounts = [0, 0]
    targetsAll, predsAll = [], []
    topk = (1, 2)
    for images_CPU, targets, blur_dicts in logger.log_every(data_loader, 100, "Test:"):
        model_time = time.time()
        images, psfs = _stage(images_CPU, blur_dicts, device, blurring_images)
        if gpu_blur:
            if psfs is None:      # the reference reads `psfs_GPU`, which only `blurring_images` assigns (:352-358, :361)
                raise UnboundLocalError("local variable 'psfs_GPU' referenced before assignment")
            blur_image_list(images, blur_dicts, psfs, resize_images)
        images = _post(images, add_noise, noise_level, add_block, quantize_image, jpeg)
        outputs = model(batcher([im.float() for im in images])[0].tensors)
        model_time = time.time() - model_time
        evaluator_time = time.time()
        target = _targets(blur_dicts, device, LEHE_blur_seg)
        total += target.size(0)
        pred = outputs.topk(max(topk), 1, True, True)[1].t()
        correct = pred.eq(target.view(1, -1).expand_as(pred))
        targetsAll.append(target[0])
        predsAll.append(pred[0])
        for k_index, k in enumerate(topk):
            correctCounts[k_index] += correct[:k].reshape(-1).float().sum(0, keepdim=True)
        logger.update(model_time=model_time, evaluator_time=time.time() - evaluator_time)
        count += 1
        if early_stop is not None and count > early_stop:
            break
    logger.synchronize_between_processes()
    accuracies = [100 * (correctCounts[0].item() / total), 100 * (correctCounts[1].item() / total)]
    print("Top 1 Accuracy: {0:.2f}%".format(accuracies[0]))
    print("Top 2 Accuracy: {0:.2f}%".format(accuracies[1]))
    mergedPreds = torch.stack(predsAll).squeeze()
    mergedTargets = torch.stack(targetsAll).squeeze()
    totalAcc, valid_class_count = 0, 0
    for classInd in range(4):                       # the reference's summary looks at labels 0..3 whatever the label set (:470)
        class_count = int((mergedTargets == classInd).sum())
        if class_count == 0:
            continue
        valid_class_count += 1
        totalAcc += int(torch.logical_and(mergedTargets == classInd, mergedPreds == mergedTargets).sum()) / class_count
    totalAcc = totalAcc / valid_class_count
    print("Top 1 Mean Acc: {0:.2f}%".format(totalAcc * 100))
    torch.set_num_threads(n_threads)
    if send_back_preds_targets:
        return accuracies, targetsAll, predsAll
    return accuracies
