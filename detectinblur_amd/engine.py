"""Epoch engines -- drop-in for the hot-path parts of the reference's engine.py
(`train_one_epoch` :30-167, `evaluate` :220-416, ensemble routers :171-218).

What one training step does, in order (reference engine.py:74-158):
  H2D as fp16 -> blur_image_list (HIP) -> expand_targets (HIP) -> .float() -> per-image norm statistics
  -> model(images, targets, newMeans, newSTDs) -> loss all-reduce for logging -> backward (DDP all-reduce
  over RCCL overlapped with it) -> SGD step -> warm-up LR step.
Differences from the reference are confined to mechanics: host->device copies are pinned + non-blocking,
the blur needs no host synchronisation, and the logged loss is fetched with one `.item()` per step.
COCO mAP (pycocotools, `coco_eval.py`) is outside the built path; `evaluate`
returns the raw detections plus timing instead of a CocoEvaluator.
"""
import math
import sys
import time

import torch

from . import utils
from .models import blur_functions, net_transforms


def _to_device(images_CPU, targets, blur_dicts, device, blurring):
    """reference engine.py:79-98: images as Half, PSFs via torch.HalfTensor(ndarray)."""
    images_GPU = [image.half().to(device, non_blocking=True) for image in images_CPU]
    targets_GPU = [{k: (v.to(device, non_blocking=True) if isinstance(v, torch.Tensor) else v) for k, v in t.items()} for t in targets]
    psfs_GPU = thetas = l1 = l2 = None
    if blurring:
        psfs_GPU = [torch.HalfTensor(bd["psf"]).to(device, non_blocking=True) for bd in blur_dicts]
        thetas = torch.tensor([bd["theta_rad"] for bd in blur_dicts], dtype=torch.float16, device=device)
        l1 = torch.tensor([bd["scale_factor_lambda1"] for bd in blur_dicts], dtype=torch.float16, device=device)
        l2 = torch.tensor([bd["scale_factor_lambda2"] for bd in blur_dicts], dtype=torch.float16, device=device)
    return images_GPU, targets_GPU, psfs_GPU, thetas, l1, l2


def train_one_epoch(model, optimizer, data_loader, device, epoch=0, print_freq=200, writer=None, distributed_mode=False,
                    blur_train=False, early_stop=False, gpu_blur=False, expand_target_boxes=False,
                    use_custom_image_norm=False, add_noise=False, noise_level=0.001, add_block=False,
                    add_jpeg_artifact=False):
    if writer is None:
        print("Warning! No tensorboard logger.")
    jpeg_compressor = None
    if add_jpeg_artifact:                                                # reference :56-62
        from .models.jpeg import DiffJPEG
        jpeg_compressor = DiffJPEG(height=100, width=100, differentiable=False, quality=10).to(device)
    model.train()
    metric_logger = utils.MetricLogger(delimiter="  ")
    metric_logger.add_meter("lr", utils.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    header = "Epoch: [{}]".format(epoch)

    lr_scheduler = None
    if epoch == 0:                                                       # reference :64-69
        lr_scheduler = utils.warmup_lr_scheduler(optimizer, min(1000, len(data_loader) - 1), 1.0 / 1000)

    iteration_count = 0
    for images_CPU, targets, blur_dicts in metric_logger.log_every(data_loader, print_freq, header):
        images_GPU, targets_GPU, psfs_GPU, thetas, l1, l2 = _to_device(images_CPU, targets, blur_dicts, device, blur_train)
        if gpu_blur and blur_train:
            blur_functions.blur_image_list(images_GPU, blur_dicts, psfs_GPU=psfs_GPU, add_noise=add_noise,
                                           noise_level=noise_level, add_block=add_block,
                                           add_jpeg_artifact=add_jpeg_artifact, jpeg_compressor=jpeg_compressor)
        if expand_target_boxes and blur_train:
            targets_GPU = utils.expand_targets(targets_GPU, blur_dicts, psfs_GPU, images_GPU)
        images_GPU = [image.float().to(device) for image in images_GPU]   # JPEG artefacts come back on the host (transforms.py:492)
        norm_means, norm_stds = utils.get_norm_params(blur_dicts, use_custom_image_norm)

        loss_dict = model(images_GPU, targets_GPU, thetas=thetas, lambda1s=l1, lambda2s=l2, newMeans=norm_means, newSTDs=norm_stds)
        losses = sum(loss for loss in loss_dict.values())

        loss_dict_reduced = utils.reduce_dict(loss_dict)                 # logging only
        losses_reduced = sum(loss for loss in loss_dict_reduced.values())
        loss_value = losses_reduced.item()
        if iteration_count % 500 == 0 and writer is not None and utils.is_main_process() and iteration_count % print_freq == 0:
            step = iteration_count + epoch * len(data_loader)
            for key, v in loss_dict_reduced.items():
                writer.add_scalar("losses/" + key, v, step)
            writer.add_scalar("losses/overallLoss", loss_value, step)
            writer.add_scalar("learningRate", optimizer.param_groups[0]["lr"], step)
        if not math.isfinite(loss_value):                                # reference :145-148
            print("Loss is {}, stopping training".format(loss_value))
            print(loss_dict_reduced)
            sys.exit(1)

        optimizer.zero_grad()
        losses.backward()
        optimizer.step()
        if lr_scheduler is not None:
            lr_scheduler.step()
        metric_logger.update(loss=losses_reduced, **loss_dict_reduced)
        metric_logger.update(lr=optimizer.param_groups[0]["lr"])
        if early_stop is not None and early_stop is not False and iteration_count > early_stop:
            break
        iteration_count += 1
    return metric_logger


# ---- ensemble routing (reference engine.py:171-218) ------------------------------------------------

def get_network_index_to_use_oracle(blur_dicts, model_indices):
    """Ground-truth routing from the blur_dict of the FIRST image: not blurred / very short exposure ->
    net 0, blur type P1/P2/P3 -> nets 1/2/3."""
    for bd in blur_dicts:
        if bd["blurring"] and bd["param_index"] is not None:
            if bd["fraction_index"] == -1:
                return model_indices[0]
            if bd["param_index"] in (0, 1, 2):
                return model_indices[bd["param_index"] + 1]
            return None
        return model_indices[0]


def get_network_index_to_use_blur_estimator(blur_estimation, model_indices):
    """16-way estimator: class 0 = sharp, 1-5 = P1 x E0..E4, 6-10 = P2, 11-15 = P3."""
    k = int(blur_estimation.argmax())
    return model_indices[0 if k == 0 or k > 15 else 1 + (k - 1) // 5]


def get_network_index_to_use_blur_estimator_LEHE(blur_estimation, model_indices):
    """4-way estimator (--LEHE): class 0 -> low-exposure net, 1/2/3 -> P1/P2/P3 high-exposure nets."""
    k = int(blur_estimation.argmax())
    return model_indices[k if k in (1, 2, 3) else 0]


@torch.no_grad()
def evaluate(model, data_loader, device, distributed_mode=False, early_stop=None, vanilla_eval=False, blurring_images=False,
             gpu_blur=False, expand_target_boxes=False, deblur_first=False, deblurer=None, use_custom_image_norm=False,
             use_ensemble=False, ensemble_models=None, blur_estimator=None, add_noise=False, noise_level=0.001, add_block=False,
             add_jpeg_artifact=False, image_output_folder=None, LEHE=False):
    """Runs the detector (or the routed ensemble) over the loader.  Returns
    {"detections": {image_id: {boxes, labels, scores}}, "targets": {image_id: expanded boxes},
     "routes": [model index per batch], "meters": MetricLogger}."""
    if deblur_first:
        raise NotImplementedError("--deblur_first is outside the built path (SURVEY.md section 2)")
    jpeg_compressor = None
    if add_jpeg_artifact:                                                # reference :248-254
        from .models.jpeg import DiffJPEG
        jpeg_compressor = DiffJPEG(height=100, width=100, differentiable=False, quality=10).to(device)
    n_threads = torch.get_num_threads()
    torch.set_num_threads(1)
    batcher = None
    if use_ensemble:
        for m in ensemble_models:
            m.eval()
        if blur_estimator is not None:                                   # reference :259-265
            batcher = net_transforms.GeneralizedRCNNTransform(800, 1333, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], crop_images=True)
            blur_estimator.eval()
    else:
        model.eval()
    metric_logger = utils.MetricLogger(delimiter="  ")
    detections, gt_boxes, gt_full, routes = {}, {}, {}, []
    count = 0
    for images_CPU, targets_CPU, blur_dicts in metric_logger.log_every(data_loader, 100, "Test:"):
        if device.type == "cuda":
            torch.cuda.synchronize()
        model_time = time.time()
        images_GPU, targets_GPU, psfs_GPU, thetas, l1, l2 = _to_device(images_CPU, targets_CPU, blur_dicts, device, blurring_images)
        if gpu_blur and blurring_images:
            blur_functions.blur_image_list(images_GPU, blur_dicts, psfs_GPU=psfs_GPU, add_noise=add_noise, noise_level=noise_level,
                                           add_block=add_block, add_jpeg_artifact=add_jpeg_artifact,
                                           jpeg_compressor=jpeg_compressor)
        if expand_target_boxes and blurring_images:
            targets_GPU = utils.expand_targets(targets_GPU, blur_dicts, psfs_GPU, images_GPU)
        images_GPU = [image.float().to(device) for image in images_GPU]   # JPEG artefacts come back on the host (transforms.py:492)
        norm_means, norm_stds = utils.get_norm_params(blur_dicts, use_custom_image_norm)

        if use_ensemble:                                                 # reference :354-366
            idx = list(range(len(ensemble_models)))
            if blur_estimator is None:
                k = get_network_index_to_use_oracle(blur_dicts, idx)
            else:
                batched, _ = batcher(images_GPU, None)
                est = blur_estimator(batched.tensors)
                k = (get_network_index_to_use_blur_estimator_LEHE if LEHE else get_network_index_to_use_blur_estimator)(est, idx)
            model = ensemble_models[k]
            routes.append(k)
        if blurring_images:
            outputs = model(images_GPU, thetas=thetas, lambda1s=l1, lambda2s=l2, newMeans=norm_means, newSTDs=norm_stds)
        else:
            outputs = model(images_GPU, killWarp=True, newMeans=norm_means, newSTDs=norm_stds)
        outputs = [{k: v.to("cpu") for k, v in t.items()} for t in outputs]
        model_time = time.time() - model_time
        for t, o in zip(targets_GPU, outputs):
            image_id = int(t["image_id"]) if "image_id" in t else count
            detections[image_id] = o
            gt_boxes[image_id] = utils.convert_to_xywh(t["boxes"]).cpu()   # what the reference writes into coco_gt (:325-342)
            if "labels" in t:
                # annotations as COCOeval sees them: the (expanded) box replaces bbox, area / iscrowd stay (:325-342)
                gt_full[image_id] = {k: t[k].detach().cpu() for k in ("boxes", "labels", "area", "iscrowd") if k in t}
        metric_logger.update(model_time=model_time)
        count += 1
        if early_stop is not None and count > early_stop:
            break
    metric_logger.synchronize_between_processes()
    print("Averaged stats:", metric_logger)
    torch.set_num_threads(n_threads)
    coco_stats = None
    if gt_full:                                                          # reference :410-414 (CocoEvaluator)
        coco_stats = coco_box_stats(detections, gt_full, device)
    return {"detections": detections, "targets": gt_boxes, "routes": routes, "meters": metric_logger, "coco_stats": coco_stats}


_STAT_NAMES = ["AP @[ IoU=0.50:0.95 | area=   all | maxDets=100 ]", "AP @[ IoU=0.50      | area=   all | maxDets=100 ]",
               "AP @[ IoU=0.75      | area=   all | maxDets=100 ]", "AP @[ IoU=0.50:0.95 | area= small | maxDets=100 ]",
               "AP @[ IoU=0.50:0.95 | area=medium | maxDets=100 ]", "AP @[ IoU=0.50:0.95 | area= large | maxDets=100 ]",
               "AR @[ IoU=0.50:0.95 | area=   all | maxDets=  1 ]", "AR @[ IoU=0.50:0.95 | area=   all | maxDets= 10 ]",
               "AR @[ IoU=0.50:0.95 | area=   all | maxDets=100 ]", "AR @[ IoU=0.50:0.95 | area= small | maxDets=100 ]",
               "AR @[ IoU=0.50:0.95 | area=medium | maxDets=100 ]", "AR @[ IoU=0.50:0.95 | area= large | maxDets=100 ]"]


def coco_box_stats(detections, ground_truth, device):
    """The 12 COCO box statistics over every rank's images (detections / ground truth are gathered first);
    the IoU runs on `device` when it is a GPU (coco_eval.CocoBoxEvaluator)."""
    from .coco_eval import CocoBoxEvaluator
    if utils.is_dist_avail_and_initialized():
        merged_d, merged_g = {}, {}
        for d, g in utils.all_gather((detections, ground_truth)):
            merged_d.update(d)
            merged_g.update(g)
        detections, ground_truth = merged_d, merged_g
    ev = CocoBoxEvaluator(ground_truth, device=device if device.type == "cuda" else None)
    ev.update({k: v for k, v in detections.items() if k in ground_truth})
    stats = ev.summarize()
    if utils.is_main_process():
        for name, v in zip(_STAT_NAMES, stats):
            print(" Average %s (%s) %s = %0.3f" % ("Precision" if name.startswith("AP") else "Recall   ", name[:2], name[3:], v))
    return stats
