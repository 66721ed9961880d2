"""Epoch engines -- drop-in for the hot-path parts of the reference's engine.py
(`train_one_epoch` :30-167, `evaluate` :220-416, ensemble routers :171-218).

What one training step does, in order (reference engine.py:74-158):
  H2D as fp16 -> blur_image_list (HIP) -> expand_targets (HIP) -> .float() -> per-image norm statistics
  -> model(images, targets, newMeans, newSTDs) -> loss all-reduce for logging -> backward (DDP all-reduce
  over RCCL overlapped with it) -> SGD step -> warm-up LR step.
Differences from the reference are confined to mechanics: images cross PCIe as the loader's pinned fp32 and
are converted to Half on the device (same round-to-nearest-even as `.half()` on the host), the blur needs no
host synchronisation, and the logged losses reach the host through pinned memory without draining the stream.
`evaluate` returns the reference's `CocoEvaluator` surface (`.coco_eval["bbox"].stats`) computed by the
in-repo `coco_eval.py` (no pycocotools), see EvaluationResult.
"""
import math
import os
import sys
import time

import torch

from . import utils
from .models import blur_functions, net_transforms


def _to_device(images_CPU, targets, blur_dicts, device, blurring, want_tables=True, defer=False):
    """reference engine.py:79-98: images as Half, PSFs via torch.HalfTensor(ndarray).
    Returns (images, targets, psfs, thetas, lambda1s, lambda2s, tables): `tables` are the batch's tap tables, being
    compacted on the side stream (None when nothing will consume them, on the CPU, or for PSFs of mixed shapes) --
    the caller hands them to `blur_image_list(tables=)` and `expand_targets(tables=)`.  `defer`: return (that tuple, event) and
    leave the hand-over to `_adopt` (the evaluation loop stages the NEXT batch while the detector runs on this one).
    On a GPU the whole batch is staged on the device's side stream: the fp32 tensors the DataLoader pinned are uploaded
    as they are (true asynchronous copies) and rounded to Half on the device -- the same round-to-nearest-even as
    `.half()` on the host -- while the main stream is still busy with the previous step; the main stream then waits for
    one event.  (Issued on the main stream, the 102 MB of a b = 8 batch at 800 x 1333 sit in front of the step: ~4 ms.)"""
    cuda = device.type == "cuda"
    if not cuda:
        return _stage(images_CPU, targets, blur_dicts, device, blurring, want_tables, False)
    from . import blur_ops
    main = torch.cuda.current_stream(device)
    side = blur_ops.side_stream(device)
    with torch.cuda.stream(side):
        out = _stage(images_CPU, targets, blur_dicts, device, blurring, want_tables, True)
        staged = torch.cuda.Event()
        staged.record(side)
    if defer:
        return out, staged
    return _adopt(out, staged, device)


def _adopt(out, staged, device):
    """The main stream takes over a batch staged on the side stream (`_to_device(defer=True)`)."""
    main = torch.cuda.current_stream(device)
    main.wait_event(staged)
    images_GPU, targets_GPU, psfs_GPU, thetas, _, _, _ = out
    # allocated from the side stream's pool, consumed on the main stream
    for t in images_GPU + [v for tg in targets_GPU for v in tg.values() if isinstance(v, torch.Tensor)] + (psfs_GPU or []) \
            + ([thetas] if thetas is not None else []):
        if t.is_cuda:
            t.record_stream(main)
    return out


def _stage(images_CPU, targets, blur_dicts, device, blurring, want_tables, cuda):
    if cuda:
        images_GPU = [image.to(device, non_blocking=True).half() for image in images_CPU]
    else:
        images_GPU = [image.half() for image in images_CPU]
    targets_GPU = [{k: (v.to(device, non_blocking=True) if isinstance(v, torch.Tensor) else v) for k, v in t.items()} for t in targets]
    psfs_GPU = thetas = l1 = l2 = tables = None
    if blurring:
        # PSFs: torch.HalfTensor(ndarray) (float64 -> float32 -> float16) into ONE pinned staging block per
        # batch and one asynchronous copy; the list entries are views of it.  Entries of another shape (the
        # 1-element [0] of a not-blurred image, reference transforms.py:455) travel on their own.
        halves = [torch.HalfTensor(bd["psf"]) for bd in blur_dicts]
        shapes = {tuple(h.shape) for h in halves if h.dim() == 2}
        psfs_GPU = [None] * len(halves)
        if len(shapes) == 1 and cuda:
            shp = next(iter(shapes))
            sel = [i for i, h in enumerate(halves) if tuple(h.shape) == shp]
            stage = torch.empty((len(sel),) + shp, dtype=torch.float16, pin_memory=True)
            for k, i in enumerate(sel):
                stage[k].copy_(halves[i])
            block = stage.to(device, non_blocking=True)
            for k, i in enumerate(sel):
                psfs_GPU[i] = block[k]
        for i, h in enumerate(halves):
            if psfs_GPU[i] is None:
                psfs_GPU[i] = h.to(device, non_blocking=True)
        # tap compaction follows the PSF upload on the same (side) stream, only when a blur or a box growth will wait
        # for it: it overlaps whatever the previous batch still has on the main stream
        active = [p for p, bd in zip(psfs_GPU, blur_dicts) if bd["blurring"]]
        if want_tables and cuda and active and len({tuple(p.shape) for p in active}) == 1 and active[0].dim() == 2 \
                and active[0].shape[0] in (128, 256):
            from . import blur_ops
            # fp16 images only (what the blur of this engine sees); the box growth reads the tap list, not the segments
            large = blur_ops.large_window_pays(blur_dicts, len(active))
            tables = blur_ops.compact_psfs_ahead(active, normalize=True, after_current=True, large_window=large)
        # theta / lambda1 / lambda2: one [3, B] pinned tensor, one copy
        scal = torch.tensor([[bd["theta_rad"] for bd in blur_dicts], [bd["scale_factor_lambda1"] for bd in blur_dicts],
                             [bd["scale_factor_lambda2"] for bd in blur_dicts]], dtype=torch.float16)
        if cuda:
            scal = scal.pin_memory().to(device, non_blocking=True)
        thetas, l1, l2 = scal[0], scal[1], scal[2]
    return images_GPU, targets_GPU, psfs_GPU, thetas, l1, l2, tables


class _StagedAhead(object):
    """The evaluation loop's look-ahead.  Iterates (batch, prepared) over a loader in loader order; `advance()`, called by the loop
    just before it launches the detector on the current batch, (1) runs `prepare` -- everything of the NEXT batch that needs no
    host synchronisation: blur, box growth, float conversion, the blur estimator -- so that work is queued on the GPU before the
    host blocks on the current detections, and (2) pulls the batch after that from the loader and queues its upload on the side
    stream (`_to_device(defer=True)`), an iteration before its `prepare` will wait for it.  `prepare(batch, staged)` gets the staged
    tuple of `_to_device` (adopted by the main stream already); `may_prepare` False keeps (1) inline with the iteration (host
    random draws in `prepare`, the CPU)."""

    def __init__(self, loader, device, blurring, want_tables, prepare, may_prepare=True):
        self.it, self.device, self.blurring, self.want_tables = iter(loader), device, blurring, want_tables
        self.prepare, self.may_prepare = prepare, may_prepare and device.type == "cuda"
        self.staged, self.prepared, self.exhausted = [], None, False

    def _pull(self):
        if self.exhausted:
            return
        try:
            batch = next(self.it)
        except StopIteration:
            self.exhausted = True
            return
        images_CPU, targets_CPU, blur_dicts = batch
        self.staged.append((batch, _to_device(images_CPU, targets_CPU, blur_dicts, self.device, self.blurring, want_tables=self.want_tables,
                                              defer=self.device.type == "cuda")))

    def _prepare_first(self):
        batch, staged = self.staged.pop(0)
        if self.device.type == "cuda":
            staged = _adopt(*staged, self.device)
        return batch, self.prepare(batch, staged)

    def advance(self, more=True):
        """`more` False: the loop will stop after the current batch (early_stop): nothing further is prepared."""
        if not more:
            return
        if not self.staged:
            self._pull()
        if self.may_prepare and self.prepared is None and self.staged:
            self.prepared = self._prepare_first()
        if len(self.staged) < 1:
            self._pull()

    def __iter__(self):
        while True:
            if self.prepared is not None:
                cur, self.prepared = self.prepared, None
            else:
                if not self.staged:
                    self._pull()
                if not self.staged:
                    return
                cur = self._prepare_first()
            yield cur


def _estimate(blur_estimator, x, graphed):
    """The blur estimator's logits; on a GPU through a HIP graph per input shape (a clone: the graph's output buffer is
    overwritten by the next replay)."""
    core = getattr(blur_estimator, "module", blur_estimator)
    if not graphed or not getattr(core, "graph_safe", False):      # a module that does not say so may synchronise in forward
        return blur_estimator(x)
    cache = core.__dict__.get("_dib_graphs")
    if cache is None:
        from .graphs import GraphCache
        cache = core.__dict__["_dib_graphs"] = GraphCache(blur_estimator)
    # the graphs read the estimator's parameters and statistics through their live pointers (in-place updates are seen);
    # storage that was REPLACED since the capture (.to(), .half()) leaves them dangling: start over
    tensors = core.__dict__.get("_dib_graph_tensors")
    if tensors is None:                  # walking the module tree costs 0.35 ms per call: the list is kept, the pointers are compared
        tensors = core.__dict__["_dib_graph_tensors"] = list(core.parameters()) + list(core.buffers())
    ptrs = tuple(t.data_ptr() for t in tensors)
    if core.__dict__.get("_dib_graph_ptrs") != ptrs:
        if "_dib_graph_ptrs" in core.__dict__:
            cache.clear()
            core.__dict__.pop("_dib_graph_tensors")      # parameters may have been re-registered
        core.__dict__["_dib_graph_ptrs"] = ptrs
    return cache(x).clone()


# Opt-in (DIB_FUSE_BLUR_EPILOGUE=1, or set the flag): the training loop leaves the blur to the model's input transform, which
# runs it TOGETHER with its float conversion + normalisation + zero-padded batch as one launch (blur_ops.sparse_blur_normalized)
# when no image of the batch needs a resize -- the blurred fp16 batch then never exists (reference
# engine.py:101, :107-110 + models/net_transforms.py:112-121, :238-247).  Bit-identical to the default path, which stays the
# default: the reference blurs COCO at native size and resizes afterwards, where this does not apply and changes nothing.
FUSE_BLUR_EPILOGUE = os.environ.get("DIB_FUSE_BLUR_EPILOGUE") == "1"


def _postpone_blur(model, images_GPU, blur_dicts, psfs_GPU, tables, acc_mode=0):
    """Hands the batch's blur to the model's input transform (`pending_blur`, consumed by its next forward) instead of launching
    it.  False when that path is not available (no fused transform, tables that do not belong to the batch)."""
    tf = getattr(getattr(model, "module", model), "transform", None)
    if tf is None or not getattr(tf, "fused", False) or tables is None:
        return False
    idx = [i for i, bd in enumerate(blur_dicts) if bd["blurring"]]
    if not idx or tables.count != len(idx) or tables.K != psfs_GPU[idx[0]].shape[0]:
        return False
    if any(psfs_GPU[i].dtype != images_GPU[i].dtype for i in idx):      # blur_image_list would convert those PSFs first
        return False
    index = [-1] * len(images_GPU)
    for k, i in enumerate(idx):
        index[i] = k
    order = None
    try:
        order = sorted(range(len(images_GPU)), key=lambda i: -int(blur_dicts[i]["psf_taps"]) if blur_dicts[i]["blurring"] else 1)
    except KeyError:
        pass
    tf.pending_blur = (index, tables, acc_mode, order)
    return True


def _tables_128(tables):
    """expand_targets refuses PSFs that are not 128 wide with the reference's own exception (utils.py:369-370): let it
    see the PSFs, not tables of another canvas."""
    return tables if tables is not None and tables.K == 128 else None


def _to_float(images_GPU, model, device):
    """reference engine.py:107-110: `image.float()` per image (images that took the JPEG round trip come back on the
    host, transforms.py:492, hence the `.to(device)`).  When the model's input transform has the fused epilogue and
    every image is a Half tensor on the GPU, the conversion is left to it: float conversion, normalisation and batch
    padding then happen in one kernel (models/net_transforms.py) instead of one pass each."""
    m = getattr(model, "module", model)
    tf = getattr(m, "transform", None)
    if tf is not None and getattr(tf, "fused", False) and all(i.is_cuda and i.dtype == torch.float16 and i.dim() == 3 for i in images_GPU):
        return images_GPU
    return [image.float().to(device) for image in images_GPU]


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
    deferred = None
    # the update guard of the GPU path (see the loop): only torch's fused SGD takes it; with another optimizer (--foreach_sgd,
    # a caller's own) a non-finite loss still stops the run one step late, that step's update applied
    guard = None
    if getattr(optimizer, "_dib_found_inf", False):
        # an earlier epoch left through an exception the caller handled (OOM retry, KeyboardInterrupt) and never reached the
        # clean-up behind the loop: the attribute is this module's (tagged), and a stale non-zero tensor would make every fused
        # step from here on a silent no-op
        if hasattr(optimizer, "found_inf"):
            del optimizer.found_inf
        optimizer._dib_found_inf = False
    if isinstance(optimizer, torch.optim.SGD) and all(g.get("fused") for g in optimizer.param_groups) and not hasattr(optimizer, "found_inf"):
        guard = torch.empty(0)
    for images_CPU, targets, blur_dicts in metric_logger.log_every(data_loader, print_freq, header):
        images_GPU, targets_GPU, psfs_GPU, thetas, l1, l2, tables = _to_device(
            images_CPU, targets, blur_dicts, device, blur_train, want_tables=gpu_blur or expand_target_boxes)
        postponed = False
        if gpu_blur and blur_train:
            if FUSE_BLUR_EPILOGUE and not (add_noise or add_block or add_jpeg_artifact):
                postponed = _postpone_blur(model, images_GPU, blur_dicts, psfs_GPU, tables)
            if not postponed:
                blur_functions.blur_image_list(images_GPU, blur_dicts, psfs_GPU=psfs_GPU, add_noise=add_noise,
                                               noise_level=noise_level, add_block=add_block,
                                               add_jpeg_artifact=add_jpeg_artifact, jpeg_compressor=jpeg_compressor, tables=tables)
        if expand_target_boxes and blur_train:
            targets_GPU = utils.expand_targets(targets_GPU, blur_dicts, psfs_GPU, images_GPU, tables=_tables_128(tables))
        images_GPU = _to_float(images_GPU, model, device)
        norm_means, norm_stds = utils.get_norm_params(blur_dicts, use_custom_image_norm)

        try:
            loss_dict = model(images_GPU, targets_GPU, thetas=thetas, lambda1s=l1, lambda2s=l2, newMeans=norm_means, newSTDs=norm_stds)
        finally:
            if postponed:      # a forward pass that raised in front of the transform must not leave the blur behind for the next batch
                getattr(model, "module", model).transform.__dict__.pop("pending_blur", None)
        losses = sum(loss for loss in loss_dict.values())

        loss_dict_reduced = utils.reduce_dict(loss_dict)                 # logging only
        losses_reduced = sum(loss for loss in loss_dict_reduced.values())
        logged = _to_host_async(losses_reduced, loss_dict_reduced)

        lr_before = optimizer.param_groups[0]["lr"]        # what the reference's writer logs: read ahead of the warm-up step (:142)
        checked = logged[2] is None
        if checked:
            # tensors on the host: nothing to overlap -- the reference's order exactly (scalars, finite-loss check, THEN the update)
            _log_step(metric_logger, writer, logged, lr_before, None, iteration_count, epoch, len(data_loader), print_freq, update_meters=False)
        elif guard is not None:
            # On the GPU the loss is read one step late (below); the UPDATE must not be: torch's fused SGD skips its step when the
            # `found_inf` tensor it is handed is non-zero (the hook GradScaler uses), so a non-finite loss -- and every step
            # behind it, until the host has seen it and exits -- leaves the weights as they were, which is the state the
            # reference exits in (engine.py:145-148 sits in front of zero_grad / backward / step).  No host synchronisation.
            bad = torch.isfinite(losses_reduced.detach()).logical_not().to(torch.float32).reshape(1)
            guard = bad if guard.numel() == 0 else torch.maximum(guard, bad)
            optimizer.found_inf = guard
            optimizer._dib_found_inf = True
        optimizer.zero_grad()
        losses.backward()
        optimizer.step()
        if lr_scheduler is not None:
            lr_scheduler.step()
        # Logging and the finite-loss check read the PREVIOUS step's numbers, which `_to_host_async` sent to pinned host
        # memory right behind that step's forward pass: reading them waits for that copy's event only, long complete, and
        # this step's forward, backward and update are all enqueued already.  (The reference reads the current loss
        # between forward and backward, engine.py:131-148: the host waits for the forward pass and the GPU idles while
        # the backward pass is issued.  A plain `.item()` here would be no better: its copy queues up behind everything
        # enqueued so far and drains the stream once per step -- 3-4 ms of idle GPU until the next step's first kernels.)
        # A non-finite loss still stops the run, one step later (its update and the next one skipped: `guard` above); the last
        # step is checked behind the loop.
        if deferred is not None:
            _log_step(*deferred)
        deferred = (metric_logger, writer, logged, lr_before, optimizer.param_groups[0]["lr"],
                    iteration_count, epoch, len(data_loader), print_freq, True, not checked)
        # reference :160-162, literally: `early_stop=False` (the signature's default) compares as 0 and ends the epoch
        # after two iterations; train.py passes --early_stop (None unless given).  The iteration that breaks is
        # logged and checked but, as in the reference, not entered into the meters.
        if early_stop is not None and iteration_count > early_stop:
            _log_step(*deferred[:9], update_meters=False, check=deferred[10])
            deferred = None
            break
        iteration_count += 1
    if deferred is not None:
        _log_step(*deferred)
    if guard is not None and hasattr(optimizer, "found_inf"):
        del optimizer.found_inf
        optimizer._dib_found_inf = False
    return metric_logger


def _to_host_async(total, loss_dict):
    """(keys, values on the host, event): the scalars of one step on their way to pinned memory, no stream drain."""
    keys = list(loss_dict.keys())
    vals = torch.stack([total.detach()] + [loss_dict[k].detach() for k in keys]).float()
    if not vals.is_cuda:
        return keys, vals, None
    host = torch.empty(vals.shape, dtype=vals.dtype, pin_memory=True)
    host.copy_(vals, non_blocking=True)
    done = torch.cuda.Event()
    done.record()
    return keys, host, done


def _log_step(metric_logger, writer, logged, lr_before, lr_after, iteration_count, epoch, n_iter, print_freq, update_meters=True, check=True):
    """reference engine.py:131-158: TensorBoard scalars every 500 iterations (learning rate as it was BEFORE this
    iteration's warm-up step), exit on a non-finite loss (`check`), meters (learning rate AFTER the step; `update_meters`)."""
    keys, host, done = logged
    if done is not None:
        done.synchronize()
    values = host.tolist()
    loss_value, loss_dict_reduced = values[0], dict(zip(keys, values[1:]))
    if check and iteration_count % 500 == 0 and writer is not None and utils.is_main_process() and iteration_count % print_freq == 0:
        step = iteration_count + epoch * n_iter
        for key, v in loss_dict_reduced.items():
            writer.add_scalar("losses/" + key, v, step)
        writer.add_scalar("losses/overallLoss", loss_value, step)
        writer.add_scalar("learningRate", lr_before, step)
    if check and not math.isfinite(loss_value):                      # reference :145-148
        print("Loss is {}, stopping training".format(loss_value))
        print(loss_dict_reduced)
        sys.exit(1)
    if update_meters:
        metric_logger.update(loss=loss_value, **loss_dict_reduced)
        metric_logger.update(lr=lr_after)


# ---- ensemble routing (reference engine.py:171-218) ------------------------------------------------

def get_network_index_to_use_oracle(blur_dicts, model_indices):
    """Ground-truth routing from the blur_dicts: a sharp image or a very short exposure -> net 0, blur type
    P1/P2/P3 (`param_index` 0/1/2) -> nets 1/2/3.  The first dict that decides wins; a blurred dict of any other type
    decides nothing and the next one is asked (None when none is left) -- the reference's loop, engine.py:171-191,
    pinned over a grid of batches by tests/test_detector_pins.py."""
    for bd in blur_dicts:
        if not (bd["blurring"] and bd["param_index"] is not None):
            return model_indices[0]
        if bd["fraction_index"] == -1:
            return model_indices[0]
        if bd["param_index"] in (0, 1, 2):
            return model_indices[bd["param_index"] + 1]
    return None


def get_network_index_to_use_blur_estimator(blur_estimation, model_indices):
    """16-way estimator: class 0 = sharp, 1-5 = P1 x E0..E4, 6-10 = P2, 11-15 = P3."""
    k = int(blur_estimation.argmax())
    return model_indices[0 if k == 0 or k > 15 else 1 + (k - 1) // 5]


def get_network_index_to_use_blur_estimator_LEHE(blur_estimation, model_indices):
    """4-way estimator (--LEHE): class 0 -> low-exposure net, 1/2/3 -> P1/P2/P3 high-exposure nets."""
    k = int(blur_estimation.argmax())
    return model_indices[k if k in (1, 2, 3) else 0]


class EvaluationResult(object):
    """What `evaluate` returns: the reference's `CocoEvaluator` surface (`.coco_eval["bbox"].stats`, `.coco_gt`,
    `.img_ids`; reference engine.py:416, train.py:350-387) plus the raw material of the run as attributes and,
    for callers that index it, as keys: "detections" {image_id: {boxes, labels, scores}}, "targets"
    {image_id: xywh boxes as written into the ground truth}, "routes" [model index per batch], "meters",
    "coco_stats" (the 12 numbers, or None when the dataset carries no annotations)."""

    def __init__(self, coco_evaluator, **extra):
        self._ce = coco_evaluator
        self._extra = extra
        for k, v in extra.items():
            setattr(self, k, v)

    def __getattr__(self, name):            # CocoEvaluator surface: coco_eval, coco_gt, img_ids, iou_types, ...
        return getattr(self.__dict__["_ce"], name)

    def __getitem__(self, key):
        return self._extra[key]

    def __contains__(self, key):
        return key in self._extra

    def keys(self):
        return self._extra.keys()


@torch.no_grad()
def evaluate(model, data_loader, device, distributed_mode=False, early_stop=None, vanilla_eval=False, blurring_images=False,
             gpu_blur=False, expand_target_boxes=False, deblur_first=False, deblurer=None, use_custom_image_norm=False,
             use_ensemble=False, ensemble_models=None, blur_estimator=None, add_noise=False, noise_level=0.001, add_block=False,
             add_jpeg_artifact=False, image_output_folder=None, LEHE=False, epoch_number=None):
    """reference engine.py:220-416.  Runs the detector (or the routed ensemble) over the loader, scores the
    detections against the dataset's COCO ground truth (boxes replaced by the expanded ones under
    `expand_target_boxes`, :325-342) and returns the evaluator (see EvaluationResult)."""
    from .coco_eval import CocoEvaluator
    from .coco_utils import get_coco_api_from_dataset
    if deblur_first:
        raise NotImplementedError("--deblur_first is outside the built path (SURVEY.md section 2)")
    jpeg_compressor = None
    if add_jpeg_artifact:                                                # reference :248-254
        from .models.jpeg import DiffJPEG
        jpeg_compressor = DiffJPEG(height=100, width=100, differentiable=False, quality=10).to(device)
    n_threads = torch.get_num_threads()
    torch.set_num_threads(1)
    batcher = None
    # batch size 1 is launch-bound: the detectors' static trunk and the estimator replay as HIP graphs (graphs.py)
    graphed = device.type == "cuda" and not os.environ.get("DIB_NO_GRAPHS")
    for m in (ensemble_models if use_ensemble else [model]):
        core = getattr(m, "module", m)
        if hasattr(core, "graph_inference"):
            core.graph_inference = graphed
    if use_ensemble:
        for m in ensemble_models:
            m.eval()
        if blur_estimator is not None:                                   # reference :259-265
            batcher = net_transforms.GeneralizedRCNNTransform(800, 1333, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], crop_images=True)
            blur_estimator.eval()
    else:
        model.eval()
    metric_logger = utils.MetricLogger(delimiter="  ")
    coco = get_coco_api_from_dataset(data_loader.dataset)                # reference :271-273
    coco_evaluator = CocoEvaluator(coco, ["bbox"], device=device if device.type == "cuda" else None)
    detections, gt_boxes, routes = {}, {}, []
    count = faulty_boxes = total_boxes = 0
    # On a GPU the per-image COCO matching (host, ~2 ms) runs on ONE worker thread, in submission order, while the main
    # thread waits for the next image's detector: the ground-truth boxes of an image are written before its update is
    # submitted, and different images touch different annotations.  `evaluator_time` then measures the hand-over.
    scorer, pending = None, []
    if device.type == "cuda" and not os.environ.get("DIB_NO_GRAPHS"):
        from concurrent.futures import ThreadPoolExecutor
        scorer = ThreadPoolExecutor(max_workers=1)
        score_stream = torch.cuda.Stream(device=device)

        def score(res):           # the box-IoU kernel of the matching on its own stream: never queued behind the detector
            with torch.cuda.stream(score_stream):
                coco_evaluator.update(res)
    try:
        # On a GPU the loop works one batch ahead (_StagedAhead): while the host waits for the detections of image i, image i + 1's
        # blur, box growth and blur-estimator pass are already queued, and image i + 2's upload runs on the side stream.  Only
        # work without host random draws goes ahead (--add_noise / --add_block / --add_jpeg_artefacts draw in `blur_image_list`).
        def prepare(batch, staged):
            images_CPU, targets_CPU, blur_dicts = batch
            images_GPU, targets_GPU, psfs_GPU, thetas, l1, l2, tables = staged
            if gpu_blur and blurring_images:
                blur_functions.blur_image_list(images_GPU, blur_dicts, psfs_GPU=psfs_GPU, add_noise=add_noise, noise_level=noise_level,
                                               add_block=add_block, add_jpeg_artifact=add_jpeg_artifact,
                                               jpeg_compressor=jpeg_compressor, tables=tables)
            if expand_target_boxes and blurring_images:
                targets_GPU = utils.expand_targets(targets_GPU, blur_dicts, psfs_GPU, images_GPU, tables=_tables_128(tables))
            images_GPU = _to_float(images_GPU, ensemble_models[0] if use_ensemble else model, device)
            est = None
            if use_ensemble and blur_estimator is not None:                  # reference :354-366 (the routing itself: below)
                batched, _ = batcher(images_GPU, None)
                est = _estimate(blur_estimator, batched.tensors, graphed)
            return images_GPU, targets_GPU, blur_dicts, thetas, l1, l2, est

        def submit(ids, outputs, gt_xywh, started):
            """detections of one batch (CPU tensors) -> result tables + the evaluator (reference :376-392)"""
            model_time = time.time() - started
            res = {}
            for image_id, o, g in zip(ids, outputs, gt_xywh):
                res[image_id] = o
                detections[image_id] = o
                gt_boxes[image_id] = g
            evaluator_time = time.time()
            if scorer is None:
                coco_evaluator.update(res)                                   # reference :388-392
            else:
                pending.append(scorer.submit(score, res))                    # scored while the GPU runs the next image
            metric_logger.update(model_time=model_time, evaluator_time=time.time() - evaluator_time)

        def finalize(entry):
            core_, handle_, ids_, gt_, started_ = entry
            submit(ids_, core_.finish(handle_), gt_, started_)

        # models whose internal warp is on need thetas / lambdas in forward: they stay on the plain loop
        pipelined = (graphed and not os.environ.get("DIB_NO_PIPELINE")
                     and not any(getattr(getattr(m, "module", m), "warp_internally", False) and blurring_images
                                 for m in (ensemble_models if use_ensemble else [model])))
        trunk_running, in_flight = None, []

        def flush():
            """empties the pipeline, in image order: heads of the image whose trunk is running, then everything on its way"""
            nonlocal trunk_running, in_flight
            if trunk_running is not None:
                torch.cuda.synchronize()
                if trunk_running[0].launch_heads(trunk_running[1]) is None:
                    raise RuntimeError("evaluate: the detector left the pipelined path between its trunk and its heads")
                in_flight.append(trunk_running)
                trunk_running = None
            if in_flight:
                torch.cuda.synchronize()
                for entry in in_flight:
                    finalize(entry)
                in_flight = []

        ahead = _StagedAhead(metric_logger.log_every(data_loader, 100, "Test:"), device, blurring_images, gpu_blur or expand_target_boxes,
                             prepare, may_prepare=not (add_noise or add_block or add_jpeg_artifact))
        for _, (images_GPU, targets_GPU, blur_dicts, thetas, l1, l2, est) in ahead:
            if device.type == "cuda":
                torch.cuda.synchronize()
            model_time = time.time()
            if expand_target_boxes and blurring_images:
                # the expanded boxes replace the ground truth's, annotation k <- target box k (reference :325-342,
                # index-wise: where the target dropped a crowd / degenerate annotation the tail keeps its box)
                for target in targets_GPU:
                    boxes = utils.convert_to_xywh(target["boxes"]).cpu().numpy().tolist()
                    anns = coco_evaluator.coco_gt.imgToAnns[int(target["image_id"].item())]
                    for k, ann in enumerate(anns):
                        total_boxes += 1
                        if k < len(boxes):
                            ann["bbox"] = boxes[k]
                        elif k + 1 == len(anns):             # counted once per image, at its last annotation (reference :336-341)
                            faulty_boxes += 1
                        else:
                            print("Faulty " + str(len(anns) - k) + " times over.")
            norm_means, norm_stds = utils.get_norm_params(blur_dicts, use_custom_image_norm)

            if use_ensemble:                                                 # reference :354-366
                idx = list(range(len(ensemble_models)))
                if blur_estimator is None:
                    k = get_network_index_to_use_oracle(blur_dicts, idx)
                else:
                    k = (get_network_index_to_use_blur_estimator_LEHE if LEHE else get_network_index_to_use_blur_estimator)(est, idx)
                model = ensemble_models[k]
                routes.append(k)
            more = early_stop is None or count + 1 <= early_stop
            core = getattr(model, "module", model)
            ids = [int(t["image_id"]) if "image_id" in t else count for t in targets_GPU]
            handle = None
            if not (pipelined and hasattr(core, "launch_trunk")):
                flush()                                                      # a detector without the split forward pass: images stay in order
            else:
                # Pipelined (GPU, graph inference): the queue is empty here (the synchronisation above).  Queue, in this order and
                # without waiting for any of it: the RoI heads + detections of the PREVIOUS image (its trunk has finished: reading
                # its proposal counts costs nothing) with their copy to pinned memory, this image's trunk, the next image's blur /
                # estimator pass and the upload of the one after; then turn the detections that were copied before the
                # synchronisation into evaluator updates while the GPU works.  Same tensors through the same kernels as the
                # plain loop below (tests/test_full_size_gpu.py compares the two).
                ready, in_flight = in_flight, []
                gt_host = [utils.convert_to_xywh(t["boxes"]).cpu() for t in targets_GPU]      # before anything is queued: no wait
                if trunk_running is not None:
                    if trunk_running[0].launch_heads(trunk_running[1]) is None:
                        raise RuntimeError("evaluate: the detector left the pipelined path between its trunk and its heads")
                    in_flight.append(trunk_running)
                    trunk_running = None
                handle = core.launch_trunk(images_GPU, killWarp=not blurring_images, newMeans=norm_means, newSTDs=norm_stds)
                if handle is not None:
                    trunk_running = (core, handle, ids, gt_host, model_time)
                    ahead.advance(more=more)
                    for entry in ready:
                        finalize(entry)
                else:                                                        # this detector does not take the path: drain, then as below
                    in_flight = ready + in_flight
                    flush()
            if handle is None:
                # the look-ahead step runs right after the detector's trunk has been launched (models/generalized_rcnn.py calls
                # the hook between the graph replay and its first wait); a model without that hook gets it before the call
                hooked = graphed and getattr(core, "graph_inference", False)
                if hooked:
                    core.__dict__["_after_trunk_launch"] = lambda: ahead.advance(more=more)
                else:
                    ahead.advance(more=more)
                if blurring_images:
                    outputs = model(images_GPU, thetas=thetas, lambda1s=l1, lambda2s=l2, newMeans=norm_means, newSTDs=norm_stds)
                else:
                    outputs = model(images_GPU, killWarp=True, newMeans=norm_means, newSTDs=norm_stds)
                if hooked and core.__dict__.pop("_after_trunk_launch", None) is not None:
                    ahead.advance(more=more)                                  # the forward pass took a path without the hook
                outputs = [{k: v.to("cpu") for k, v in t.items()} for t in outputs]
                submit(ids, outputs, [utils.convert_to_xywh(t["boxes"]).cpu() for t in targets_GPU], model_time)
            count += 1
            if early_stop is not None and count > early_stop:
                break
        flush()                                                              # the pipeline's tail
    finally:                                                             # also on an error in the loop: no stray thread, thread count restored
        if scorer is not None:
            scorer.shutdown(wait=True)
        torch.set_num_threads(n_threads)
    for f in pending:
        f.result()                                                       # re-raises anything the scoring thread hit
    metric_logger.synchronize_between_processes()
    print("Averaged stats:", metric_logger)
    print("Number of Faulty boxes: " + str(faulty_boxes) + " Total number of boxes: " + str(total_boxes))
    coco_evaluator.synchronize_between_processes()                       # a collective on every rank, shards empty or not
    coco_evaluator.accumulate()
    stats = coco_evaluator.summarize() if utils.is_main_process() else coco_evaluator.coco_eval["bbox"].summarize()
    has_gt = len(coco_evaluator.coco_gt.dataset.get("annotations", [])) > 0
    return EvaluationResult(coco_evaluator, detections=detections, targets=gt_boxes, routes=routes, meters=metric_logger,
                            coco_stats=stats if has_gt else None)
