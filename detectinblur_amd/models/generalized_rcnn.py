"""`GeneralizedRCNN` with the reference's extended forward signature (reference
models/generalized_rcnn.py:78): per-image normalisation statistics travel with the call
(`newMeans` / `newSTDs`); `thetas / lambda1s / lambda2s / killWarp` drive the "squint" warper
(`--warp_in_model`, models/warper.py; reference generalized_rcnn.py:131-141).
"""
from collections import OrderedDict

import torch
from torch import nn


class GeneralizedRCNN(nn.Module):
    def __init__(self, backbone, rpn, roi_heads, transform, warp_internally=False):
        super().__init__()
        self.transform, self.backbone, self.rpn, self.roi_heads = transform, backbone, rpn, roi_heads
        self.warp_internally = warp_internally
        if warp_internally:
            from .warper import Warper
            self.warper = Warper()
        # Inference through a HIP graph of the static trunk (backbone + FPN + RPN head + proposal filtering): off by
        # default, switched on by engine.evaluate on a GPU (graphs.py).  Same kernels, same results, ~300 launches fewer
        # for the interpreter per image.
        self.graph_inference = False

    def _trunk(self, x):
        feats = self.backbone(x)
        if isinstance(feats, torch.Tensor):
            feats = OrderedDict([("0", feats)])
        self._feat_names = list(feats.keys())
        fl = list(feats.values())
        from .net_transforms import ImageList
        anchors = self.rpn.anchor_generator(ImageList(x, [None] * x.shape[0]), fl)[0]
        # `anchors` comes out of the generator's one-entry cache and is read by a captured graph through its raw pointer: it
        # travels with the outputs so that the StaticGraph keeps it alive after the cache has moved on to another geometry
        return tuple(fl) + tuple(self.rpn.propose_static(fl, anchors, self._sizes[x.shape[0]])) + (anchors,)

    def _sync_graphs_with_weights(self, cache):
        """A captured trunk reads the FPN / RPN parameters through their live pointers and the trunk's convolutions through the
        cached batch-norm folds (backbone._folded).  Before every replay: (1) if any parameter or buffer of the trunk was
        REPLACED (another storage: .to(), .half(), a re-assigned .data) the graphs hold dangling pointers -- drop them; (2) folds
        whose convolution or statistics changed IN PLACE since (an optimizer step between two evaluations, load_state_dict)
        are rewritten in place, so the graphs read current values.  ~300 attribute reads per call, nothing on the device
        unless something changed."""
        from .backbone import refresh_folded
        tensors = self.__dict__.get("_trunk_tensors")
        if tensors is None:
            tensors = self.__dict__["_trunk_tensors"] = ([p for p in self.backbone.parameters()] + [b for b in self.backbone.buffers()]
                                                         + [p for p in self.rpn.parameters()])
        ptrs = tuple(t.data_ptr() for t in tensors)
        if self.__dict__.get("_trunk_ptrs") != ptrs:
            if "_trunk_ptrs" in self.__dict__:
                cache.clear()
                self.__dict__.pop("_trunk_tensors")         # parameters may have been re-registered
            for key in ("_dib_fold_watch", "_dib_fold_state"):      # refresh_folded's shortcut watches the old tensors
                self.backbone.__dict__.pop(key, None)
            self.__dict__["_trunk_ptrs"] = ptrs
        refresh_folded(self.backbone)

    def _forward_graphed(self, images, original_sizes):
        from ..graphs import GraphCache
        x = images.tensors
        n = x.shape[0]
        sizes = self.__dict__.setdefault("_sizes", {})
        if n not in sizes:
            sizes[n] = torch.zeros((n, 2), dtype=x.dtype, device=x.device)       # read by the captured graph: never reallocated
        host = torch.tensor([[float(s[1]), float(s[0])] for s in images.image_sizes], dtype=x.dtype).pin_memory()
        sizes[n].copy_(host, non_blocking=True)
        cache = self.__dict__.get("_trunk_graphs")
        if cache is None:
            cache = self.__dict__["_trunk_graphs"] = GraphCache(self._trunk)
        self._sync_graphs_with_weights(cache)
        outs = cache(x)
        hook = self.__dict__.pop("_after_trunk_launch", None)
        if hook is not None:
            # engine.evaluate's look-ahead: the trunk is queued, the host is about to wait for it -- the moment to queue the next
            # image's blur / estimator pass and the upload of the one after (one call per forward pass, then forgotten)
            hook()
        features = OrderedDict(zip(self._feat_names, outs[:-4]))
        proposals, _ = self.rpn.unpad(*outs[-4:-1])
        detections, _ = self.roi_heads(features, proposals, images.image_sizes, None)
        return self.transform.postprocess(detections, images.image_sizes, original_sizes)

    # ---- the inference forward pass in three parts, for engine.evaluate's pipelined loop -------------------------------------------
    # launch_trunk: transform + trunk graph replay (no synchronisation); launch_heads: proposals to the host (one small copy: call it
    # when the trunk has finished and it costs nothing), RoI heads, detections padded to detections_per_img rows, rescaled to the
    # original image and copied to pinned host memory -- all queued, nothing waited for; finish: slices the pinned rows once the
    # stream has drained.  Same tensors through the same kernels as `forward`; `None` from launch_trunk / launch_heads means
    # "not on this path" (training, CPU, internal warping, no graph inference, shapes the detection kernels do not take).
    def launch_trunk(self, images, killWarp=False, newMeans=None, newSTDs=None):
        if not (self.graph_inference and not self.training and not torch.is_grad_enabled()) or (self.warp_internally and not killWarp):
            return None
        if not all(isinstance(i, torch.Tensor) and i.is_cuda for i in images):
            return None
        from ..graphs import GraphCache
        original_sizes = [(int(img.shape[-2]), int(img.shape[-1])) for img in images]
        batch, _ = self.transform(images, None, newMeans, newSTDs)
        x = batch.tensors
        n = x.shape[0]
        sizes = self.__dict__.setdefault("_sizes", {})
        if n not in sizes:
            sizes[n] = torch.zeros((n, 2), dtype=x.dtype, device=x.device)
        host = torch.tensor([[float(s[1]), float(s[0])] for s in batch.image_sizes], dtype=x.dtype).pin_memory()
        sizes[n].copy_(host, non_blocking=True)
        cache = self.__dict__.get("_trunk_graphs")
        if cache is None:
            cache = self.__dict__["_trunk_graphs"] = GraphCache(self._trunk)
        self._sync_graphs_with_weights(cache)
        outs = cache(x)
        return {"images": batch, "original_sizes": original_sizes, "outs": outs, "feat_names": list(self._feat_names)}

    def launch_heads(self, handle):
        from .net_transforms import resize_boxes
        outs, batch = handle["outs"], handle["images"]
        features = OrderedDict(zip(handle["feat_names"], outs[:-4]))
        proposals, _ = self.rpn.unpad(*outs[-4:-1])
        padded, head_outputs = self.roi_heads.padded_detections(features, proposals, batch.image_sizes)
        if padded is None:
            # shapes the detection kernels do not take: finish this image the plain way (with its synchronisations) right here
            detections = self.roi_heads.postprocess_detections(head_outputs[0], head_outputs[1], proposals, batch.image_sizes)
            detections = self.transform.postprocess(detections, batch.image_sizes, handle["original_sizes"])
            handle["done"] = [{k: v.to("cpu") for k, v in d.items()} for d in detections]
            return handle
        ring = self.__dict__.setdefault("_pinned_ring", {"slot": 0, "bufs": {}})
        pins = []
        for (boxes, scores, labels, count), size, original in zip(padded, batch.image_sizes, handle["original_sizes"]):
            k = boxes.shape[0]
            boxes = resize_boxes(boxes, size, original)                          # transform.postprocess, on the padded rows
            # pinned rows are reused `depth` images later: two batches are in flight in engine.evaluate's pipeline (the heads of batch
            # j are queued before batch j - 1 is read), so the ring holds three batches' worth, and at least 8
            depth = ring["depth"] = max(ring.get("depth", 8), 3 * len(padded) + 2)
            slot = ring["slot"] = (ring["slot"] + 1) % depth
            buf = ring["bufs"].get((slot, k))
            if buf is None:
                buf = ring["bufs"][(slot, k)] = (torch.empty((k, 5), dtype=torch.float32).pin_memory(),
                                                 torch.empty((k + 1,), dtype=torch.int64).pin_memory())
            f32, i64 = buf
            f32[:, :4].copy_(boxes, non_blocking=True)
            f32[:, 4].copy_(scores, non_blocking=True)
            i64[:k].copy_(labels, non_blocking=True)
            i64[k:].copy_(count.reshape(1), non_blocking=True)
            pins.append((f32, i64, k))
        handle["pinned"] = pins
        return handle

    @staticmethod
    def finish(handle):
        """The detections of `launch_heads` as CPU tensors (the stream that ran it must have drained: engine.evaluate's next wait)."""
        if "done" in handle:
            return handle["done"]
        out = []
        for f32, i64, k in handle["pinned"]:
            n = int(i64[k])
            out.append({"boxes": f32[:n, :4].clone(), "labels": i64[:n].clone(), "scores": f32[:n, 4].clone()})
        return out

    def forward(self, images, targets=None, thetas=None, lambda1s=None, lambda2s=None, killWarp=False, newMeans=None,
                newSTDs=None):
        if self.training and targets is None:
            raise ValueError("In training mode, targets should be passed")
        if self.training:
            for target in targets:
                boxes = target["boxes"]
                if not isinstance(boxes, torch.Tensor):
                    raise ValueError("Expected target boxes to be of type Tensor, got {:}.".format(type(boxes)))
                if boxes.dim() != 2 or boxes.shape[-1] != 4:
                    # the reference's message is two literals joined without a space (generalized_rcnn.py:99-101): kept verbatim
                    raise ValueError("Expected target boxes to be a tensor" "of shape [N, 4], got {:}.".format(boxes.shape))
        original_sizes = [(int(img.shape[-2]), int(img.shape[-1])) for img in images]
        images, targets = self.transform(images, targets, newMeans, newSTDs)
        if (self.graph_inference and not self.training and targets is None and not torch.is_grad_enabled() and images.tensors.is_cuda
                and not (self.warp_internally and not killWarp)):
            return self._forward_graphed(images, original_sizes)
        # degenerate-box check (reference generalized_rcnn.py:119-129).  The flag is computed on the device right here
        # and copied to pinned host memory behind the transform; it is READ only once the whole forward pass has been
        # enqueued, after waiting for that copy alone -- the GPU reaches it before it starts the backbone, so the host
        # neither stalls behind the forward pass nor leaves the queue empty when it goes on to issue the backward pass
        # (a plain `bool(flag)` at either place drains the queue once per step).  Same ValueError, raised before
        # anything is returned.
        degenerate = flag_ready = None
        if targets is not None:
            flags = [(t["boxes"][:, 2:] <= t["boxes"][:, :2]).any() for t in targets]
            if flags:
                degenerate = torch.stack(flags).any()
                if degenerate.is_cuda:
                    if getattr(self, "_flag_host", None) is None:
                        self._flag_host = torch.zeros((), dtype=torch.bool).pin_memory()
                    self._flag_host.copy_(degenerate, non_blocking=True)
                    flag_ready = torch.cuda.Event()
                    flag_ready.record()
                    degenerate = self._flag_host
        if self.warp_internally and not killWarp:
            # squint: stretch along the blur axes, run the trunk, un-stretch every pyramid level
            features = self.backbone(self.warper(images.tensors, thetas, lambda1s, lambda2s))
            if isinstance(features, torch.Tensor):
                features = OrderedDict([("0", features)])
            for key, feature in features.items():
                features[key] = self.warper(feature, thetas, 1 / lambda1s, 1 / lambda2s)
        else:
            features = self.backbone(images.tensors)
        if isinstance(features, torch.Tensor):
            features = OrderedDict([("0", features)])
        proposals, proposal_losses = self.rpn(images, features, targets)
        detections, detector_losses = self.roi_heads(features, proposals, images.image_sizes, targets)
        detections = self.transform.postprocess(detections, images.image_sizes, original_sizes)
        if flag_ready is not None:
            flag_ready.synchronize()
        if degenerate is not None and bool(degenerate):
            for idx, target in enumerate(targets):
                boxes = target["boxes"]
                bad = boxes[:, 2:] <= boxes[:, :2]
                if bad.any():
                    bb = boxes[bad.any(dim=1).nonzero().view(-1)[0]].tolist()
                    raise ValueError("All bounding boxes should have positive height and width."
                                     " Found invaid box {} for target at index {}.".format(bb, idx))
        if self.training:
            losses = {}
            losses.update(detector_losses)
            losses.update(proposal_losses)
            return losses
        return detections
