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
