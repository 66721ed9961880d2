"""Input transform of the detector -- drop-in for the reference's models/net_transforms.py:
per-image normalisation with statistics supplied per call (`newMeans` / `newSTDs`, reference :112-118),
resize to min side 800 / max side 1333 (:36-46), zero-padded batching to a multiple of 32 (:238-247),
the `crop_images` mode the blur-estimator input uses (:226-236), and box rescaling (:302-316).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn


class ImageList(object):
    """Batched images (one padded tensor) + the un-padded size of each."""

    def __init__(self, tensors, image_sizes):
        self.tensors = tensors
        self.image_sizes = image_sizes

    def to(self, device):
        return ImageList(self.tensors.to(device), self.image_sizes)


def resize_boxes(boxes, original_size, new_size):
    # float32 ratios computed on the host: a 0-dim device tensor per ratio would cost one blocking
    # H2D copy each (4 per image); a Python float holding the float32 quotient multiplies identically
    rh, rw = (float(np.float32(n) / np.float32(o)) for n, o in zip(new_size, original_size))
    x1, y1, x2, y2 = boxes.unbind(1)
    return torch.stack((x1 * rw, y1 * rh, x2 * rw, y2 * rh), dim=1)


class GeneralizedRCNNTransform(nn.Module):
    def __init__(self, min_size, max_size, image_mean, image_std, crop_images=False, training=True, normalize_images=True):
        super().__init__()
        self.min_size = min_size if isinstance(min_size, (list, tuple)) else (min_size,)
        self.max_size = max_size
        self.image_mean, self.image_std = image_mean, image_std
        self.crop_images = crop_images
        self.training = training
        self.normalize_images = normalize_images
        # Fused epilogue (csrc/dib_epilogue.hip): float conversion + normalisation + zero-padded batch in one launch for
        # CUDA batches that need no resize; bit-identical to the module-by-module path below, which stays the checker
        # (tests/test_epilogue_gpu.py) and serves every other case.
        self.fused = True

    _stat_cache = {}

    @classmethod
    def _stat(cls, values, dtype, device):
        """Per-channel statistics as a device tensor.  The rows repeat from step to step (ImageNet
        stats, or one of the 15 custom-norm rows), so each distinct row is uploaded once: a fresh
        `as_tensor(..., device=)` per image is a blocking H2D copy that stalls the host behind all
        queued GPU work, 16 times per batch."""
        key = (tuple(float(v) for v in values), dtype, str(device))
        t = cls._stat_cache.get(key)
        if t is None:
            if len(cls._stat_cache) > 256:
                cls._stat_cache.clear()
            t = cls._stat_cache[key] = torch.as_tensor(key[0], dtype=dtype, device=device)
        return t

    def normalize(self, image, mean, std):
        mean = self._stat(mean, image.dtype, image.device)
        std = self._stat(std, image.dtype, image.device)
        return (image - mean[:, None, None]) / std[:, None, None]

    def resize(self, image, target):
        h, w = image.shape[-2:]
        scale = self._scale(h, w, self._target_size())
        if scale != 1.0:
            image = F.interpolate(image[None], scale_factor=scale, mode="bilinear", recompute_scale_factor=True,
                                  align_corners=False)[0]
        if target is None:
            return image, target
        self._resize_target(target, (h, w), tuple(image.shape[-2:]), scale)
        return image, target

    @staticmethod
    def _resize_target(target, src, dst, scale):
        """What the resize does to a target (reference net_transforms.py:52-55, :165-172, :283-299): masks by nearest neighbour
        with the image's scale factor, boxes and keypoints by the ratio of the sizes."""
        if "masks" in target:
            target["masks"] = F.interpolate(target["masks"][:, None].float(), scale_factor=scale)[:, 0].byte()
        target["boxes"] = resize_boxes(target["boxes"], src, dst)
        if "keypoints" in target:
            rh, rw = (float(np.float32(n) / np.float32(o)) for n, o in zip(dst, src))
            kp = target["keypoints"].clone()
            kp[..., 0] *= rw
            kp[..., 1] *= rh
            target["keypoints"] = kp

    def batch_images(self, images, size_divisible=32):
        shapes = [list(img.shape) for img in images]
        stride = float(size_divisible)
        if self.crop_images:
            size = [min(s[i] for s in shapes) for i in range(3)]
            size[1] = int(math.floor(size[1] / stride) * stride)
            size[2] = int(math.floor(size[2] / stride) * stride)
            out = self._new_batch(images[0], [len(images)] + size)
            for img, dst in zip(images, out):
                dst.copy_(img[:size[0], :size[1], :size[2]])
            return out
        size = [max(s[i] for s in shapes) for i in range(3)]
        size[1] = int(math.ceil(size[1] / stride) * stride)
        size[2] = int(math.ceil(size[2] / stride) * stride)
        out = self._new_batch(images[0], [len(images)] + size)
        for img, dst in zip(images, out):
            dst[:img.shape[0], :img.shape[1], :img.shape[2]].copy_(img)
        return out

    def _new_batch(self, like, shape):
        fmt = torch.channels_last if getattr(self, "channels_last", False) else torch.contiguous_format
        return torch.empty(shape, dtype=like.dtype, device=like.device, memory_format=fmt).zero_()

    def _target_size(self):
        """The min-side target of the next image: one draw on torch's CPU generator when training (reference
        net_transforms.py:141-149, :155-156), the largest scale otherwise."""
        if self.training:
            return float(self.min_size[int(torch.empty(1).uniform_(0.0, float(len(self.min_size))).item())])
        return float(self.min_size[-1])

    def _scale(self, h, w, size):
        lo, hi = float(min(h, w)), float(max(h, w))
        scale = size / lo
        if hi * scale > self.max_size:
            scale = self.max_size / hi
        return scale

    def _qualifies_for_fused(self, images):
        if not (self.fused and self.normalize_images and not self.crop_images and images):
            return False
        first = images[0]
        return (first.is_cuda and first.dtype in (torch.float16, torch.float32)
                and not any(i.dim() != 3 or i.shape[0] != 3 or i.dtype != first.dtype or not i.is_cuda for i in images))

    @staticmethod
    def _materialize(images, pending):
        """The blur that `pending_blur` postponed, as its own launch: the images, blurred."""
        from .. import blur_ops
        index, tables, acc_mode = pending
        return blur_ops.sparse_blur(list(images), list(index), tables, acc_mode)

    def _forward_fused(self, images, targets, newMeans, newSTDs, pending=None):
        """None when the batch does not qualify (not on the GPU, mixed dtypes, crop mode, images that are not 3 x H x W);
        generator draws are consumed exactly as the unfused path would.  Images that need the resize of :151-175 (every
        native-size COCO image) are resized by the same launch (dib_normalize_resize_pad).
        `pending` = (table_index, tables, acc_mode): the images are still UNBLURRED (engine.py, opt-in `FUSE_BLUR_EPILOGUE`); when
        no image needs a resize the blur and this epilogue are ONE launch (blur_ops.sparse_blur_normalized: the blurred fp16
        batch never exists), otherwise the blur is launched here first and everything goes on as usual."""
        if not (self.fused and self.normalize_images and not self.crop_images and images):
            return None
        first = images[0]
        if not (first.is_cuda and first.dtype in (torch.float16, torch.float32)):
            return None
        if any(i.dim() != 3 or i.shape[0] != 3 or i.dtype != first.dtype or not i.is_cuda for i in images):
            return None
        sizes = [self._target_size() for _ in images]           # one generator draw per image when training, in image order
        from .. import blur_ops
        n = len(images)
        means = newMeans if newMeans is not None else np.tile(np.asarray(self.image_mean, dtype=np.float64), (n, 1))
        stds = newSTDs if newSTDs is not None else np.tile(np.asarray(self.image_std, dtype=np.float64), (n, 1))
        hw = [(int(i.shape[-2]), int(i.shape[-1])) for i in images]
        out_hw, scales = [], []
        for (h, w), s in zip(hw, sizes):
            scale = self._scale(h, w, s)
            scales.append(scale)
            # F.interpolate(..., scale_factor=scale, recompute_scale_factor=True): output size int(size * scale), per dimension
            out_hw.append((h, w) if scale == 1.0 else (int(h * scale), int(w * scale)))
        if any(oh <= 0 or ow <= 0 for oh, ow in out_hw):
            return None                                         # degenerate sliver: let interpolate raise what it raises
        Hp = int(math.ceil(max(h for h, _ in out_hw) / 32.0) * 32)
        Wp = int(math.ceil(max(w for _, w in out_hw) / 32.0) * 32)
        batch = None
        if pending is not None:
            if out_hw == hw and first.dtype == torch.float16:
                batch = blur_ops.sparse_blur_normalized(images, pending[0], pending[1], means, stds, Hp, Wp,
                                                        getattr(self, "channels_last", False), pending[2], order=pending[3] if len(pending) > 3 else None)
                self.last_epilogue = "fused into the blur" if batch is not None else "own launch"
            if batch is None:
                images = self._materialize(images, pending[:3])
        if batch is None:
            batch = blur_ops.normalize_pad(images, means, stds, Hp, Wp, getattr(self, "channels_last", False), out_sizes=out_hw)
        if targets is not None:
            for t, src, dst, scale in zip(targets, hw, out_hw, scales):
                self._resize_target(t, src, dst, scale)
        return ImageList(batch, out_hw), targets

    def forward(self, images, targets=None, newMeans=None, newSTDs=None):
        images = list(images)
        if targets is not None:
            targets = [dict(t) for t in targets]        # shallow copies: the caller's dicts stay untouched
        for image in images:
            if image.dim() != 3:
                raise ValueError("images is expected to be a list of 3d tensors of shape [C, H, W], got {}".format(image.shape))
        pending = self.__dict__.pop("pending_blur", None)      # set by engine.py for ONE call: the images are still unblurred
        if pending is not None and not self._qualifies_for_fused(images):
            images = self._materialize(images, pending[:3])
            pending = None
        fused = self._forward_fused(images, targets, newMeans, newSTDs, pending)
        if fused is not None:
            return fused
        # Half images only arrive from engine.py when the fused path was expected to take them: convert as the
        # reference's engine would have (engine.py:107-110)
        images = [i.float() if i.dtype == torch.float16 else i for i in images]
        for i, image in enumerate(images):
            if image.dim() != 3:
                raise ValueError("images is expected to be a list of 3d tensors of shape [C, H, W], got {}".format(image.shape))
            if self.normalize_images:
                if newMeans is not None:
                    image = self.normalize(image, newMeans[i, :], newSTDs[i, :])
                else:
                    image = self.normalize(image, self.image_mean, self.image_std)
            image, t = self.resize(image, targets[i] if targets is not None else None)
            images[i] = image
            if targets is not None and t is not None:
                targets[i] = t
        sizes = [(int(img.shape[-2]), int(img.shape[-1])) for img in images]
        return ImageList(self.batch_images(images), sizes), targets

    def postprocess(self, result, image_shapes, original_image_sizes):
        if self.training:
            return result
        for i, (pred, im_s, o_im_s) in enumerate(zip(result, image_shapes, original_image_sizes)):
            result[i]["boxes"] = resize_boxes(pred["boxes"], im_s, o_im_s)
        return result
