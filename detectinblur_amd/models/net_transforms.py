"""Input transform of the detector -- drop-in for the reference's models/net_transforms.py:
per-image normalisation with statistics supplied per call (`newMeans` / `newSTDs`, reference :112-118),
resize to min side 800 / max side 1333 (:36-46), zero-padded batching to a multiple of 32 (:238-247),
the `crop_images` mode the blur-estimator input uses (:226-236), and box rescaling (:302-316).
"""
import math

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
    rh, rw = (torch.tensor(n, dtype=torch.float32, device=boxes.device) / torch.tensor(o, dtype=torch.float32, device=boxes.device)
              for n, o in zip(new_size, original_size))
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

    def normalize(self, image, mean, std):
        mean = torch.as_tensor(mean, dtype=image.dtype, device=image.device)
        std = torch.as_tensor(std, dtype=image.dtype, device=image.device)
        return (image - mean[:, None, None]) / std[:, None, None]

    def resize(self, image, target):
        h, w = image.shape[-2:]
        if self.training:
            size = float(self.min_size[int(torch.empty(1).uniform_(0.0, float(len(self.min_size))).item())])
        else:
            size = float(self.min_size[-1])
        lo, hi = float(min(h, w)), float(max(h, w))
        scale = size / lo
        if hi * scale > self.max_size:
            scale = self.max_size / hi
        if scale != 1.0:
            image = F.interpolate(image[None], scale_factor=scale, mode="bilinear", recompute_scale_factor=True,
                                  align_corners=False)[0]
        if target is None:
            return image, target
        target["boxes"] = resize_boxes(target["boxes"], (h, w), image.shape[-2:])
        return image, target

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

    def forward(self, images, targets=None, newMeans=None, newSTDs=None):
        images = list(images)
        if targets is not None:
            targets = [dict(t) for t in targets]        # shallow copies: the caller's dicts stay untouched
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
