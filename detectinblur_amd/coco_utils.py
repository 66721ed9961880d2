"""Data feed with the reference's item layout `(image, target, blur_dict)` (reference
coco_utils.py:231-240).  COCO itself (torchvision.datasets.CocoDetection + pycocotools) is not
available offline, so the COCO-shaped synthetic dataset below stands in for it: same tensors, same
target keys (`boxes` xyxy float32, `labels` int64, `image_id`, `area`, `iscrowd`; coco_utils.py:90-102).
"""
import numpy as np
import torch
import torch.utils.data


class SyntheticCocoDetection(torch.utils.data.Dataset):
    """Seeded random images + boxes: image i is `torch.rand(3, H, W)` from generator seed `seed + i`;
    `boxes_per_image` boxes with x1, y1 uniform and w, h uniform in [32, 400] clipped to the image,
    labels uniform in 1..num_classes-1."""

    def __init__(self, num_images=64, size=(800, 1333), boxes_per_image=8, num_classes=91, transforms=None, seed=1337,
                 as_tensor=True):
        self.num_images, self.size, self.boxes_per_image, self.num_classes = num_images, size, boxes_per_image, num_classes
        self._transforms, self.seed, self.as_tensor = transforms, seed, as_tensor
        self.epoch_number = None

    def __len__(self):
        return self.num_images

    def __getitem__(self, idx):
        H, W = self.size
        g = torch.Generator().manual_seed(self.seed + idx)
        img = torch.rand(3, H, W, generator=g)
        n = self.boxes_per_image
        x1 = torch.rand(n, generator=g) * (W - 34)
        y1 = torch.rand(n, generator=g) * (H - 34)
        w = 32 + torch.rand(n, generator=g) * (400 - 32)
        h = 32 + torch.rand(n, generator=g) * (400 - 32)
        boxes = torch.stack([x1, y1, torch.minimum(x1 + w, torch.tensor(float(W))), torch.minimum(y1 + h, torch.tensor(float(H)))], 1)
        labels = torch.randint(1, self.num_classes, (n,), generator=g)
        target = {"boxes": boxes, "labels": labels, "image_id": torch.tensor([idx]),
                  "area": (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]), "iscrowd": torch.zeros(n, dtype=torch.int64)}
        if not self.as_tensor:   # PIL image, as CocoDetection hands to the transforms (needed by --cpu_blur)
            from PIL import Image
            img = Image.fromarray((img.permute(1, 2, 0).numpy() * 255).astype(np.uint8))
        blur_dict = {}
        if self._transforms is not None:
            img, target, blur_dict = self._transforms(img, target, blur_dict)
        return img, target, blur_dict


def get_coco(root, image_set, transforms, mode="instances", synthetic=None):
    """(dataset, num_classes).  `root` is ignored when `synthetic` (a dict of SyntheticCocoDetection
    kwargs) is given; a real COCO tree needs torchvision + pycocotools, which this image lacks."""
    if synthetic is None:
        raise RuntimeError("COCO needs torchvision.datasets.CocoDetection and pycocotools (not installed); "
                           "pass --synthetic to train/evaluate on COCO-shaped synthetic data")
    return SyntheticCocoDetection(transforms=transforms, **synthetic), 91
