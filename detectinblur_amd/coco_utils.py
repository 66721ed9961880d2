"""Data feed with the reference's item layout `(image, target, blur_dict)` (reference coco_utils.py).

Two datasets, one layout:
  * `CocoDetection` -- a real COCO tree (`{root}/train2017`, `{root}/annotations/instances_train2017.json`, ...)
    read with `json` + PIL only: neither torchvision.datasets.CocoDetection nor pycocotools is needed for the
    box path (reference coco_utils.py:231-271).  `ConvertCocoPolysToMask` turns the raw annotations into the
    target dict of reference coco_utils.py:51-104 (`boxes` xyxy float32 clipped to the image, `labels` int64,
    `masks` uint8 [N, h, w] rasterised from the polygons by native host code -- pycocotools' rleFrPoly restated, bit-identical
    to the reference's own maskApi.c --, `keypoints` when present, `image_id`, `area`, `iscrowd`).  Training drops images
    without a usable annotation (:106-147).
  * `SyntheticCocoDetection` -- COCO-shaped random images and boxes, the offline stand-in (`--synthetic`).
`get_coco_api_from_dataset` (reference :218-226) hands `engine.evaluate` the ground truth as a `CocoGT`
(the few fields of pycocotools' COCO object the evaluation reads: `dataset`, `imgToAnns`, `imgs`).
"""
import json
import os
from collections import defaultdict

import numpy as np
import torch
import torch.utils.data

from . import transforms as T


class CocoGT(object):
    """The slice of pycocotools.coco.COCO the evaluation path touches (reference coco_eval.py:24-29,
    engine.py:325-342): `dataset`, `imgs`, `imgToAnns` (lists of the SAME annotation dicts, so in-place
    edits of `bbox` are seen by the evaluator), `getAnnIds` / `loadAnns` / `getCatIds`."""

    def __init__(self, dataset=None):
        self.dataset = dataset if dataset is not None else {"images": [], "annotations": [], "categories": []}
        self.createIndex()

    def createIndex(self):
        self.anns, self.imgs, self.cats = {}, {}, {}
        self.imgToAnns = defaultdict(list)
        for ann in self.dataset.get("annotations", []):
            self.imgToAnns[ann["image_id"]].append(ann)
            self.anns[ann["id"]] = ann
        for img in self.dataset.get("images", []):
            self.imgs[img["id"]] = img
        for cat in self.dataset.get("categories", []):
            self.cats[cat["id"]] = cat

    def getAnnIds(self, imgIds, iscrowd=None):
        ids = imgIds if isinstance(imgIds, (list, tuple)) else [imgIds]
        return [a["id"] for i in ids for a in self.imgToAnns.get(i, []) if iscrowd is None or a["iscrowd"] == iscrowd]

    def loadAnns(self, ids):
        return [self.anns[i] for i in ids]

    def getCatIds(self):
        return sorted(self.cats)

    def getImgIds(self):
        return sorted(self.imgs)


def _one_object_mask(segmentation, height, width):
    """`coco_mask.decode(coco_mask.frPyObjects(segmentation, height, width))` reduced with `any` over the object's parts (reference
    coco_utils.py:37-43), as one uint8 [height, width] array: the segmentation forms pycocotools' frPyObjects accepts
    (cocoapi/PythonAPI/pycocotools/_mask.pyx:288-308) -- a list of polygons (more than four numbers each), a list of [x, y, w, h]
    boxes (four numbers each: rasterised as their four corners), a list of uncompressed RLE dicts, or one polygon / box / RLE dict
    -- rasterised by native host code (csrc/host/dib_host.c: dib_mask_or_polygon, the restatement of maskApi.c's rleFrPoly)."""
    import ctypes
    from . import _hostlib
    lib = _hostlib.lib()
    mask = np.zeros((height, width), dtype=np.uint8)
    mp = mask.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte))

    def polygon(p):
        xy = np.array(p, dtype=np.float64)
        if lib.dib_mask_or_polygon(_hostlib.dptr(xy), int(len(p) / 2), height, width, mp) != 0:
            raise ValueError("polygon of %d numbers on a %d x %d image cannot be rasterised" % (len(p), height, width))

    def box(b):
        xs, ys, xe, ye = float(b[0]), float(b[1]), float(b[0]) + float(b[2]), float(b[1]) + float(b[3])
        polygon([xs, ys, xs, ye, xe, ye, xe, ys])                           # maskApi.c:148-156 (rleFrBbox)

    def runs(r):
        if tuple(r["size"]) != (height, width):
            raise ValueError("RLE of size %s on a %d x %d image" % (r["size"], height, width))
        cnts = np.array(r["counts"], dtype=np.uint32)
        if lib.dib_mask_or_runs(cnts.ctypes.data_as(ctypes.POINTER(ctypes.c_uint)), len(cnts), height, width, mp) != 0:
            raise ValueError("bad RLE")

    def is_rle(o):
        return type(o) == dict and "counts" in o and "size" in o

    o = segmentation
    if type(o) == list and len(o) == 0:
        pass            # no outline (box-only annotations): an empty mask, where pycocotools raises IndexError
    elif type(o) == list and len(o[0]) == 4:
        for b in o:
            box(b)
    elif type(o) == list and len(o[0]) > 4:
        for p in o:
            polygon(p)
    elif type(o) == list and is_rle(o[0]):
        for r in o:
            runs(r)
    elif type(o) == list and len(o) == 4:
        box(o)
    elif type(o) == list and len(o) > 4:
        polygon(o)
    elif is_rle(o):
        runs(o)
    else:
        raise Exception("input type is not supported.")
    return mask


def convert_coco_poly_to_mask(segmentations, height, width):
    """reference coco_utils.py:34-49: one uint8 mask per object, [N, height, width] ([0, height, width] for no object)."""
    masks = [torch.from_numpy(_one_object_mask(seg, height, width)) for seg in segmentations]
    if masks:
        return torch.stack(masks, dim=0)
    return torch.zeros((0, height, width), dtype=torch.uint8)


class ConvertCocoPolysToMask(object):
    """reference coco_utils.py:51-104: crowd annotations are dropped from the target, xywh -> xyxy, clamped to the image, boxes
    without positive extent dropped together with their labels, masks and keypoints; `area` / `iscrowd` keep one entry per
    non-crowd annotation (NOT filtered by `keep`, as in the reference).  `masks` uint8 [N, h, w] from the objects'
    `segmentation` (polygons / RLE), `keypoints` float32 [N, K, 3] when the annotations carry them.  `with_masks=False` skips the
    masks (Faster R-CNN never reads them; the reference always builds them)."""

    def __init__(self, with_masks=True):
        self.with_masks = with_masks

    def __call__(self, image, target, blur_dict=None):
        blur_dict = {} if blur_dict is None else blur_dict
        w, h = image.size
        image_id = torch.tensor([target["image_id"]])
        anno = [obj for obj in target["annotations"] if obj["iscrowd"] == 0]
        boxes = torch.as_tensor([obj["bbox"] for obj in anno], dtype=torch.float32).reshape(-1, 4)
        boxes[:, 2:] += boxes[:, :2]
        boxes[:, 0::2].clamp_(min=0, max=w)
        boxes[:, 1::2].clamp_(min=0, max=h)
        classes = torch.tensor([obj["category_id"] for obj in anno], dtype=torch.int64)
        masks = convert_coco_poly_to_mask([obj.get("segmentation", []) for obj in anno], h, w) if self.with_masks else None
        keypoints = None
        if anno and "keypoints" in anno[0]:
            keypoints = torch.as_tensor([obj["keypoints"] for obj in anno], dtype=torch.float32)
            if keypoints.shape[0]:
                keypoints = keypoints.view(keypoints.shape[0], -1, 3)
        keep = (boxes[:, 3] > boxes[:, 1]) & (boxes[:, 2] > boxes[:, 0])
        out = {"boxes": boxes[keep], "labels": classes[keep]}
        if masks is not None:
            out["masks"] = masks[keep]
        out["image_id"] = image_id
        if keypoints is not None:
            out["keypoints"] = keypoints[keep]
        out["area"] = torch.tensor([obj["area"] for obj in anno])
        out["iscrowd"] = torch.tensor([obj["iscrowd"] for obj in anno])
        return image, out, blur_dict


class CocoDetection(torch.utils.data.Dataset):
    """torchvision.datasets.CocoDetection + the reference's subclass (coco_utils.py:231-240) without either
    dependency: item = `(PIL RGB image, {"image_id", "annotations"}, {})` run through `transforms`."""

    def __init__(self, img_folder, ann_file, transforms=None):
        self.root = img_folder
        with open(ann_file) as f:
            self.coco = CocoGT(json.load(f))
        self.ids = list(sorted(self.coco.imgs.keys()))
        self._transforms = transforms
        self._epoch_number = 0

    def __len__(self):
        return len(self.ids)

    def _load_image(self, image_id):
        from PIL import Image
        return Image.open(os.path.join(self.root, self.coco.imgs[image_id]["file_name"])).convert("RGB")

    def __getitem__(self, idx):
        image_id = self.ids[idx]
        img = self._load_image(image_id)
        target = dict(image_id=image_id, annotations=self.coco.loadAnns(self.coco.getAnnIds(image_id)))
        blur_dict = {}
        if self._transforms is not None:
            img, target, blur_dict = self._transforms(img, target)
        return img, target, blur_dict


def _has_valid_annotation(anno):
    """reference coco_utils.py:106-132."""
    if len(anno) == 0:
        return False
    if all(any(o <= 1 for o in obj["bbox"][2:]) for obj in anno):          # every box (close to) empty
        return False
    if "keypoints" not in anno[0]:
        return True
    return sum(sum(1 for v in ann["keypoints"][2::3] if v > 0) for ann in anno) >= 10


def _coco_remove_images_without_annotations(dataset, cat_list=None):
    """reference coco_utils.py:106-147: a Subset of the images that carry at least one usable annotation."""
    ids = []
    for ds_idx, img_id in enumerate(dataset.ids):
        anno = dataset.coco.loadAnns(dataset.coco.getAnnIds(img_id, iscrowd=None))
        if cat_list:
            anno = [obj for obj in anno if obj["category_id"] in cat_list]
        if _has_valid_annotation(anno):
            ids.append(ds_idx)
    return torch.utils.data.Subset(dataset, ids)


class SyntheticCocoDetection(torch.utils.data.Dataset):
    """Seeded random images + boxes from generator seed `seed + i`: the boxes first, then image i = `torch.rand(3, H, W)`;
    `boxes_per_image` boxes with x1, y1 uniform and w, h uniform in [32, 400] clipped to the image,
    labels uniform in 1..num_classes-1.  `sizes`: a list of (H, W) that the items cycle through (image i has
    sizes[i % len(sizes)]: the ragged sizes real COCO images come in) instead of the one `size`."""

    def __init__(self, num_images=64, size=(800, 1333), boxes_per_image=8, num_classes=91, transforms=None, seed=1337,
                 as_tensor=True, sizes=None):
        self.num_images, self.size, self.boxes_per_image, self.num_classes = num_images, size, boxes_per_image, num_classes
        self.sizes = [tuple(x) for x in sizes] if sizes else None
        self._transforms, self.seed, self.as_tensor = transforms, seed, as_tensor
        self.epoch_number = None
        self._epoch_number = 0

    def __len__(self):
        return self.num_images

    def size_of(self, idx):
        return self.sizes[idx % len(self.sizes)] if self.sizes else self.size

    def _target(self, idx, g):
        H, W = self.size_of(idx)
        n = self.boxes_per_image
        x1 = torch.rand(n, generator=g) * (W - 34)
        y1 = torch.rand(n, generator=g) * (H - 34)
        w = 32 + torch.rand(n, generator=g) * (400 - 32)
        h = 32 + torch.rand(n, generator=g) * (400 - 32)
        boxes = torch.stack([x1, y1, torch.minimum(x1 + w, torch.tensor(float(W))), torch.minimum(y1 + h, torch.tensor(float(H)))], 1)
        labels = torch.randint(1, self.num_classes, (n,), generator=g)
        return {"boxes": boxes, "labels": labels, "image_id": torch.tensor([idx]),
                "area": (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]), "iscrowd": torch.zeros(n, dtype=torch.int64)}

    def __getitem__(self, idx):
        H, W = self.size_of(idx)
        g = torch.Generator().manual_seed(self.seed + idx)
        target = self._target(idx, g)             # drawn first: `annotations` below then needs none of the image's draws
        img = torch.rand(3, H, W, generator=g)
        if not self.as_tensor:   # PIL image, as CocoDetection hands to the transforms (needed by --cpu_blur)
            from PIL import Image
            img = Image.fromarray((img.permute(1, 2, 0).numpy() * 255).astype(np.uint8))
        blur_dict = {}
        if self._transforms is not None:
            img, target, blur_dict = self._transforms(img, target, blur_dict)
        return img, target, blur_dict

    def annotations(self, idx):
        """The target of item idx without rendering the image or running the transforms (same generator stream)."""
        return self._target(idx, torch.Generator().manual_seed(self.seed + idx))


def convert_to_coco_api(ds):
    """reference coco_utils.py:150-215 for datasets that are not a CocoDetection: one pass over the items'
    targets -> COCO-style ground truth (annotation ids from 1, bbox as xywh)."""
    dataset = {"images": [], "categories": [], "annotations": []}
    categories, ann_id = set(), 1
    for img_idx in range(len(ds)):
        if hasattr(ds, "annotations"):
            targets, (height, width) = ds.annotations(img_idx), ds.size_of(img_idx)
        else:
            img, targets, _ = ds[img_idx]
            height, width = (img.height, img.width) if hasattr(img, "height") else (img.shape[-2], img.shape[-1])
        image_id = int(targets["image_id"].item())
        dataset["images"].append({"id": image_id, "height": height, "width": width})
        bboxes = targets["boxes"].clone()
        bboxes[:, 2:] -= bboxes[:, :2]
        bboxes = bboxes.tolist()
        labels, areas, iscrowd = targets["labels"].tolist(), targets["area"].tolist(), targets["iscrowd"].tolist()
        for i in range(len(bboxes)):
            dataset["annotations"].append({"image_id": image_id, "bbox": bboxes[i], "category_id": labels[i], "area": areas[i],
                                           "iscrowd": iscrowd[i], "id": ann_id})
            categories.add(labels[i])
            ann_id += 1
    dataset["categories"] = [{"id": i} for i in sorted(categories)]
    return CocoGT(dataset)


def get_coco_api_from_dataset(dataset):
    """reference coco_utils.py:218-226."""
    for _ in range(10):
        if isinstance(dataset, CocoDetection):
            break
        if isinstance(dataset, torch.utils.data.Subset):
            dataset = dataset.dataset
    if isinstance(dataset, CocoDetection):
        return dataset.coco
    return convert_to_coco_api(dataset)


def get_coco(root, image_set, transforms, mode="instances", synthetic=None, with_masks=True):
    """reference coco_utils.py:243-271.  Returns `(dataset, num_classes)` (the reference's train.get_dataset
    adds the 91; folded in here).  With `synthetic` (a dict of SyntheticCocoDetection kwargs) `root` is ignored.
    `with_masks=False` (what the detection drivers pass unless `--with_masks` is given): the targets carry no `masks` -- the
    reference rasterises every object's full-resolution mask in the loader and resizes it with the image every step for a
    Faster R-CNN that never reads it; the default here keeps the reference's target dict."""
    if synthetic is not None:
        return SyntheticCocoDetection(transforms=transforms, **synthetic), 91
    if root is None:
        raise RuntimeError("no --data_path given: point it at a COCO tree (train2017/, val2017/, annotations/) "
                           "or pass --synthetic for COCO-shaped synthetic data")
    anno_file_template = "{}_{}2017.json"
    PATHS = {"train": ("train2017", os.path.join("annotations", anno_file_template.format(mode, "train"))),
             "val": ("val2017", os.path.join("annotations", anno_file_template.format(mode, "val")))}
    t = [ConvertCocoPolysToMask(with_masks=with_masks)]
    if transforms is not None:
        t.append(transforms)
    img_folder, ann_file = PATHS[image_set]
    img_folder, ann_file = os.path.join(root, img_folder), os.path.join(root, ann_file)
    if not os.path.isfile(ann_file):
        raise FileNotFoundError("COCO annotation file %s not found (pass --synthetic to run without a dataset)" % ann_file)
    dataset = CocoDetection(img_folder, ann_file, transforms=T.Compose(t))
    if image_set == "train":
        dataset = _coco_remove_images_without_annotations(dataset)
    return dataset, 91


def get_coco_kp(root, image_set, transforms):
    return get_coco(root, image_set, transforms, mode="person_keypoints")

