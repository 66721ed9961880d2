"""Drop-in for the hot-path parts of the reference's utils.py.

  * expand_targets / fix_bounding_box_squeeze   reference utils.py:360-434  (HIP, one launch/image)
  * get_norm_params                              reference utils.py:219-273  (host tables)
  * collate_fn, distributed helpers, meters      reference utils.py:474-785  (see utils_dist.py)
"""
import numpy as np
import torch

from . import blur_ops

# ---------------------------------------------------------------------------------------------
# boxes
# ---------------------------------------------------------------------------------------------


def expand_targets(targets_GPU, blur_dicts, psfs_GPU, images_GPU):
    """Grows every ground-truth box by the PSF's non-zero extent, then clamps (in place on
    target["boxes"]; returns the same list).  Reference utils.py:360-392."""
    idx = [i for i, bd in enumerate(blur_dicts) if bd["blurring"]]
    if not idx:
        return targets_GPU
    for i in idx:
        # "This function is not flexible on purpose" (utils.py:366-370)
        if psfs_GPU[i].shape[0] != 128:
            raise Exception("Trying to expand with filters that are not 128 wide!")
    psfs = []
    for i in idx:
        p, img = psfs_GPU[i], images_GPU[i]
        psfs.append(p if p.dtype in (torch.float16, torch.float32) else p.float())
    if len({p.dtype for p in psfs}) != 1:
        psfs = [p.float() for p in psfs]
    tables = blur_ops.compact_psfs_cached(psfs, normalize=True)
    for k, i in enumerate(idx):
        boxes = targets_GPU[i]["boxes"]
        shape = images_GPU[i].shape
        work = boxes if (boxes.dtype == torch.float32 and boxes.is_contiguous()) else boxes.float().contiguous()
        blur_ops.expand_boxes(work, tables, k, int(shape[1]), int(shape[2]))
        if work is not boxes:
            boxes.copy_(work)
    return targets_GPU


def fix_bounding_box_squeeze(target, image_shape):
    """Clamp to the image, open up degenerate boxes by one pixel each way, clamp again.
    Reference utils.py:395-434.  image_shape is (C, H, W)."""
    boxes = target["boxes"]
    work = boxes if (boxes.dtype == torch.float32 and boxes.is_contiguous()) else boxes.float().contiguous()
    blur_ops.clamp_boxes(work, int(image_shape[1]), int(image_shape[2]))
    if work is not boxes:
        boxes.copy_(work)
    return target


def convert_to_xywh(boxes):
    xmin, ymin, xmax, ymax = boxes.unbind(1)
    return torch.stack((xmin, ymin, xmax - xmin, ymax - ymin), dim=1)


# ---------------------------------------------------------------------------------------------
# per-image normalisation statistics
# ---------------------------------------------------------------------------------------------

_CANONICAL_MEAN = (0.485, 0.456, 0.406)
_CANONICAL_STD = (0.229, 0.224, 0.225)
# channel std of COCO train images after blurring: rows = (clean, E0..E4), per blur type P1..P3
# (data published in the reference, utils.py:228-230)
_BLUR_STD = np.asarray([
    [[0.2384, 0.2334, 0.2370], [0.2337, 0.2288, 0.2325], [0.2270, 0.2221, 0.2261],
     [0.2209, 0.2161, 0.2203], [0.2127, 0.2082, 0.2126], [0.2087, 0.2043, 0.2088]],
    [[0.2384, 0.2334, 0.2370], [0.2337, 0.2287, 0.2325], [0.2267, 0.2218, 0.2258],
     [0.2184, 0.2137, 0.2180], [0.2048, 0.2006, 0.2051], [0.1950, 0.1911, 0.1957]],
    [[0.2384, 0.2334, 0.2370], [0.2337, 0.2287, 0.2325], [0.2266, 0.2217, 0.2258],
     [0.2182, 0.2136, 0.2178], [0.2012, 0.1972, 0.2017], [0.1824, 0.1790, 0.1838]],
])
_BLUR_STD_SCALED = (_BLUR_STD * 0.229) / 0.2384     # utils.py:232-234


def get_norm_params(blur_dicts, use_custom_image_norm):
    """Returns (means, stds), each float64 [B,3] (or [1,3] when blur_dicts is None).
    Reference utils.py:219-273, including its quirk that a blurred image whose param_index is
    outside 0..2 (the stored-PSF off-by-one, transforms.py:427-428) keeps all-zero rows."""
    if blur_dicts is None:
        return np.array([_CANONICAL_MEAN]), np.array([_CANONICAL_STD])
    n = len(blur_dicts)
    means, stds = np.zeros((n, 3)), np.zeros((n, 3))
    for i, bd in enumerate(blur_dicts):
        custom = use_custom_image_norm and bd["blurring"] and bd["param_index"] is not None
        if not custom or bd["fraction_index"] == -1:
            means[i], stds[i] = _CANONICAL_MEAN, _CANONICAL_STD
        elif bd["param_index"] in (0, 1, 2):
            means[i] = _CANONICAL_MEAN
            stds[i] = _BLUR_STD_SCALED[bd["param_index"], bd["fraction_index"] + 1]
    return means, stds


def collate_fn(batch):
    return tuple(zip(*batch))
