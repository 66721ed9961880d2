"""`--aspect-ratio-group-factor k` (reference train.py:401, :192-198; group_by_aspect_ratio.py): training
batches hold images of similar aspect ratio, so the detector's batch padding (net_transforms.batch_images)
wastes little.  Aspect ratios come from the annotation file (no image is opened); datasets that cannot say
(`get_height_and_width` / COCO metadata / a fixed synthetic size) are read item by item."""
import bisect
import math
from collections import defaultdict

import numpy as np
import torch.utils.data
from torch.utils.data.sampler import BatchSampler, Sampler


class GroupedBatchSampler(BatchSampler):
    """Batches of `batch_size` indices that share a group id, in an order as close as possible to the base
    sampler's.  The number of batches is fixed at len(sampler) // batch_size: when the base sampler runs dry
    the fullest partial batches are topped up with indices of their own group seen earlier."""

    def __init__(self, sampler, group_ids, batch_size):
        if not isinstance(sampler, Sampler):
            raise ValueError("sampler should be an instance of torch.utils.data.Sampler, but got sampler={}".format(sampler))
        self.sampler, self.group_ids, self.batch_size = sampler, group_ids, batch_size

    def __len__(self):
        return len(self.sampler) // self.batch_size

    def __iter__(self):
        pending, seen = defaultdict(list), defaultdict(list)
        emitted, want = 0, len(self)
        for idx in self.sampler:
            g = self.group_ids[idx]
            pending[g].append(idx)
            seen[g].append(idx)
            if len(pending[g]) == self.batch_size:
                yield pending.pop(g)
                emitted += 1
        for g, part in sorted(((g, p) for g, p in pending.items() if p), key=lambda kv: len(kv[1]), reverse=True):
            if emitted >= want:
                break
            need = self.batch_size - len(part)
            pool = seen[g] * math.ceil(need / len(seen[g]))
            yield part + pool[:need]
            emitted += 1
        assert emitted == want


def compute_aspect_ratios(dataset, indices=None):
    indices = range(len(dataset)) if indices is None else indices
    if hasattr(dataset, "get_height_and_width"):
        hw = [dataset.get_height_and_width(i) for i in indices]
        return [float(w) / float(h) for h, w in hw]
    if hasattr(dataset, "coco") and hasattr(dataset, "ids"):               # CocoDetection: straight from the json
        infos = [dataset.coco.imgs[dataset.ids[i]] for i in indices]
        return [float(m["width"]) / float(m["height"]) for m in infos]
    if isinstance(dataset, torch.utils.data.Subset):
        return compute_aspect_ratios(dataset.dataset, [dataset.indices[i] for i in indices])
    if hasattr(dataset, "size") and not callable(dataset.size):            # SyntheticCocoDetection: one fixed (H, W)
        if getattr(dataset, "sizes", None):
            return [float(dataset.size_of(i)[1]) / float(dataset.size_of(i)[0]) for i in indices]
        return [float(dataset.size[1]) / float(dataset.size[0])] * len(indices)
    out = []
    for i in indices:                                                      # last resort: load every item
        img = dataset[i][0]
        h, w = (img.height, img.width) if hasattr(img, "height") else img.shape[-2:]
        out.append(float(w) / float(h))
    return out


def create_aspect_ratio_groups(dataset, k=0):
    """Group id per item: the index of its aspect ratio among 2k+1 bin edges spaced geometrically in [1/2, 2]
    (k = 0: one edge at 1.0, i.e. portrait vs landscape)."""
    ratios = compute_aspect_ratios(dataset)
    bins = sorted((2 ** np.linspace(-1, 1, 2 * k + 1)).tolist()) if k > 0 else [1.0]
    groups = [bisect.bisect_right(bins, r) for r in ratios]
    print("Using {} as bins for aspect ratio quantization".format([0] + bins + [np.inf]))
    print("Count of instances per bin: {}".format(np.unique(groups, return_counts=True)[1]))
    return groups
