"""Writes tests/golden/masks.npz + masks.json: segmentation masks and the model transform's handling of them, from the REFERENCE.

TEST INFRASTRUCTURE ONLY.  Runs in the build container (needs /root/reference and oracle/_ref/libmaskapi.so = the reference's own
cocoapi/common/maskApi.c compiled by oracle/Makefile):
    python oracle/gen_mask_goldens.py
  * polygons -> masks: maskApi.c rleFrPoly (:162-218) + rleDecode (:43-47) through oracle/ref_maskapi.py, for the seeded
    polygons of `polygon_cases()`; objects of several parts are reduced with `any` as reference coco_utils.py:37-43 does;
  * the detector's input transform on targets that carry masks and keypoints (reference models/net_transforms.py:36-56 masks,
    :165-172 + :283-299 keypoints), run through the imported reference class (oracle/ref_harness.load_net_transforms).
The fixture holds inputs and expected outputs only."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness          # noqa: E402
import ref_maskapi          # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def polygon_cases():
    """(name, segmentation as COCO json holds it, height, width): polygons with fractional / integral / out-of-image vertices,
    objects of several polygons, degenerate outlines, boxes (four numbers), one uncompressed RLE."""
    rs = np.random.RandomState(20240)
    cases = []
    for i in range(24):
        h, w = int(rs.randint(8, 70)), int(rs.randint(8, 90))
        parts = []
        for _ in range(int(rs.randint(1, 4))):
            k = int(rs.randint(3, 10))
            xy = rs.uniform(-6, max(h, w) + 6, size=2 * k)
            if i % 3 == 0:
                xy = np.round(xy)
            if i % 5 == 0:
                xy = np.round(xy * 2) / 2
            parts.append([float(v) for v in xy])
        cases.append(("poly%d" % i, parts, h, w))
    cases.append(("point", [[5.0, 5.0, 5.0, 5.0, 5.0, 5.0]], 12, 12))
    cases.append(("line", [[1.0, 1.0, 9.0, 7.0, 1.0, 1.0]], 10, 12))
    cases.append(("outside", [[-9.0, -9.0, -2.0, -9.0, -2.0, -2.0]], 10, 10))
    cases.append(("cover", [[-1.0, -1.0, 30.0, -1.0, 30.0, 30.0, -1.0, 30.0]], 9, 11))
    cases.append(("boxes", [[2.0, 3.0, 5.0, 4.0], [0.5, 0.5, 3.25, 2.0]], 12, 14))
    cases.append(("rle", {"size": [6, 5], "counts": [3, 4, 2, 7, 5, 9]}, 6, 5))
    return cases


def ref_object_mask(seg, h, w):
    """pycocotools frPyObjects + decode + any (reference coco_utils.py:37-43) on top of the reference's C functions."""
    out = np.zeros((h, w), dtype=np.uint8)

    def poly(p):
        return ref_maskapi.poly_mask(p, h, w)[0]

    def box(b):
        xs, ys, xe, ye = b[0], b[1], b[0] + b[2], b[1] + b[3]
        return poly([xs, ys, xs, ye, xe, ye, xe, ys])          # maskApi.c:148-156

    if isinstance(seg, dict):
        buf, v = [], 0
        for c in seg["counts"]:
            buf += [v] * int(c)
            v = 1 - v
        return np.array(buf[:h * w] + [0] * max(0, h * w - len(buf)), dtype=np.uint8).reshape(w, h).T.copy()
    for part in seg:                      # _mask.pyx:292-295: a list of boxes (four numbers each) or of polygons (more)
        out |= box(part) if len(seg[0]) == 4 else poly(part)
    return out


def transform_inputs():
    g = torch.Generator().manual_seed(99)
    imgs = [torch.rand(3, 50, 70, generator=g), torch.rand(3, 60, 45, generator=g)]
    tgts = []
    for (h, w), n in (((50, 70), 3), ((60, 45), 1)):
        masks = (torch.rand(n, h, w, generator=g) > 0.6).to(torch.uint8)
        kp = torch.cat((torch.rand(n, 4, 2, generator=g) * torch.tensor([float(w), float(h)]), torch.ones(n, 4, 1)), dim=2)
        boxes = torch.tensor([[3.5, 4.25, 40.0, 44.5]] * n)
        tgts.append({"boxes": boxes, "labels": torch.arange(1, n + 1), "masks": masks, "keypoints": kp})
    return imgs, tgts


def main():
    assert ref_harness.available() and ref_maskapi.available()
    store, meta = {}, {"cases": []}
    for name, seg, h, w in polygon_cases():
        store["mask_" + name] = ref_object_mask(seg, h, w)
        meta["cases"].append({"name": name, "segmentation": seg, "height": h, "width": w})
    NT = ref_harness.load_net_transforms()
    imgs, tgts = transform_inputs()
    for tag, training in (("train", True), ("eval", False)):
        t = NT.GeneralizedRCNNTransform(64, 100, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], training=training)
        torch.manual_seed(5)
        il, out = t([i.clone() for i in imgs], [{k: v.clone() for k, v in d.items()} for d in tgts])
        store["nt_%s_sizes" % tag] = np.array(il.image_sizes)
        for k, d in enumerate(out):
            for f in ("boxes", "masks", "keypoints"):
                store["nt_%s_%s%d" % (tag, f, k)] = d[f].numpy()
    np.savez_compressed(os.path.join(GOLD, "masks.npz"), **store)
    with open(os.path.join(GOLD, "masks.json"), "w") as f:
        json.dump(meta, f)
    print("wrote %d arrays" % len(store))


if __name__ == "__main__":
    main()
