"""Segmentation masks of ConvertCocoPolysToMask (reference coco_utils.py:34-104) and what the detector's input
transform does to targets that carry masks and keypoints (reference models/net_transforms.py:36-56, :165-172, :283-299).
  * tests/golden/masks.{npz,json}: the reference's own cocoapi/common/maskApi.c (rleFrPoly + rleDecode, compiled by oracle/Makefile)
    and its own GeneralizedRCNNTransform on seeded inputs, written by oracle/gen_mask_goldens.py;
  * where oracle/_ref/libmaskapi.so is present (the build container): 2,000 random polygons straight against the reference's C."""
import json
import os

import numpy as np
import pytest
import torch

import gen_mask_goldens as GM
from detectinblur_amd import coco_utils
from detectinblur_amd.models.net_transforms import GeneralizedRCNNTransform

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
G = np.load(os.path.join(GOLD, "masks.npz"))
with open(os.path.join(GOLD, "masks.json")) as f:
    META = json.load(f)


@pytest.mark.parametrize("case", META["cases"], ids=[c["name"] for c in META["cases"]])
def test_object_masks_equal_the_reference(case):
    got = coco_utils._one_object_mask(case["segmentation"], case["height"], case["width"])
    want = G["mask_" + case["name"]]
    assert got.dtype == np.uint8 and got.shape == want.shape and np.array_equal(got, want)


def test_random_polygons_against_the_reference_library():
    import ref_maskapi
    if not ref_maskapi.available():
        pytest.skip("oracle/_ref/libmaskapi.so (the reference's maskApi.c) is built in the build container only")
    rs = np.random.RandomState(7)
    for it in range(2000):
        h, w = int(rs.randint(1, 90)), int(rs.randint(1, 120))
        k = int(rs.randint(1, 12))
        xy = rs.uniform(-15, max(h, w) + 15, size=2 * k)
        if it % 3 == 0:
            xy = np.round(xy)
        if it % 7 == 0:
            xy = np.round(xy * 2) / 2
        if it % 11 == 0:
            xy[:] = xy[0]
        want = ref_maskapi.poly_mask(xy, h, w)[0]
        if len(xy) > 4:
            got = coco_utils._one_object_mask([[float(v) for v in xy]], h, w)
            assert np.array_equal(got, want), (it, h, w, xy.tolist())


def test_convert_coco_polys_to_mask_target():
    """Keys in the reference's order, crowd objects dropped, degenerate boxes dropped from boxes / labels / masks / keypoints but
    not from area / iscrowd (reference coco_utils.py:51-104)."""
    from PIL import Image
    img = Image.new("RGB", (40, 30))
    tri, quad = [[2.0, 2.0, 20.0, 3.0, 10.0, 25.0]], [[5.0, 5.0, 30.0, 5.0, 30.0, 20.0, 5.0, 20.0], [32.0, 22.0, 38.0, 22.0, 35.0, 28.0]]
    anno = [{"bbox": [2, 2, 18, 23], "category_id": 3, "iscrowd": 0, "area": 100.0, "segmentation": tri, "keypoints": [3, 4, 2, 5, 6, 1]},
            {"bbox": [5, 5, 0, 15], "category_id": 4, "iscrowd": 0, "area": 0.0, "segmentation": quad, "keypoints": [1, 1, 0, 2, 2, 0]},
            {"bbox": [1, 1, 9, 9], "category_id": 5, "iscrowd": 1, "area": 81.0, "segmentation": {"size": [30, 40], "counts": [1200]}},
            {"bbox": [5, 5, 25, 23], "category_id": 6, "iscrowd": 0, "area": 300.0, "segmentation": quad, "keypoints": [9, 9, 2, 8, 8, 2]}]
    _, t, bd = coco_utils.ConvertCocoPolysToMask()(img, {"image_id": 17, "annotations": anno})
    assert list(t) == ["boxes", "labels", "masks", "image_id", "keypoints", "area", "iscrowd"] and bd == {}
    assert t["labels"].tolist() == [3, 6] and t["boxes"].shape == (2, 4)
    assert t["masks"].dtype == torch.uint8 and t["masks"].shape == (2, 30, 40)
    assert np.array_equal(t["masks"][0].numpy(), coco_utils._one_object_mask(tri, 30, 40))
    both = coco_utils._one_object_mask(quad[:1], 30, 40) | coco_utils._one_object_mask(quad[1:], 30, 40)
    assert np.array_equal(t["masks"][1].numpy(), both) and both.sum() > coco_utils._one_object_mask(quad[:1], 30, 40).sum()
    assert t["keypoints"].shape == (2, 2, 3) and t["keypoints"][1].tolist() == [[9, 9, 2], [8, 8, 2]]
    assert t["area"].tolist() == [100.0, 0.0, 300.0] and t["iscrowd"].tolist() == [0, 0, 0]
    _, t, _ = coco_utils.ConvertCocoPolysToMask(with_masks=False)(img, {"image_id": 17, "annotations": anno})
    assert "masks" not in t
    _, t, _ = coco_utils.ConvertCocoPolysToMask()(img, {"image_id": 3, "annotations": []})
    assert t["masks"].shape == (0, 30, 40) and t["boxes"].shape == (0, 4)
    with pytest.raises(Exception, match="input type is not supported"):
        coco_utils._one_object_mask("abc", 4, 4)
    with pytest.raises(TypeError):          # a flat list of numbers: pycocotools' frPyObjects takes len() of its first element too
        coco_utils._one_object_mask([3.0, 2.0, 11.0, 2.5, 8.0, 9.0], 12, 14)


@pytest.mark.parametrize("device", ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def test_transform_resizes_masks_and_keypoints_like_the_reference(device):
    imgs, tgts = GM.transform_inputs()
    for tag, training in (("train", True), ("eval", False)):
        t = GeneralizedRCNNTransform(64, 100, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], training=training)
        torch.manual_seed(5)
        il, out = t([i.clone().to(device) for i in imgs], [{k: v.clone().to(device) for k, v in d.items()} for d in tgts])
        assert np.array_equal(np.array(il.image_sizes), G["nt_%s_sizes" % tag])
        for k, d in enumerate(out):
            assert d["masks"].dtype == torch.uint8
            assert np.array_equal(d["masks"].cpu().numpy(), G["nt_%s_masks%d" % (tag, k)])
            for f in ("boxes", "keypoints"):
                assert np.allclose(d[f].cpu().numpy(), G["nt_%s_%s%d" % (tag, f, k)], rtol=0, atol=0 if device == "cpu" else 1e-5), (tag, f)
