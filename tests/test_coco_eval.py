"""COCO box evaluation without pycocotools: the IoU kernel against the
reference's C bbIou (oracle/_ref/libmaskapi.so built from cocoapi/common/maskApi.c, and golden vectors
made with it), the evaluator against the 12 statistics, precision and recall arrays of the reference's
own COCOeval run on the same synthetic ground truth / detections (tests/golden/coco.npz)."""
import numpy as np
import pytest
import torch

import gen_goldens as GG
import ref_maskapi
from detectinblur_amd.coco_eval import CocoBoxEvaluator, _iou_cpu


def _run(device):
    gt, dt = GG.coco_eval_inputs()
    ev = CocoBoxEvaluator(gt, device=device)
    ev.update(dt)
    stats = ev.summarize()
    return ev, stats


def test_evaluator_matches_reference_cocoeval_cpu(golden):
    ev, stats = _run(None)
    assert np.array_equal(ev.precision, golden.coco["coco_precision"])
    assert np.array_equal(ev.recall, golden.coco["coco_recall"])
    assert np.allclose(stats, golden.coco["coco_stats"], rtol=0, atol=1e-15)


def test_numpy_iou_restatement_matches_reference_c(golden):
    d, g, c = golden.coco["iou_dt"], golden.coco["iou_gt"], golden.coco["iou_crowd"]
    assert np.array_equal(_iou_cpu(d, g, c), golden.coco["iou_out"])
    assert np.array_equal(_iou_cpu(d, g, np.zeros_like(c)), golden.coco["iou_out_nocrowd"])
    if ref_maskapi.available():          # the reference C routine itself, when it was built here
        assert np.array_equal(ref_maskapi.bb_iou(d, g, c), golden.coco["iou_out"])


@pytest.mark.gpu
def test_gpu_iou_bit_exact_vs_reference_c(golden):
    from detectinblur_amd.models.detector_ops import coco_box_iou
    d, g, c = golden.coco["iou_dt"], golden.coco["iou_gt"], golden.coco["iou_crowd"]
    got = coco_box_iou(torch.from_numpy(d).cuda(), torch.from_numpy(g).cuda(), torch.from_numpy(c).cuda()).cpu().numpy()
    assert np.array_equal(got, golden.coco["iou_out"])
    got = coco_box_iou(torch.from_numpy(d).cuda(), torch.from_numpy(g).cuda(), None).cpu().numpy()
    assert np.array_equal(got, golden.coco["iou_out_nocrowd"])
    if ref_maskapi.available():
        rs = np.random.RandomState(9)
        d2 = np.concatenate([rs.uniform(0, 900, (300, 2)), np.exp(rs.uniform(-2, 6, (300, 2)))], 1)
        g2 = np.concatenate([rs.uniform(0, 900, (170, 2)), np.exp(rs.uniform(-2, 6, (170, 2)))], 1)
        c2 = (rs.random_sample(170) < 0.2).astype(np.uint8)
        got = coco_box_iou(torch.from_numpy(d2).cuda(), torch.from_numpy(g2).cuda(), torch.from_numpy(c2).cuda()).cpu().numpy()
        assert np.array_equal(got, ref_maskapi.bb_iou(d2, g2, c2))


@pytest.mark.gpu
def test_evaluator_on_gpu_matches_reference_cocoeval(golden):
    ev, stats = _run("cuda")
    assert np.array_equal(ev.precision, golden.coco["coco_precision"])
    assert np.allclose(stats, golden.coco["coco_stats"], rtol=0, atol=1e-15)


def test_native_matching_equals_the_interpreted_loop_nest():
    """dib_coco_match / dib_coco_match_image (csrc/host/dib_host.c) against the loop nest in Python they restate (pycocotools
    evaluateImg, reference cocoapi/PythonAPI/pycocotools/cocoeval.py:235-310): random detections and ground truth with crowds, exact
    hits, IoUs on the thresholds, equal scores, more than maxDets detections of one category, categories with only ground truth or
    only detections, empty images."""
    from detectinblur_amd.coco_eval import CocoBoxEvaluator as E
    rng = np.random.default_rng(0)
    for trial in range(60):
        D, G = int(rng.integers(0, 260)), int(rng.integers(0, 25))
        cats = [1, 2, 3, 5, 8, 13, 90]
        boxes = np.concatenate((rng.random((D, 2)) * 300, 5 + rng.random((D, 2)) * 150), axis=1)
        scores = np.round(rng.random(D) * 50) / 50 if trial % 2 else rng.random(D)
        labels = rng.choice(cats[:3] if trial % 4 == 0 else cats, size=D)
        g = dict(labels=rng.choice(cats, size=G), crowd=(rng.random(G) < 0.2).astype(np.int64), area=rng.random(G) * 30000,
                 boxes=np.zeros((G, 4)))
        iou_all = np.round(rng.random((D, G)) * 20) / 20 if trial % 3 == 0 else rng.random((D, G))
        ev_c, ev_py = E({}, cats=cats), E({}, cats=cats)
        ev_c._match_image(7, boxes, scores, labels, g, iou_all)
        ev_py._match_image_py(7, boxes, scores, labels, g, iou_all)
        assert set(ev_c.results) == set(ev_py.results)
        for key, recs in ev_py.results.items():
            for x, y in zip(ev_c.results[key], recs):
                assert x["n_gt"] == y["n_gt"] and np.array_equal(x["scores"], y["scores"])
                assert x["dtm"].dtype == bool and np.array_equal(x["dtm"], y["dtm"]) and np.array_equal(x["dt_ig"], y["dt_ig"])
    # and the per-category routine on its own
    for trial in range(200):
        D, G = int(rng.integers(0, 40)), int(rng.integers(0, 12))
        ious = np.round(rng.random((D, G)) * 20) / 20 if trial % 2 else rng.random((D, G))
        args = (ious, np.sort(rng.random(D))[::-1].copy(), rng.random(D) * 20000, (rng.random(G) < 0.3).astype(np.int64), rng.random(G) * 20000)
        for x, y in zip(E._match(*args), E._match_py(*args)):
            assert x["n_gt"] == y["n_gt"] and np.array_equal(x["dtm"], y["dtm"]) and np.array_equal(x["dt_ig"], y["dt_ig"])


def test_native_accumulate_equals_the_interpreted_form():
    """dib_coco_accumulate_cat against the interpreted accumulate (pycocotools COCOeval.accumulate, reference
    cocoapi/PythonAPI/pycocotools/cocoeval.py:315-420): equal scores across images (stable ranking), categories without counted
    ground truth, images with more than maxDets detections, empty categories -- bit-identical precision and recall arrays."""
    from detectinblur_amd.coco_eval import CocoBoxEvaluator as E
    rng = np.random.default_rng(3)
    cats = list(range(1, 31))
    ev = E({}, cats=cats)
    for img in range(60):
        D, G = int(rng.integers(0, 130)), int(rng.integers(0, 10))
        boxes = np.concatenate((rng.random((D, 2)) * 300, 5 + rng.random((D, 2)) * 150), axis=1)
        scores = np.round(rng.random(D) * 40) / 40 if img % 2 else rng.random(D)
        labels = rng.choice(cats[:1] if img % 7 == 0 else (cats[:6] if img % 3 == 0 else cats[:-2]), size=D)
        g = dict(labels=rng.choice(cats[:-1], size=G), crowd=(rng.random(G) < 0.2).astype(np.int64), area=rng.random(G) * 30000, boxes=np.zeros((G, 4)))
        ev.images.append(img)
        ev._match_image(img, boxes, scores, labels, g, np.round(rng.random((D, G)) * 20) / 20)
    p1, r1 = ev.accumulate()
    p2, r2 = ev.accumulate_py()
    assert np.array_equal(p1, p2) and np.array_equal(r1, r2)
    assert (p1 > -1).sum() > 10000 and (p1[:, :, -1] == -1).all()          # the last category never occurs: untouched
    s1 = ev.summarize()
    ev.precision, ev.recall = p2, r2
    assert np.array_equal(s1, ev.summarize())
