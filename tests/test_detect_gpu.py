"""Box bookkeeping kernels of the training step (csrc/dib_detect.hip: dib_box_match / dib_box_encode_matched / dib_box_decode /
dib_box_pool / dib_box_labels) against the tensor code they replace on CUDA tensors (detectinblur_amd/models/detector_ops.py:
box_iou + Matcher + BoxCoder, torchvision's arithmetic as the reference's models/faster_rcnn.py:150-159,198-229 configures it).
Bit-exact: integer matches and labels, and float32 boxes / targets computed with the same operations in the same order."""
import math

import pytest
import torch

from detectinblur_amd.models import detector_ops as ops

pytestmark = pytest.mark.gpu


def _boxes(g, n, W=300.0, H=200.0):
    xy = torch.rand(n, 2, generator=g) * torch.tensor([W - 40.0, H - 40.0])
    wh = 2 + torch.rand(n, 2, generator=g) * torch.tensor([W / 2.0, H / 2.0])
    return torch.cat((xy, xy + wh), dim=1)


def _case(seed, counts, M, per_image_cands):
    g = torch.Generator().manual_seed(seed)
    gts = [_boxes(g, n).cuda() for n in counts]
    cand = torch.stack([_boxes(g, M) for _ in counts]) if per_image_cands else _boxes(g, M)
    return gts, cand.cuda()


def _torch_match(matcher, gts, cand):
    out = []
    for i, g in enumerate(gts):
        c = cand[i] if cand.dim() == 3 else cand
        out.append(matcher(ops.box_iou(g, c)) if g.shape[0] else torch.full((c.shape[0],), -1, dtype=torch.int64, device=c.device))
    return torch.stack(out)


@pytest.mark.parametrize("allow_low", [True, False])
@pytest.mark.parametrize("per_image", [False, True])
def test_match_equals_box_iou_plus_matcher(allow_low, per_image):
    """Ragged ground truth (0, 1, 7, 256 boxes), 5000 candidates: ties between duplicate ground-truth boxes (lowest index wins),
    candidates equal to a ground truth (IoU exactly 1), candidates far from everything, a zero-area candidate ON a zero-area
    ground truth (0 / 0 = NaN, which torch's max propagates), thresholds that many IoUs fall close to."""
    counts = [7, 0, 1, 256, 33]
    gts, cand = _case(3, counts, 5000, per_image)
    gts[0][5] = gts[0][2]                                               # duplicates: argmax ties
    gts[3][200] = gts[3][10]
    point = torch.tensor([50.0, 60.0, 50.0, 60.0]).cuda()
    gts[4][3] = point                                                    # zero area
    if per_image:
        cand[0, 11] = gts[0][2]; cand[3, 12] = gts[3][10]; cand[4, 13] = point; cand[2, 14] = torch.tensor([1e4, 1e4, 1e4 + 5, 1e4 + 5]).cuda()
    else:
        cand[11] = gts[0][2]; cand[12] = gts[3][10]; cand[13] = point; cand[14] = torch.tensor([1e4, 1e4, 1e4 + 5, 1e4 + 5]).cuda()
    for high, low in ((0.7, 0.3), (0.5, 0.5), (0.05, 0.01)):
        matcher = ops.Matcher(high, low, allow_low_quality_matches=allow_low)
        gt_cat, offs = ops.cat_boxes(gts)
        assert offs == [0, 7, 7, 8, 264, 297]
        got = ops.match_boxes_hip(matcher, gt_cat, offs, cand, shared=not per_image)
        want = _torch_match(matcher, gts, cand)
        assert got.dtype == torch.int64 and got.shape == want.shape
        assert torch.equal(got, want), (high, low, int((got != want).sum()))
        assert (got[1] == -1).all()
        assert {int(v) for v in got.unique()} >= {-1, 0}


def test_match_every_image_empty_and_limits():
    cand = _boxes(torch.Generator().manual_seed(0), 100).cuda()
    m = ops.Matcher(0.7, 0.3, True)
    gt_cat, offs = ops.cat_boxes([cand.new_zeros((0, 4))] * 3)
    assert gt_cat is None and offs == [0, 0, 0, 0]
    assert (ops.match_boxes_hip(m, gt_cat, offs, cand, shared=True) == -1).all()
    from detectinblur_amd import _lib
    with pytest.raises(_lib.DibError):                                   # more ground truth than the kernel's LDS table
        big, o = ops.cat_boxes([_boxes(torch.Generator().manual_seed(1), 257).cuda()])
        ops.match_boxes_hip(m, big, o, cand, shared=True)
    with pytest.raises(_lib.DibError):
        ops.match_boxes_hip(m, None, [0] * 34, cand, shared=True)       # 33 images


@pytest.mark.parametrize("per_image", [False, True])
def test_encode_matched_equals_boxcoder_encode(per_image):
    counts = [7, 0, 1, 40]
    gts, cand = _case(5, counts, 3000, per_image)
    gt_cat, offs = ops.cat_boxes(gts)
    g = torch.Generator().manual_seed(6)
    match = torch.stack([torch.randint(-2, max(n, 1), (3000,), generator=g) for n in counts]).cuda()
    for weights in ((1.0, 1.0, 1.0, 1.0), (10.0, 10.0, 5.0, 5.0)):
        coder = ops.BoxCoder(weights)
        tg, mb = ops.encode_matched_hip(coder, gt_cat, offs, match, cand, shared=not per_image, want_targets=True, want_matched=True)
        for i, gt in enumerate(gts):
            c = cand[i] if per_image else cand
            ref = gt[match[i].clamp(min=0)] if gt.shape[0] else torch.zeros_like(c)
            assert torch.equal(mb[i], ref)
            want = coder.encode(ref, c)
            assert torch.equal(tg[i].isnan(), want.isnan())
            assert torch.equal(tg[i].nan_to_num(7.0, 8.0, 9.0), want.nan_to_num(7.0, 8.0, 9.0)), (i, (tg[i] - want).abs().nan_to_num().max())
    only_targets = ops.encode_matched_hip(coder, gt_cat, offs, match, cand, shared=not per_image)
    assert only_targets[1] is None and torch.equal(only_targets[0].nan_to_num(7.0, 8.0, 9.0), tg.nan_to_num(7.0, 8.0, 9.0))


def test_decode_equals_boxcoder_decode():
    """Deltas beyond the log(1000 / 16) clip, NaN deltas, anchors repeated for every image without a repeated tensor."""
    g = torch.Generator().manual_seed(8)
    A, N = 7000, 3
    anchors = _boxes(g, A).cuda()
    deltas = (torch.randn(N * A, 4, generator=g) * torch.tensor([0.5, 0.5, 2.5, 2.5])).cuda()
    deltas[5] = float("nan")
    deltas[6, 2] = 50.0
    for weights in ((1.0, 1.0, 1.0, 1.0), (10.0, 10.0, 5.0, 5.0)):
        coder = ops.BoxCoder(weights)
        got = ops.decode_boxes_hip(coder, deltas, anchors)
        want = coder.decode(deltas, torch.cat([anchors] * N)).reshape(-1, 4)
        assert torch.equal(got.isnan(), want.isnan()) and bool(got[5].isnan().all())
        assert torch.equal(got.nan_to_num(0.0), want.nan_to_num(0.0)), float((got - want).abs().nan_to_num().max())
    assert float(got[6, 2] - got[6, 0]) == pytest.approx(float(math.exp(coder.clip) * (anchors[6, 2] - anchors[6, 0])), rel=1e-5)


def test_pool_and_labels_equal_the_tensor_form():
    counts, P = [4, 0, 9, 1], 300
    g = torch.Generator().manual_seed(12)
    gts = [_boxes(g, n).cuda() for n in counts]
    labels = [torch.randint(1, 91, (n,), generator=g).cuda() for n in counts]
    props = torch.stack([_boxes(g, P) for _ in counts]).cuda()
    props[0, :4] = gts[0] + 1.5
    props[2, :9] = gts[2] - 1.0
    ok = (torch.arange(P)[None, :] < torch.tensor([300, 300, 280, 20])[:, None]).cuda()
    gt_cat, offs = ops.cat_boxes(gts)
    G = max(counts)
    cands = ops.pool_boxes_hip(props, gt_cat, offs, G)
    pad, valid = ops.pad_boxes(gts)
    unit = torch.tensor([0.0, 0.0, 1.0, 1.0]).cuda()
    want = torch.cat((props, torch.where(valid[..., None], pad, unit)), dim=1)
    assert torch.equal(cands, want)
    matcher = ops.Matcher(0.5, 0.5, False)
    m = ops.match_boxes_hip(matcher, gt_cat, offs, cands, shared=False)
    m_want = ops.match_batched(matcher, ops.box_iou_batched(pad, want), valid)
    assert torch.equal(m, m_want)
    lab = ops.pool_labels_hip(m, torch.cat(labels), offs, ok, P)
    live = torch.cat((ok, valid), dim=1)
    for i, n in enumerate(counts):
        w = labels[i][m[i].clamp(min=0)] if n else torch.zeros_like(m[i])
        w = torch.where(m[i] == -1, 0, w); w = torch.where(m[i] == -2, -1, w)
        assert torch.equal(lab[i], torch.where(live[i], w, -1)), i
    assert int((lab > 0).sum()) >= 13 and (lab[1][ok[1].nonzero().flatten()] == 0).all()
    assert torch.equal(ops.pool_labels_hip(m, torch.cat(labels), offs, None, P)[:, :P][ok], lab[:, :P][ok])


def _detector_parts():
    from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
    torch.manual_seed(0)
    return fasterrcnn_resnet50_fpn(pretrained=False, pretrained_backbone=False, num_classes=91, box_batch_size_per_image=64)


def test_rpn_and_roi_training_paths_equal_the_tensor_paths(monkeypatch):
    """The modules' own switch (ops.HIP_BOXES): RPN labels / regression targets / proposals and the RoI heads' samples with
    the kernels against the same calls with the tensor expressions on the same device."""
    from tests.test_detector_ops import _ragged_targets
    model = _detector_parts()
    rpn, heads = model.rpn, model.roi_heads
    g = torch.Generator().manual_seed(4)
    anchors = _boxes(g, 6000).cuda()
    targets = _ragged_targets(9, [5, 0, 1, 12], device="cuda")
    targets[3]["boxes"][7] = targets[3]["boxes"][2]
    props = torch.stack([_boxes(g, 300) for _ in range(4)]).cuda()
    props[3, :12] = targets[3]["boxes"] + 1.0
    ok = (torch.arange(300)[None, :] < torch.tensor([300, 300, 280, 20])[:, None]).cuda()
    deltas = (torch.randn(4 * 6000, 4, generator=g) * 0.3).cuda()
    out = {}
    for flag in (True, False):
        monkeypatch.setattr(ops, "HIP_BOXES", flag)
        lab, reg = rpn.assign_and_encode([anchors] * 4, targets)
        lab2, matched = rpn.assign_targets([anchors] * 4, targets)
        torch.manual_seed(21); torch.cuda.manual_seed(21)
        sel = heads.select_training_samples((props.clone(), ok.clone()), [dict(t) for t in targets])
        out[flag] = (lab, reg, lab2, matched, torch.stack(sel[0]), sel[1], sel[2], sel[3])
    for a, b in zip(out[True], out[False]):
        assert a.shape == b.shape and a.dtype == b.dtype
        assert torch.equal(a.nan_to_num(7.0, 8.0, 9.0), b.nan_to_num(7.0, 8.0, 9.0)) if a.is_floating_point() else torch.equal(a, b)
    assert int((out[True][0] == 1).sum()) > 10 and int((out[True][5] > 0).sum()) > 0


def _head_outputs(seed, R, C=91, spread=1.5):
    g = torch.Generator().manual_seed(seed)
    logits = (torch.randn(R, C, generator=g) * spread).cuda()
    deltas = (torch.randn(R, 4 * C, generator=g) * torch.tensor([1.0, 1.0, 2.0, 2.0]).repeat(C)).cuda()
    rois = _boxes(g, R, 1333.0, 800.0).cuda()
    return logits, deltas, rois


def test_det_candidates_equal_softmax_decode_clip_and_tests():
    """Scores bit-equal to F.softmax (ATen's reduction order repeated), boxes bit-equal to clip(BoxCoder.decode), the same candidates
    dropped, count and largest coordinate as torch computes them."""
    import torch.nn.functional as F
    coder = ops.BoxCoder((10.0, 10.0, 5.0, 5.0))
    for R, shape in ((1000, (800, 1333)), (37, (480, 640)), (3, (800, 1088))):
        logits, deltas, rois = _head_outputs(R, R)
        logits[0, 5] = 30.0                                                              # a saturated row
        deltas[1, 8:12] = float("nan")                                                   # class 2 of RoI 1: NaN box -> dropped
        s, b, stats = ops.det_candidates_hip(coder, logits, deltas, rois, shape, 0.05, 1e-2)
        want_s = F.softmax(logits, -1)[:, 1:].t()
        want_b = ops.clip_boxes_to_image(coder.decode(deltas, rois), shape)[:, 1:].permute(1, 0, 2)
        ok = (want_s > 0.05) & ((want_b[..., 2] - want_b[..., 0]) >= 1e-2) & ((want_b[..., 3] - want_b[..., 1]) >= 1e-2)
        assert torch.equal(s > float("-inf"), ok) and torch.equal(s[ok], want_s[ok])
        assert torch.equal(b.isnan(), want_b.isnan()) and torch.equal(b.nan_to_num(0.0), want_b.nan_to_num(0.0))
        assert int(stats[0]) == int(ok.sum()) > 0 and not bool(ok[1, 1])
        assert float(stats[1:2].view(torch.float32)) == float(want_b[ok].max())


@pytest.mark.parametrize("R,spread", [(1000, 1.5), (1000, 4.0), (37, 1.5), (1, 1.5)])
def test_detections_equal_the_tensor_path(R, spread, monkeypatch):
    """RoIHeads.postprocess_detections with the kernels (one NMS set per class on boxes moved apart exactly as batched_nms moves
    them) against the tensor path on the same head outputs: the same detections, scores and labels in the same order."""
    heads = _detector_parts().roi_heads
    logits, deltas, rois = _head_outputs(100 + R, R, spread=spread)
    out = {}
    for flag in (True, False):
        monkeypatch.setattr(ops, "HIP_BOXES", flag)
        out[flag] = heads.postprocess_detections(logits, deltas, [rois], [(800, 1333)])[0]
    a, b = out[True], out[False]
    assert a["labels"].dtype == b["labels"].dtype == torch.int64 and a["boxes"].shape == b["boxes"].shape
    assert torch.equal(a["scores"], b["scores"]) and torch.equal(a["labels"], b["labels"]) and torch.equal(a["boxes"], b["boxes"])
    assert a["scores"].numel() == (min(100, a["scores"].numel()) if R > 1 else a["scores"].numel()) and (R < 37 or a["scores"].numel() > 10)
    assert (a["scores"][:-1] >= a["scores"][1:]).all()


def test_detections_two_images_and_none_left(monkeypatch):
    heads = _detector_parts().roi_heads
    l1, d1, r1 = _head_outputs(7, 300)
    l2, d2, r2 = _head_outputs(8, 200)
    out = {}
    for flag in (True, False):
        monkeypatch.setattr(ops, "HIP_BOXES", flag)
        out[flag] = heads.postprocess_detections(torch.cat((l1, l2)), torch.cat((d1, d2)), [r1, r2], [(800, 1333), (600, 900)])
    for a, b in zip(out[True], out[False]):
        assert torch.equal(a["scores"], b["scores"]) and torch.equal(a["labels"], b["labels"]) and torch.equal(a["boxes"], b["boxes"])
    monkeypatch.setattr(ops, "HIP_BOXES", True)
    heads.score_thresh = 0.999
    none = heads.postprocess_detections(l1, d1, [r1], [(800, 1333)])[0]
    assert none["boxes"].shape == (0, 4) and none["scores"].numel() == 0 and none["labels"].dtype == torch.int64
