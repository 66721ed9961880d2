"""Detector ops: the plain-PyTorch fp32 restatements against brute-force definitions (CPU), and the
HIP kernels against the plain-PyTorch fp32 reference (GPU).  Tolerances are stated per test."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from detectinblur_amd.models import detector_ops as ops


def _brute_roi_align(feat, rois, scale, P, sr):
    """Literal definition: average of sr x sr bilinear samples per bin (aligned=False)."""
    K, C = rois.shape[0], feat.shape[1]
    H, W = feat.shape[-2:]
    out = torch.zeros(K, C, P, P, dtype=torch.float64)
    f = feat.double()
    for k in range(K):
        b = int(rois[k, 0])
        x1, y1, x2, y2 = [float(v) * scale for v in rois[k, 1:]]
        rw, rh = max(x2 - x1, 1.0), max(y2 - y1, 1.0)
        for ph in range(P):
            for pw in range(P):
                acc = torch.zeros(C, dtype=torch.float64)
                for iy in range(sr):
                    y = y1 + ph * rh / P + (iy + 0.5) * rh / P / sr
                    for ix in range(sr):
                        x = x1 + pw * rw / P + (ix + 0.5) * rw / P / sr
                        if y < -1 or y > H or x < -1 or x > W:
                            continue
                        yy, xx = max(y, 0.0), max(x, 0.0)
                        y0, x0 = int(yy), int(xx)
                        if y0 >= H - 1:
                            y0 = y1i = H - 1; yy = float(y0)
                        else:
                            y1i = y0 + 1
                        if x0 >= W - 1:
                            x0 = x1i = W - 1; xx = float(x0)
                        else:
                            x1i = x0 + 1
                        ly, lx = yy - y0, xx - x0
                        acc += (1 - ly) * (1 - lx) * f[b, :, y0, x0] + (1 - ly) * lx * f[b, :, y0, x1i] + ly * (1 - lx) * f[b, :, y1i, x0] + ly * lx * f[b, :, y1i, x1i]
                out[k, :, ph, pw] = acc / (sr * sr)
    return out


def _rois(rs, n, N, H, W, scale):
    x1 = rs.uniform(-5, W / scale - 4, n); y1 = rs.uniform(-5, H / scale - 4, n)
    w = rs.uniform(0.3, W / scale * 0.7, n); h = rs.uniform(0.3, H / scale * 0.7, n)
    return torch.tensor(np.stack([rs.randint(0, N, n), x1, y1, x1 + w, y1 + h], 1), dtype=torch.float32)


def test_roi_align_torch_matches_definition():
    rs = np.random.RandomState(0)
    feat = torch.tensor(rs.randn(2, 3, 11, 13), dtype=torch.float32)
    rois = _rois(rs, 9, 2, 11, 13, 0.25)
    got = ops.roi_align_torch(feat, rois, 0.25, 7, 2)
    want = _brute_roi_align(feat, rois, 0.25, 7, 2)
    assert torch.allclose(got.double(), want, atol=1e-5)


def test_nms_torch_matches_greedy_definition():
    rs = np.random.RandomState(1)
    c = rs.uniform(0, 100, (60, 2)); s = rs.uniform(5, 40, (60, 2))
    boxes = torch.tensor(np.concatenate([c - s / 2, c + s / 2], 1), dtype=torch.float32)
    scores = torch.tensor(rs.rand(60), dtype=torch.float32)
    keep = ops.nms(boxes, scores, 0.4).tolist()
    order = scores.argsort(descending=True).tolist()
    want = []
    for i in order:
        if all(float(ops.box_iou(boxes[i:i + 1], boxes[j:j + 1])) <= 0.4 for j in want):
            want.append(i)
    assert keep == want


def test_box_coder_roundtrip_and_matcher():
    rs = np.random.RandomState(2)
    a = torch.tensor(rs.uniform(0, 50, (20, 2)), dtype=torch.float32)
    p = torch.cat([a, a + torch.tensor(rs.uniform(5, 30, (20, 2)), dtype=torch.float32)], 1)
    g = p + torch.tensor(rs.uniform(-2, 2, (20, 4)), dtype=torch.float32)
    coder = ops.BoxCoder((10., 10., 5., 5.))
    assert torch.allclose(coder.decode(coder.encode(g, p), p)[:, 0], g, atol=1e-3)
    q = torch.tensor([[0.9, 0.2, 0.1, 0.45], [0.1, 0.6, 0.2, 0.40]])
    assert ops.Matcher(0.7, 0.3, False)(q.clone()).tolist() == [0, -2, -1, -2]
    assert ops.Matcher(0.7, 0.3, True)(q.clone()).tolist() == [0, 1, -1, -2]


def test_fixed_size_sampler_follows_the_sampling_rule():
    """sample_pos_neg_fixed: per row min(#pos, P) positives and min(#neg, B - n_pos) negatives, all
    drawn from the right sets, no duplicates -- the rule of sample_pos_neg without ragged outputs."""
    torch.manual_seed(0)
    rows = [(300, 5000), (10, 5000), (0, 400), (200, 30), (0, 0)]          # (#pos, #neg) of 6000 entries
    labels = torch.full((len(rows), 6000), -1.0)
    for i, (npos, nneg) in enumerate(rows):
        perm = torch.randperm(6000)
        labels[i, perm[:npos]] = 1.0
        labels[i, perm[npos:npos + nneg]] = 0.0
    B, frac = 256, 0.5
    pos_idx, pos_ok, neg_idx, neg_ok = ops.sample_pos_neg_fixed(labels, B, frac)
    assert pos_idx.shape == (len(rows), 128) and neg_idx.shape == (len(rows), 256)
    for i, (npos, nneg) in enumerate(rows):
        p, n = pos_idx[i][pos_ok[i]], neg_idx[i][neg_ok[i]]
        n_pos = min(npos, int(B * frac))
        assert p.numel() == n_pos and n.numel() == min(nneg, B - n_pos)
        assert (labels[i, p] == 1).all() and (labels[i, n] == 0).all()
        assert p.unique().numel() == p.numel() and n.unique().numel() == n.numel()
    # different draws pick different subsets, each entry about equally often
    hits = torch.zeros(6000)
    for _ in range(200):
        pi, po, _, _ = ops.sample_pos_neg_fixed(labels[:1], B, frac)
        hits[pi[0][po[0]]] += 1
    chosen = hits[labels[0] == 1]
    assert chosen.min() > 40 and chosen.max() < 140        # expectation 200 * 128 / 300 = 85


def _ragged_targets(seed, counts, H=200, W=300, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    out = []
    for n in counts:
        xy = torch.rand(n, 2, generator=g) * torch.tensor([W - 40.0, H - 40.0])
        wh = 8 + torch.rand(n, 2, generator=g) * torch.tensor([W / 2.0, H / 2.0])
        boxes = torch.cat((xy, torch.minimum(xy + wh, torch.tensor([float(W), float(H)]))), dim=1)
        out.append({"boxes": boxes.to(device), "labels": torch.randint(1, 91, (n,), generator=g).to(device)})
    return out


@pytest.mark.parametrize("device", ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def test_batched_rpn_assignment_equals_the_per_image_loop(device):
    """Anchor labelling + matched boxes for every image in one set of tensor ops (ground truth padded to the longest list)
    against the per-image loop it replaces: ragged lists, an image without any box, duplicate boxes (ties), low-quality
    matches -- identical labels and matched boxes."""
    from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
    torch.manual_seed(0)
    rpn = fasterrcnn_resnet50_fpn(pretrained=False, pretrained_backbone=False, num_classes=91).rpn
    g = torch.Generator().manual_seed(4)
    cx, cy = torch.rand(4000, generator=g) * 300, torch.rand(4000, generator=g) * 200
    w, h = 4 + torch.rand(4000, generator=g) * 150, 4 + torch.rand(4000, generator=g) * 120
    anchors = torch.stack((cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2), 1).to(device)
    targets = _ragged_targets(9, [5, 0, 1, 12, 3], device=device)
    targets[3]["boxes"][7] = targets[3]["boxes"][2]                      # a duplicate: two ground truths tie everywhere
    targets[0]["boxes"][0] = anchors[17]                                  # an exact hit (IoU 1)
    lab, matched = rpn.assign_targets([anchors] * 5, targets)
    lab2, matched2 = rpn.assign_targets_per_image([anchors] * 5, targets)
    assert lab.shape == (5, 4000) and matched.shape == (5, 4000, 4)
    assert torch.equal(lab, lab2) and torch.equal(matched, matched2)
    assert set(lab.unique().tolist()) == {-1.0, 0.0, 1.0} and float(lab[1].abs().sum()) == 0.0


@pytest.mark.parametrize("device", ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def test_batched_roi_sampling_equals_the_per_image_loop(device):
    """RoI sampling for every image at once against the per-image loop: same candidates, labels, sampled RoIs, regression
    targets and validity flags for the same generator state (ragged ground truth, an image without boxes, short proposal lists)."""
    from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
    torch.manual_seed(0)
    heads = fasterrcnn_resnet50_fpn(pretrained=False, pretrained_backbone=False, num_classes=91, box_batch_size_per_image=64).roi_heads
    targets = _ragged_targets(11, [4, 0, 9, 1], device=device)
    g = torch.Generator().manual_seed(5)
    xy = torch.rand(4, 300, 2, generator=g) * torch.tensor([250.0, 160.0])
    wh = 5 + torch.rand(4, 300, 2, generator=g) * torch.tensor([140.0, 100.0])
    boxes = torch.cat((xy, xy + wh), dim=2).to(device)
    boxes[0, :4] = targets[0]["boxes"] + 1.5                              # some proposals close to ground truth
    boxes[2, :9] = targets[2]["boxes"] - 2.0
    ok = (torch.arange(300)[None, :] < torch.tensor([300, 300, 280, 20])[:, None]).to(device)
    out = {}
    for flag in (True, False):
        heads.batched = flag
        torch.manual_seed(21)
        if device == "cuda":
            torch.cuda.manual_seed(21)
        out[flag] = heads.select_training_samples((boxes.clone(), ok.clone()), [dict(t) for t in targets])
    heads.batched = True
    (ra, la, ta, oa), (rb, lb, tb, ob) = out[True], out[False]
    assert all(torch.equal(x, y) for x, y in zip(ra, rb)) and torch.equal(la, lb) and torch.equal(oa, ob)
    assert torch.allclose(ta, tb, rtol=0, atol=0, equal_nan=True)
    assert int((la > 0).sum()) > 0 and int((la == 0).sum()) > 0


def test_masked_fastrcnn_loss_equals_the_ragged_form():
    from detectinblur_amd.models.roi_heads import fastrcnn_loss
    torch.manual_seed(1)
    M, C = 40, 7
    logits, reg = torch.randn(M, C), torch.randn(M, C * 4)
    labels = torch.randint(0, C, (M,))
    targets = torch.randn(M, 4)
    ok = torch.rand(M) > 0.3
    cls_m, box_m = fastrcnn_loss(logits, reg, torch.where(ok, labels, torch.full_like(labels, -1)),
                                 torch.where(ok[:, None], targets, torch.full_like(targets, float("nan"))), ok)
    keep = torch.where(ok)[0]
    l, r, t = labels[keep], reg[keep].reshape(-1, C, 4), targets[keep]
    pos = torch.where(l > 0)[0]
    cls_r = torch.nn.functional.cross_entropy(logits[keep], l)
    box_r = torch.nn.functional.smooth_l1_loss(r[pos, l[pos]], t[pos], beta=1 / 9, reduction="sum") / l.numel()
    assert torch.allclose(cls_m, cls_r, atol=1e-6) and torch.allclose(box_m, box_r, atol=1e-6)


def test_matcher_low_quality_restore_without_sync():
    q = torch.tensor([[0.1, 0.6, 0.2, 0.05], [0.3, 0.2, 0.25, 0.05]])
    m = ops.Matcher(0.7, 0.3, allow_low_quality_matches=True)(q)
    # column 1 is gt 0's best (0.6, between thresholds -> restored to 0); column 0 is gt 1's best (0.3)
    assert m.tolist() == [1, 0, ops.Matcher.BELOW_LOW, ops.Matcher.BELOW_LOW]


@pytest.mark.gpu
@pytest.mark.parametrize("shape,scale", [((2, 16, 50, 68), 0.25), ((1, 256, 25, 34), 0.125), ((3, 8, 7, 9), 1 / 32)])
def test_roi_align_hip_forward_backward(shape, scale):
    """HIP RoIAlign vs the plain-PyTorch fp32 reference: forward <= 5e-5 abs (sample coordinates up to
    ~60 carry ~4e-6 of fp32 rounding, the two paths associate them differently), backward <= 2e-4 abs
    (fp32 atomics change the summation order)."""
    rs = np.random.RandomState(3)
    N, C, H, W = shape
    feat = torch.tensor(rs.randn(*shape), dtype=torch.float32)
    rois = _rois(rs, 37, N, H, W, scale)
    f_ref = feat.clone().requires_grad_(True)
    ref = ops.roi_align_torch(f_ref, rois, scale, 7, 2)
    gout = torch.tensor(rs.randn(*ref.shape), dtype=torch.float32)
    ref.backward(gout)
    f_gpu = feat.cuda().requires_grad_(True)
    out = ops.roi_align(f_gpu, rois.cuda(), scale, 7, 2)
    out.backward(gout.cuda())
    assert torch.allclose(out.cpu(), ref.detach(), atol=5e-5)
    assert torch.allclose(f_gpu.grad.cpu(), f_ref.grad, atol=2e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,scale", [((2, 16, 50, 68), 0.25), ((1, 256, 25, 34), 0.125), ((2, 320, 13, 17), 1 / 16)])
def test_roi_align_channels_last_matches_planar(shape, scale):
    """NHWC kernels against the planar (NCHW) HIP kernels: forward bit-identical (same arithmetic
    order per element), backward <= 2e-4 abs (atomics), and against the fp32 torch reference."""
    rs = np.random.RandomState(11)
    N, C, H, W = shape
    feat = torch.tensor(rs.randn(*shape), dtype=torch.float32).cuda()
    rois = _rois(rs, 53, N, H, W, scale).cuda()
    f_a = feat.clone().requires_grad_(True)
    f_b = feat.clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out_a = ops.roi_align(f_a, rois, scale, 7, 2)
    out_b = ops.roi_align(f_b, rois, scale, 7, 2)
    assert out_b.is_contiguous() and torch.equal(out_a, out_b)
    gout = torch.tensor(rs.randn(*out_a.shape), dtype=torch.float32).cuda()
    out_a.backward(gout)
    out_b.backward(gout)
    assert f_b.grad.is_contiguous(memory_format=torch.channels_last)
    assert torch.allclose(f_a.grad, f_b.grad, atol=2e-4)
    f_ref = feat.cpu().requires_grad_(True)
    ref = ops.roi_align_torch(f_ref, rois.cpu(), scale, 7, 2)
    ref.backward(gout.cpu())
    assert torch.allclose(out_b.cpu(), ref.detach(), atol=5e-5)
    assert torch.allclose(f_b.grad.cpu(), f_ref.grad, atol=2e-4)


@pytest.mark.gpu
def test_multiscale_roi_align_one_launch_matches_per_level():
    """MultiScaleRoIAlign on a channels-last pyramid (one launch, device-side level map) against the
    per-level planar path, forward and backward."""
    rs = np.random.RandomState(5)
    sizes = [(64, 96), (32, 48), (16, 24), (8, 12)]
    feats = OrderedDict((str(i), torch.tensor(rs.randn(2, 32, h, w), dtype=torch.float32).cuda()) for i, (h, w) in enumerate(sizes))
    boxes = []
    for _ in range(2):
        c = rs.uniform(0, 1, (40, 2)) * [384, 256]
        wh = np.exp(rs.uniform(np.log(8), np.log(380), (40, 2)))
        b = np.concatenate([c - wh / 2, c + wh / 2], 1)
        boxes.append(torch.tensor(np.clip(b, 0, [383, 255, 383, 255]), dtype=torch.float32).cuda())
    pool = ops.MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
    res = {}
    for name, fmt in (("planar", torch.contiguous_format), ("nhwc", torch.channels_last)):
        fs = OrderedDict((k, v.clone().contiguous(memory_format=fmt).requires_grad_(True)) for k, v in feats.items())
        out = pool(fs, boxes, [(256, 384), (256, 384)])
        out.square().sum().backward()
        res[name] = (out.detach(), [v.grad for v in fs.values()])
    assert torch.equal(res["planar"][0], res["nhwc"][0])
    for a, b in zip(res["planar"][1], res["nhwc"][1]):
        a = torch.zeros_like(b) if a is None else a      # the per-level path leaves unused levels without a gradient
        assert torch.allclose(a, b, atol=5e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 63, 64, 65, 700, 1025, 3000, 4097, 9000])
def test_nms_hip_matches_torch(n):
    rs = np.random.RandomState(n)
    c = rs.uniform(0, 400, (n, 2)); s = rs.uniform(5, 80, (n, 2))
    boxes = torch.tensor(np.concatenate([c - s / 2, c + s / 2], 1), dtype=torch.float32)
    scores = torch.tensor(rs.permutation(n) / float(n), dtype=torch.float32)   # distinct scores: unique order
    want = ops.nms(boxes, scores, 0.5)
    got = ops.nms(boxes.cuda(), scores.cuda(), 0.5)
    assert got.cpu().tolist() == want.tolist()
    groups = torch.tensor(rs.randint(0, 3, n))
    assert ops.batched_nms(boxes.cuda(), scores.cuda(), groups.cuda(), 0.5).cpu().tolist() == ops.batched_nms(boxes, scores, groups, 0.5).tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [200, 5000])
def test_nms_sets_sorted_matches_per_set_nms(n):
    """B box sets in one launch pair, with a validity mask, against NMS run set by set on the
    compacted (valid-only) boxes -- the semantics RPN.filter_proposals relies on."""
    rs = np.random.RandomState(n)
    B = 3
    c = rs.uniform(0, 600, (B, n, 2)); s = rs.uniform(5, 120, (B, n, 2))
    boxes = torch.tensor(np.concatenate([c - s / 2, c + s / 2], 2), dtype=torch.float32)
    valid = torch.tensor(rs.random_sample((B, n)) > 0.2)
    valid[1] = True
    keep, count = ops.nms_sets_sorted(boxes.cuda(), valid.cuda(), 0.7)
    keep_c, count_c = ops.nms_sets_sorted(boxes, valid, 0.7)            # CPU restatement
    fake = torch.arange(n, 0, -1, dtype=torch.float32)
    for b in range(B):
        idx = torch.where(valid[b])[0]
        want = idx[ops.nms(boxes[b, idx].cuda(), fake[idx].cuda(), 0.7).cpu()]
        k = int(count[b])
        assert keep[b, :k].cpu().tolist() == want.tolist()
        assert int(keep[b, k:].abs().sum()) == 0
        assert keep_c[b, :int(count_c[b])].tolist() == want.tolist()
    k2, c2 = ops.nms_sets_sorted(boxes.cuda(), None, 0.7)
    for b in range(B):
        assert k2[b, :int(c2[b])].cpu().tolist() == ops.nms(boxes[b].cuda(), fake.cuda(), 0.7).cpu().tolist()


class _ReluLog(object):
    """Records the sign pattern of every ReLU output `backbone.bias_act` produces, in call order (the trunk's ReLUs all go through
    it, fused or not).  Two forward passes of the same trunk through different kernels differ by ~1e-6 of summation-order noise,
    which can land a pre-activation that is ~0 on the other side of its ReLU; the gradient at such a position then differs by
    its full value.  `flips(other)` counts those positions, and `tol(flips)` is the bound the comparison may use: 1e-4 of the
    Frobenius norm when no mask flipped (a dropped mask or accumulate in one stage is off by per cents), and a bound growing
    with the square root of the number of flips otherwise (one flipped position moves a gradient by ~1e-3 of its norm)."""

    def __init__(self, B):
        self.B, self.real, self.masks = B, B.bias_act, []

    def __enter__(self):
        def logged(x, bias, residual=None, relu=True):
            y = self.real(x, bias, residual, relu)
            if relu:
                self.masks.append((y.detach() > 0))
            return y
        self.B.bias_act = logged
        return self

    def __exit__(self, *exc):
        self.B.bias_act = self.real

    def flips(self, other):
        a, b = list(self.masks), list(other.masks)
        if abs(len(a) - len(b)) == 1:       # the fused stem (conv + bias + ReLU + max-pool in one node) does not go through bias_act
            (a if len(a) > len(b) else b).pop(0)
        assert len(a) == len(b) and len(a) > 0
        return sum(int((x != y).sum()) for x, y in zip(a, b))


def _tol(flips):
    return 1e-4 if flips == 0 else min(3e-2, 1e-4 + 4e-3 * flips ** 0.5)


def _detours(B, on):
    """the shape-based kernel detours of the trunk (GEMM for small-M 1x1, planar small-M 3x3): another summation order"""
    B.LINEAR_1X1 = B.NCHW_SMALL_3X3 = on



@pytest.mark.gpu
def test_fused_conv_epilogue_matches_eager_ops():
    """bias + residual + ReLU in one HIP pass (channels-last) against the eager torch ops, forward and
    backward, through a whole ResNet-50 trunk; and the scalar (C % 4 != 0) kernel on its own."""
    from detectinblur_amd.models import backbone as B
    torch.manual_seed(0)
    m = B.ResNet50Body().cuda().to(memory_format=torch.channels_last)
    for mod in m.modules():
        if isinstance(mod, B.FrozenBatchNorm2d):
            mod.weight.uniform_(0.5, 1.5); mod.bias.uniform_(-.2, .2); mod.running_mean.uniform_(-.2, .2); mod.running_var.uniform_(0.5, 1.5)
    x = torch.randn(2, 3, 96, 128, device="cuda").contiguous(memory_format=torch.channels_last)
    try:
        for detours in (False, True):       # identical convolution kernels on both sides first, then with the detours on
            _detours(B, detours)
            res, logs = {}, {}
            for fuse in (True, False):
                B.FUSE_EPILOGUE = fuse
                for p in m.parameters():
                    p.grad = None
                with _ReluLog(B) as log:
                    ys = m(x)
                sum(y.square().mean() for y in ys).backward()
                res[fuse] = ([y.detach().clone() for y in ys], m.layer2[0].conv1.weight.grad.clone(), m.conv1.weight.grad.clone())
                logs[fuse] = log
            flips = logs[True].flips(logs[False])
            assert flips < 40, (detours, flips)
            for a, b in zip(res[True][0], res[False][0]):
                assert torch.allclose(a, b, rtol=1e-4, atol=1e-4 * float(b.abs().max()))
            for k in (1, 2):     # layer2's and the stem's weight gradients: every fused epilogue and mask of the trunk lies behind them
                assert float((res[True][k] - res[False][k]).norm()) <= _tol(flips) * float(res[False][k].norm()), (detours, flips, k)
    finally:
        B.FUSE_EPILOGUE = True
        _detours(B, True)
    # scalar kernel + bias gradient
    t = torch.randn(2, 6, 5, 7, device="cuda").contiguous(memory_format=torch.channels_last)
    bias = torch.randn(6, device="cuda", requires_grad=True)
    r = torch.randn_like(t)
    a = (t.clone().requires_grad_(True), r.clone().requires_grad_(True))
    y = B.bias_act(a[0] * 1.0, bias, a[1], relu=True)
    y.sum().backward()
    want = torch.relu(t + bias.detach().reshape(1, -1, 1, 1) + r)
    assert torch.equal(y.detach(), want)
    mask = (want > 0).float()
    assert torch.equal(a[0].grad, mask) and torch.equal(a[1].grad, mask)
    assert torch.allclose(bias.grad, mask.sum(dim=(0, 2, 3)))


@pytest.mark.gpu
def test_block_entry_node_equals_the_plain_graph():
    """conv1 + skip of an identity bottleneck as one autograd node (gradient accumulate + ReLU mask in one pass, the
    producer's own mask pass skipped) against the graph autograd builds by itself, through a ResNet stage: at every block
    boundary the gradient the fused node delivers (already masked) equals the plain graph's gradient times the ReLU mask
    to 2e-5 of its maximum (the fused node runs deep 1x1 contractions as GEMMs where the plain graph calls MIOpen).  Two forward
    passes of the same stage differ by ~1e-6 for that reason and by MIOpen's kernel choice between calls,
    which can flip the sign of a pre-activation that is ~0: such positions (a handful) are excluded, and the final
    gradients are compared at the tolerance one flipped mask allows."""
    from detectinblur_amd.models import backbone as B
    torch.manual_seed(2)
    body = B.ResNet50Body().cuda().to(memory_format=torch.channels_last)
    for mod in body.modules():
        if isinstance(mod, B.FrozenBatchNorm2d):
            mod.weight.uniform_(0.5, 1.5); mod.bias.uniform_(-.2, .2); mod.running_mean.uniform_(-.2, .2); mod.running_var.uniform_(0.5, 1.5)
    x0 = torch.randn(2, 512, 20, 28, device="cuda").contiguous(memory_format=torch.channels_last)

    def run(flag):
        B.BLOCK_ENTRY = flag
        grads, outs = {}, []
        x = x0.clone().requires_grad_(True)
        y = x
        with _ReluLog(B) as log:
            for i, blk in enumerate(body.layer3):            # downsample block + 5 identity blocks
                y = blk(y)
                outs.append(y.detach().clone())
                y.register_hook(lambda g, i=i: grads.__setitem__(i, g.detach().clone()))
        for p in body.parameters():
            p.grad = None
        (y.square().mean() + (outs[2] * 0).sum()).backward()
        return grads, outs, x.grad.clone(), [p.grad.clone() for p in body.layer3.parameters()], log

    try:
        for detours in (False, True):
            _detours(B, detours)
            run(True); run(False)                           # MIOpen's find step happens here
            fused, plain = run(True), run(False)
            # every ReLU of the stage, block-internal ones included: no flipped mask -> 1e-4 of the norm (a dropped mask or
            # accumulate is off by tens of per cent), otherwise the bound one flipped position per sqrt allows
            flips = fused[4].flips(plain[4])
            assert flips < 40, (detours, flips)
            tol = _tol(flips)
            for i in range(5):                                   # outputs of blocks 0..4 feed an identity block's fused entry
                same_mask = (fused[1][i] > 0) == (plain[1][i] > 0)
                want = plain[0][i] * (plain[1][i] > 0)           # the plain graph hands over the unmasked sum
                assert float(((fused[0][i] - want) * same_mask).norm()) <= tol * float(want.norm()), (detours, i, flips)
                assert float((fused[0][i] != 0).float().mean()) < 0.8 < float((plain[0][i] != 0).float().mean())
            assert float((fused[2] - plain[2]).norm()) <= tol * float(plain[2].norm()), (detours, flips)
            for a, b in zip(fused[3], plain[3]):
                assert float((a - b).norm()) <= tol * float(b.norm()) + 1e-12, (detours, flips)
    finally:
        B.BLOCK_ENTRY = True
        _detours(B, True)


@pytest.mark.gpu
def test_entry_nodes_with_a_partly_frozen_trunk():
    """`trainable_backbone_layers=3` (the reference's default with pretrained weights): the stem and layer1 are frozen, so
    the first block of layer2 sees an input that needs no gradient (plain path), the blocks behind it take the entry
    nodes; through the whole FPN backbone, losses and every trainable gradient agree with the plain graph."""
    from detectinblur_amd.models import backbone as B
    torch.manual_seed(4)
    net = B.resnet_fpn_backbone("resnet50", False, trainable_layers=3).cuda().to(memory_format=torch.channels_last)
    for mod in net.modules():
        if isinstance(mod, B.FrozenBatchNorm2d):
            mod.weight.uniform_(0.5, 1.5); mod.bias.uniform_(-.2, .2); mod.running_mean.uniform_(-.2, .2); mod.running_var.uniform_(0.5, 1.5)
    assert not any(p.requires_grad for p in net.body.layer1.parameters()) and all(p.requires_grad for p in net.body.layer2.parameters())
    x = torch.randn(2, 3, 128, 160, device="cuda").contiguous(memory_format=torch.channels_last)
    try:
        for detours in (False, True):
            _detours(B, detours)
            res, logs = {}, {}
            for flag in (True, False, True, False):
                B.BLOCK_ENTRY = flag
                for p in net.parameters():
                    p.grad = None
                with _ReluLog(B) as log:
                    out = net(x)
                loss = sum(v.square().mean() for v in out.values())
                loss.backward()
                res[flag] = [loss.detach()] + [p.grad.clone() for p in net.parameters() if p.requires_grad]
                logs[flag] = log
            assert len(res[True]) > 50
            flips = logs[True].flips(logs[False])
            assert flips < 40, (detours, flips)
            for a, b in zip(res[True], res[False]):
                assert float((a - b).norm()) <= _tol(flips) * float(b.norm()) + 1e-12, (detours, flips)
    finally:
        B.BLOCK_ENTRY = True
        _detours(B, True)


@pytest.mark.gpu
def test_small_m_1x1_convolutions_through_the_gemm_path_equal_conv2d():
    """layer4 / top-lateral 1x1 convolutions run as F.linear on the NHWC view (hipBLASLt): same values and gradients as
    MIOpen's convolution within fp32 summation-order noise, output still channels-last."""
    from detectinblur_amd.models import backbone as B
    torch.manual_seed(1)
    conv = torch.nn.Conv2d(2048, 512, 1, bias=True).cuda()
    x = torch.randn(8, 2048, 25, 42, device="cuda").contiguous(memory_format=torch.channels_last)
    g = torch.randn(8, 512, 25, 42, device="cuda").contiguous(memory_format=torch.channels_last)
    outs = {}
    try:
        for flag in (True, False):
            B.LINEAR_1X1 = flag
            xi = x.clone().requires_grad_(True)
            conv.zero_grad()
            y = B.conv1x1(xi, conv.weight, conv.bias, conv)
            assert y.is_contiguous(memory_format=torch.channels_last) and y.shape == (8, 512, 25, 42)
            y.backward(g)
            outs[flag] = (y.detach(), xi.grad, conv.weight.grad.clone(), conv.bias.grad.clone())
    finally:
        B.LINEAR_1X1 = True
    for a, b in zip(outs[True], outs[False]):
        assert float((a - b).norm()) <= 2e-4 * float(b.norm())
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-3 * float(b.abs().max()))
    # large-M shapes stay on MIOpen: the same call (MIOpen may still switch kernels between two calls on a fresh box while
    # its search results arrive, so the comparison allows summation-order noise)
    small = torch.nn.Conv2d(64, 256, 1, bias=False).cuda()
    xs = torch.randn(2, 64, 40, 40, device="cuda").contiguous(memory_format=torch.channels_last)
    assert not B._as_gemm(xs, small)
    assert torch.allclose(B.conv1x1(xs, small.weight, None, small), torch.nn.functional.conv2d(xs, small.weight), rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_relu_mask_backward_equals_threshold_backward():
    """The mask written by the fused epilogue (one byte per 4 elements) reproduces torch's ReLU backward bit for bit, for
    channels-last and for planar incoming gradients, with and without a residual; exact zeros carry no gradient."""
    from detectinblur_amd.models import backbone as B
    torch.manual_seed(3)
    for res in (False, True):
        t = torch.randn(3, 64, 37, 53, device="cuda").contiguous(memory_format=torch.channels_last)
        t[0, :, 5, 7] = 0.0                                     # exact zeros after the bias below (bias 0 on those channels)
        bias = torch.randn(64, device="cuda")
        bias[::2] = 0.0
        r = torch.randn_like(t) if res else None
        want = torch.relu(t + bias.reshape(1, -1, 1, 1) + (r if res else 0))
        for planar_grad in (False, True):
            a = t.clone().requires_grad_(True)
            rr = r.clone().requires_grad_(True) if res else None
            y = B.bias_act(a * 1.0, bias, rr, relu=True)
            assert torch.equal(y.detach(), want)
            g = torch.randn(3, 64, 37, 53, device="cuda")
            if not planar_grad:
                g = g.contiguous(memory_format=torch.channels_last)
            y.backward(g)
            ref = torch.ops.aten.threshold_backward(g, want, 0)
            assert torch.equal(a.grad, ref)
            if res:
                assert torch.equal(rr.grad, ref)
        if not res:
            assert (want[0, ::2, 5, 7] == 0).all() and (a.grad[0, ::2, 5, 7] == 0).all()


@pytest.mark.gpu
def test_non_finite_boxes_do_not_leave_the_feature_maps():
    """A diverged step produces NaN / inf proposals; pooling and NMS must stay inside their buffers
    (the level of such a box is clamped, its samples contribute nothing or garbage, never a fault)."""
    rs = np.random.RandomState(2)
    sizes = [(64, 96), (32, 48), (16, 24), (8, 12)]
    feats = OrderedDict((str(i), torch.tensor(rs.randn(1, 16, h, w), dtype=torch.float32).cuda()
                         .contiguous(memory_format=torch.channels_last).requires_grad_(True)) for i, (h, w) in enumerate(sizes))
    nan, inf = float("nan"), float("inf")
    boxes = [torch.tensor([[10.0, 10, 60, 70], [nan, nan, nan, nan], [0, 0, inf, inf], [-inf, 5, 40, nan], [1e30, 1e30, 2e30, 2e30],
                           [5, 5, 5, 5]], dtype=torch.float32).cuda()]
    pool = ops.MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
    out = pool(feats, boxes, [(256, 384)])
    out[0].sum().backward()                       # finite RoI: finite gradient path
    torch.cuda.synchronize()
    assert torch.isfinite(out[0]).all()
    keep, count = ops.nms_sets_sorted(boxes[0][None], None, 0.5)
    torch.cuda.synchronize()
    assert 1 <= int(count[0]) <= 6


@pytest.mark.gpu
def test_fpn_topdown_merge_equals_bias_add_interpolate_add():
    """dib_fpn_topdown_merge_nhwc (lateral + bias + nearest-upsampled upper level, one in-place pass) against the three torch
    ops it replaces: the forward bit for bit (same additions in the same order, ATen's nearest index also for sizes that
    are not exact doubles), the gradients of lateral and upper level bit for bit, the bias gradient to summation order."""
    from detectinblur_amd.models import backbone as B
    F = torch.nn.functional
    torch.manual_seed(21)
    for (N, C, H, W, Ht, Wt) in ((8, 256, 50, 84, 25, 42), (2, 256, 51, 101, 26, 51), (1, 64, 7, 9, 4, 5), (3, 8, 33, 20, 11, 7)):
        lat = torch.randn(N, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
        top = torch.randn(N, C, Ht, Wt, device="cuda").contiguous(memory_format=torch.channels_last)
        bias = torch.randn(C, device="cuda")
        g = torch.randn(N, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
        res = {}
        try:
            for flag in (True, False):
                B.FUSE_TOPDOWN = flag
                l0 = lat.clone().requires_grad_(True)
                t0 = top.clone().requires_grad_(True)
                b0 = bias.clone().requires_grad_(True)
                y = B.topdown_merge(l0 * 1.0, b0, t0)        # * 1.0: a fresh tensor, as a convolution output is
                y.backward(g)
                res[flag] = (y.detach().clone(), l0.grad.clone(), t0.grad.clone(), b0.grad.clone())
        finally:
            B.FUSE_TOPDOWN = True
        want = lat + bias.reshape(1, -1, 1, 1) + F.interpolate(top, size=(H, W), mode="nearest")
        assert torch.equal(res[False][0], want)
        assert res[True][0].is_contiguous(memory_format=torch.channels_last)
        for k in range(3):
            assert torch.equal(res[True][k], res[False][k]), (N, C, H, W, k)
        assert torch.allclose(res[True][3], res[False][3], rtol=1e-4, atol=1e-3)


@pytest.mark.gpu
def test_rpn_head_fused_predictors_equal_the_module_graph():
    """RPNHead on the GPU (bias + ReLU epilogue with sign mask, both 1 x 1 predictors as one convolution) against the plain
    conv -> relu -> (cls_logits, bbox_pred) graph: outputs and every gradient within fp32 summation-order noise."""
    from detectinblur_amd.models import rpn as R
    torch.manual_seed(8)
    head = R.RPNHead(256, 3).cuda()
    for p in head.parameters():
        torch.nn.init.normal_(p, std=0.05)
    feats = [torch.randn(2, 256, h, w, device="cuda").contiguous(memory_format=torch.channels_last) for h, w in ((40, 56), (20, 28), (7, 9))]
    res = {}
    try:
        for flag in (True, False):
            R.FUSE_HEAD = flag
            fs = [f.clone().requires_grad_(True) for f in feats]
            head.zero_grad()
            logits, deltas = head(fs)
            assert [tuple(t.shape) for t in logits] == [(2, 3, 40, 56), (2, 3, 20, 28), (2, 3, 7, 9)]
            assert [tuple(t.shape) for t in deltas] == [(2, 12, 40, 56), (2, 12, 20, 28), (2, 12, 7, 9)]
            loss = sum((l * l).sum() for l in logits) + sum((d.sin()).sum() for d in deltas)
            loss.backward()
            res[flag] = ([t.detach().clone() for t in logits + deltas], [f.grad.clone() for f in fs], [p.grad.clone() for p in head.parameters()])
    finally:
        R.FUSE_HEAD = True
    for a, b in zip(res[True][0], res[False][0]):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-4)
    for group in (1, 2):
        for a, b in zip(res[True][group], res[False][group]):
            assert float((a - b).norm()) <= 2e-3 * float(b.norm()) + 1e-6          # a ReLU sign can flip on 1e-7 noise


def _reference_filter(props, obj, counts, pre, post, size_wh, thresh, min_size):
    """torchvision's RegionProposalNetwork.filter_proposals for ONE image, restated with plain torch ops: per-level top-k,
    clip, drop small boxes, one batched NMS over all levels, top post-NMS."""
    from detectinblur_amd.models import detector_ops as ops
    idx, lv, off = [], [], 0
    for l, n in enumerate(counts):
        k = min(pre, n)
        idx.append(obj[off:off + n].topk(k)[1] + off)
        lv.append(torch.full((k,), l, dtype=torch.int64, device=obj.device))
        off += n
    idx, lv = torch.cat(idx), torch.cat(lv)
    b, s = props[idx].clone(), obj[idx]
    b[:, 0::2] = b[:, 0::2].clamp(min=0, max=size_wh[0])
    b[:, 1::2] = b[:, 1::2].clamp(min=0, max=size_wh[1])
    ok = ((b[:, 2] - b[:, 0]) >= min_size) & ((b[:, 3] - b[:, 1]) >= min_size)
    b, s, lv = b[ok], s[ok], lv[ok]
    keep = ops.batched_nms(b, s, lv, thresh)[:post]
    return b[keep], s[keep]


def _canon(boxes, scores):
    """rows [score, box] ordered by score (descending) and then by the box coordinates: two proposals with the same score
    (40 seeds of random logits produce one such pair) may come out in either order."""
    rows = torch.cat([scores[:, None], boxes], dim=1).cpu().numpy()
    order = np.lexsort((rows[:, 4], rows[:, 3], rows[:, 2], rows[:, 1], -rows[:, 0]))
    return torch.from_numpy(rows[order])


def _check_rpn_filter(device, seed=31):
    from detectinblur_amd.models import rpn as R
    torch.manual_seed(seed)
    rpn = R.RegionProposalNetwork(None, None, 0.7, 0.3, 256, 0.5, dict(training=300, testing=150), dict(training=200, testing=100), 0.7)
    counts = [1200, 300, 75, 27]
    N, A = 3, sum(counts)
    for training in (True, False):
        rpn.train(training)
        cxy = torch.rand(N, A, 2, device=device) * torch.tensor([220.0, 160.0], device=device) - 10
        wh = torch.rand(N, A, 2, device=device) * 60
        wh[:, ::17] = 0.0                                              # degenerate boxes: dropped by the size test
        props = torch.cat([cxy - wh / 2, cxy + wh / 2], dim=-1)
        obj = torch.randn(N * A, 1, device=device)
        sizes = [(150, 200), (140, 190), (150, 180)]                   # (h, w)
        boxes, scores = rpn.filter_proposals(props, obj, sizes, counts)
        (pb, ok), _ = rpn.filter_proposals(props, obj, sizes, counts, padded=True)
        pre, post = (300, 200) if training else (150, 100)
        for i in range(N):
            wb, ws = _reference_filter(props[i], obj.view(N, A)[i], counts, pre, post, (float(sizes[i][1]), float(sizes[i][0])), 0.7, rpn.min_size)
            assert boxes[i].shape == wb.shape and 0 < wb.shape[0] <= post, (training, i, boxes[i].shape, wb.shape)
            assert torch.equal(scores[i], ws)                       # same scores in the same (descending) order
            assert torch.equal(_canon(boxes[i], scores[i]), _canon(wb, ws))      # boxes: up to the order among EQUAL scores
            assert int(ok[i].sum()) == wb.shape[0] and torch.equal(_canon(pb[i][ok[i]], scores[i]), _canon(wb, ws))


def test_rpn_filter_per_level_sets_equal_torchvisions_batched_nms_on_cpu():
    """RegionProposalNetwork._filter (one NMS set per image and level, survivors ranked by a top-k) against the plain
    restatement of torchvision's filter_proposals (one batched NMS per image over all levels): same boxes, same scores,
    same order, in training and in inference mode, ragged levels, degenerate boxes, unpadded and padded forms."""
    for seed in (31, 1039, 7):                                   # 1039: two proposals with exactly equal scores
        _check_rpn_filter(torch.device("cpu"), seed)


@pytest.mark.gpu
def test_rpn_filter_per_level_sets_equal_torchvisions_batched_nms_on_gpu():
    _check_rpn_filter(torch.device("cuda"))


@pytest.mark.gpu
def test_stem_bias_relu_maxpool_in_one_pass_equals_the_three_torch_ops():
    """dib_stem_pool_forward / _backward against bias add -> relu -> max_pool2d(3, 2, 1) and autograd through them: pooled
    values bit for bit (even / odd sizes, borders), the dense gradient of the convolution output bit for bit where one window
    claims a pixel and to one rounding where up to four do, ties at zero (no gradient) and positive ties (first in row-major
    window order, ATen's rule)."""
    from detectinblur_amd.models import backbone as B
    F = torch.nn.functional
    torch.manual_seed(17)
    for (N, C, H, W) in ((2, 64, 40, 56), (1, 64, 37, 51), (3, 8, 9, 12), (1, 4, 1, 1), (1, 4, 2, 5)):
        x = torch.randn(N, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
        if H > 2 and W > 2:                                       # exact ties between neighbours (positive and negative)
            x[:, :, 1::3, 1::4] = x[:, :, 0::3, 0::4][:, :, :x[:, :, 1::3, 1::4].shape[2], :x[:, :, 1::3, 1::4].shape[3]]
        bias = torch.randn(C, device="cuda")
        xa = x.clone().requires_grad_(True)
        want = F.max_pool2d(torch.relu(xa + bias.reshape(1, -1, 1, 1)), 3, stride=2, padding=1)
        g = torch.randn_like(want)
        want.backward(g)
        xb = x.clone().requires_grad_(True)
        got = B._StemPool.apply(xb * 1.0, bias)
        assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
        assert torch.equal(got, want.detach())
        got.backward(g)
        assert torch.allclose(xb.grad, xa.grad, rtol=1e-6, atol=1e-6), (N, C, H, W, float((xb.grad - xa.grad).abs().max()))
    # through the module: the body with and without the fused stem
    body = B.ResNet50Body().cuda()
    img = torch.randn(2, 3, 96, 128, device="cuda").contiguous(memory_format=torch.channels_last)
    res, logs = {}, {}
    try:
        for flag in (True, False):
            B.FUSE_STEM_POOL = flag
            body.zero_grad()
            with _ReluLog(B) as log:
                feats = body(img)
            sum((f * f).mean() for f in feats).backward()
            res[flag] = ([f.detach().clone() for f in feats], body.conv1.weight.grad.clone())
            logs[flag] = log
    finally:
        B.FUSE_STEM_POOL = True
    for a, b in zip(res[True][0], res[False][0]):
        assert float((a - b).norm()) <= 1e-4 * float(b.norm())            # 50 random-init layers behind the stem
    # conv1's weight gradient has crossed ~50 random-init layers twice: tight unless a ReLU sign flipped on the way
    flips = logs[True].flips(logs[False])
    assert float((res[True][1] - res[False][1]).norm()) <= _tol(flips) * float(res[False][1].norm()), flips
    try:                                                                   # the stem itself: bit for bit
        with torch.no_grad():
            fused = B.stem(img, body.conv1, body.bn1)
            B.FUSE_STEM_POOL = False
            plain = B.stem(img, body.conv1, body.bn1)
    finally:
        B.FUSE_STEM_POOL = True
    assert torch.equal(fused, plain)


@pytest.mark.gpu
def test_fold_all_equals_the_per_convolution_folds_bit_for_bit():
    """All frozen batch-norm folds of the ResNet body in one launch per 32 (dib_fold_bn_multi) against the tensor expressions
    they replace -- scale = w_bn * rsqrt(var + eps), shift = b_bn - mean * scale, w * scale[co] -- forward (folded weights and
    shifts of all 53 pairs, contiguous and channels-last weights, the 7x7 stem with 147 elements per channel) and backward
    (dw = g * scale[co]): equal bit for bit.  Frozen pairs are left to the inference cache; the parked results are gone after
    the forward pass."""
    from detectinblur_amd.models import backbone as B
    torch.manual_seed(3)
    body = B.ResNet50Body().cuda()
    body.layer2.to(memory_format=torch.channels_last)
    for p in body.layer1.parameters():
        p.requires_grad_(False)
    for mod in body.modules():
        if isinstance(mod, B.FrozenBatchNorm2d):
            mod.weight.uniform_(0.5, 1.5); mod.bias.uniform_(-.2, .2); mod.running_mean.uniform_(-.2, .2); mod.running_var.uniform_(0.5, 1.5)
    x = torch.randn(1, 3, 64, 64, device="cuda")
    pairs = B._begin_step_folds(body, x)
    try:
        assert pairs is not None and len(pairs) == 53 - 10                          # layer1 (3 blocks x 3 + downsample) is frozen
        parked = [conv.__dict__["_dib_step_fold"] for conv, _ in pairs]
        g = [torch.randn_like(w) for w, _ in parked]
        torch.autograd.backward([w for w, _ in parked], g)
        got_dw = [conv.weight.grad.clone() for conv, _ in pairs]
        for (conv, bn), (wf, shift) in zip(pairs, parked):
            conv.weight.grad = None
            scale, want_shift = bn.affine()
            want = conv.weight * scale.reshape(-1, 1, 1, 1)
            assert torch.equal(wf, want) and wf.stride() == want.stride(), conv
            assert torch.equal(shift, want_shift) and not shift.requires_grad
        for (conv, bn), gi in zip(pairs, g):
            (conv.weight * bn.affine()[0].reshape(-1, 1, 1, 1)).backward(gi)
        for (conv, _), dw in zip(pairs, got_dw):
            assert torch.equal(conv.weight.grad, dw)
        assert "_dib_step_fold" not in body.layer1[0].conv1.__dict__
    finally:
        B._end_step_folds(pairs)
    assert all("_dib_step_fold" not in conv.__dict__ for conv, _ in pairs)
    # through the module: the parked folds are used and cleaned up, the result equals the per-convolution path
    for p in body.parameters():
        p.grad = None
    img = torch.randn(2, 3, 96, 128, device="cuda").contiguous(memory_format=torch.channels_last)
    res = {}
    try:
        for flag in (True, False):
            B.FOLD_ALL = flag
            body.zero_grad()
            feats = body(img)
            assert all("_dib_step_fold" not in conv.__dict__ for conv, _ in pairs)
            sum((f * f).mean() for f in feats).backward()
            res[flag] = ([f.detach().clone() for f in feats], body.layer4[2].conv3.weight.grad.clone())
    finally:
        B.FOLD_ALL = True
    for a, b in zip(res[True][0], res[False][0]):
        assert float((a - b).norm()) <= 1e-5 * float(b.norm())                      # same folded weights; MIOpen's own run-to-run noise
    assert float((res[True][1] - res[False][1]).norm()) <= 1e-4 * float(res[False][1].norm())



def test_cat_boxes_offsets_and_alignment():
    from detectinblur_amd.models import detector_ops as ops
    a, b, c = torch.rand(3, 4), torch.zeros((0, 4)), torch.rand(2, 4)
    cat, offs = ops.cat_boxes([a, b, c])
    assert offs == [0, 3, 3, 5] and torch.equal(cat, torch.cat((a, c)))
    assert ops.cat_boxes([b, b]) == (None, [0, 0, 0])
    one, offs = ops.cat_boxes([b, c])
    assert offs == [0, 0, 2] and torch.equal(one, c)
    assert not ops.hip_boxes_ok(a)                                # CPU tensors keep the tensor expressions
