"""dib_topk_levels (csrc/dib_topk.hip) -- the RPN's per-level top-k + gather + clip + size test in one launch -- against
torch.sort / torch.topk and against the tensor form of the same selection in models/rpn.py (torchvision's filter_proposals as the
reference's models/faster_rcnn.py:198-207 configures it).  Exact: scores, indices (ties by ascending index), boxes, flags."""
import pytest
import torch

from detectinblur_amd.models import detector_ops as ops

pytestmark = pytest.mark.gpu


def _want(values, counts, ks, K):
    """stable descending sort per level = descending score, ascending index among equals, NaN first"""
    N = values.shape[0]
    s = torch.full((N, len(counts), K), float("-inf"))
    ix = torch.zeros((N, len(counts), K), dtype=torch.int64)
    off = 0
    for l, (c, k) in enumerate(zip(counts, ks)):
        k = min(k, c)
        v, i = torch.sort(values[:, off:off + c].cpu(), dim=1, descending=True, stable=True)
        s[:, l, :k], ix[:, l, :k] = v[:, :k], i[:, :k]
        off += c
    return s, ix


def _same(a, b):
    return torch.equal(a.isnan(), b.isnan()) and torch.equal(a.nan_to_num(0.0, 1e30, -1e30), b.nan_to_num(0.0, 1e30, -1e30))


@pytest.mark.parametrize("counts,ks,K", [
    ([201600, 50400, 12600, 3150, 819], [2000, 2000, 2000, 2000, 819], 2000),        # the training shapes at 800 x 1333
    ([201600, 50400, 12600, 3150, 819], [1000, 1000, 1000, 1000, 819], 1000),        # evaluation
    ([5000], [2048], 2048), ([7], [7], 16), ([7], [3], 3), ([1], [1], 1), ([1025, 64, 2], [1, 64, 5], 64), ([10000], [1000], 1000)])
def test_distinct_scores_equal_torch_topk(counts, ks, K):
    g = torch.Generator().manual_seed(sum(counts))
    v = (torch.randn(3, sum(counts), generator=g) * 3).cuda()
    s, ix, _, _ = ops.topk_levels_hip(v, counts, ks, K, want_index=True)
    ws, wi = _want(v, counts, ks, K)
    assert torch.equal(s.cpu(), ws) and torch.equal(ix.cpu(), wi)
    off = 0
    for l, (c, k) in enumerate(zip(counts, ks)):                                       # and torch.topk itself, where it is defined (no ties)
        tv, ti = v[:, off:off + c].topk(min(k, c), dim=1)
        assert torch.equal(s[:, l, :min(k, c)], tv) and torch.equal(ix[:, l, :min(k, c)], ti)
        off += c


def test_ties_nan_and_infinities():
    """Quantised scores (every value many times, ties across the k-th place), NaN (ranked first, as torch does), +-inf, a constant
    row (every element ties: the index-ordered path), all -inf."""
    g = torch.Generator().manual_seed(5)
    v = torch.round(torch.randn(6, 30000, generator=g) * 4) / 4
    v[0, 17] = float("nan"); v[0, 29999] = float("nan"); v[0, 5] = float("inf"); v[0, 6] = float("-inf")
    v[1] = 0.25
    v[2] = float("-inf")
    v[3, :] = -0.0; v[3, ::2] = 0.0                                                    # signed zeros compare equal
    v[4, 100:] = float("-inf")
    v = v.cuda()
    for counts, ks, K in (([30000], [2000], 2000), ([20000, 10000], [1500, 700], 1500), ([30000], [1], 1)):
        s, ix, _, _ = ops.topk_levels_hip(v, counts, ks, K, want_index=True)
        ws, wi = _want(v, counts, ks, K)
        assert _same(s.cpu(), ws) and torch.equal(ix.cpu(), wi)
        assert torch.equal(torch.signbit(s.cpu()[3]), torch.signbit(ws[3]))             # signed zeros tie (index order), their sign survives


def test_boxes_clip_and_flags_equal_the_tensor_form():
    from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
    torch.manual_seed(0)
    rpn = fasterrcnn_resnet50_fpn(pretrained=False, pretrained_backbone=False, num_classes=91).rpn
    g = torch.Generator().manual_seed(9)
    counts = [12000, 3000, 750, 190, 48]
    A, N = sum(counts), 3
    obj = torch.randn(N, A, generator=g).cuda()
    xy = torch.rand(N, A, 2, generator=g) * torch.tensor([1400.0, 900.0]) - 60.0
    wh = torch.rand(N, A, 2, generator=g) * torch.tensor([300.0, 300.0])
    wh[:, ::7] = 0.0005                                                                 # slivers: below min_size after clipping
    props = torch.cat((xy, xy + wh), dim=2).cuda()
    props[1, 5] = float("nan")
    obj[1, 5] = 50.0                                                                    # a winner with NaN coordinates
    sizes = torch.tensor([[1333.0, 800.0], [1200.0, 800.0], [640.0, 480.0]]).cuda()
    for training in (True, False):
        rpn.train(training)
        K = max(min(rpn._n(rpn._pre), n) for n in counts)
        ws, wb, wv = rpn._select_levels(props, obj, sizes, counts, K)
        s, _, b, v = ops.topk_levels_hip(obj, counts, [min(rpn._n(rpn._pre), n) for n in counts], K, props, sizes, rpn.min_size)
        assert torch.equal(s, ws) and torch.equal(v, wv) and v.dtype == torch.bool
        assert _same(b.cpu(), wb.cpu())
        assert 0 < int(v.sum()) < v.numel() and not bool(v[1, 0, 0])
    # the whole filter, kernels against tensor expressions
    out = {}
    for flag in (True, False):
        ops.HIP_BOXES = flag
        try:
            out[flag] = rpn._filter(props, obj.reshape(-1, 1), sizes, counts)
        finally:
            ops.HIP_BOXES = True
    for a, b in zip(out[True], out[False]):
        n = out[True][2]
        live = torch.arange(a.shape[1], device=a.device)[None, :] < n[:, None] if a.dim() > 1 else None
        if live is None:
            assert torch.equal(a, b)
        else:
            assert _same(a[live].cpu(), b[live].cpu())                                 # rows past the count are padding on both sides


def test_split_rows_give_the_same_winners():
    """Long levels selected in pieces and merged: identical scores, boxes and flags, ties included (quantised scores)."""
    g = torch.Generator().manual_seed(13)
    counts = [201600, 50400, 12600, 3150, 819]
    A, N = sum(counts), 2
    obj = (torch.round(torch.randn(N, A, generator=g) * 64) / 64).cuda()
    xy = torch.rand(N, A, 2, generator=g) * torch.tensor([1300.0, 780.0])
    props = torch.cat((xy, xy + torch.rand(N, A, 2, generator=g) * 200), dim=2).cuda()
    sizes = torch.tensor([[1333.0, 800.0], [1000.0, 800.0]]).cuda()
    for K in (1000, 2000):
        ks = [min(K, c) for c in counts]
        s, _, b, v = ops.topk_levels_hip(obj, counts, ks, K, props, sizes, 1e-3)
        s2, b2, v2 = ops.topk_levels_split_hip(obj, counts, ks, K, props, sizes, 1e-3)
        assert torch.equal(s, s2) and torch.equal(b, b2) and torch.equal(v, v2)
    assert ops.TOPK_PIECE < counts[0] and -(-counts[0] // ops.TOPK_PIECE) + -(-counts[1] // ops.TOPK_PIECE) + 3 <= ops.TOPK_MAX_LEVELS


def test_limits_are_refused():
    from detectinblur_amd import _lib
    v = torch.zeros(1, 100).cuda()
    with pytest.raises(_lib.DibError):
        ops.topk_levels_hip(v, [100], [10], 4096)
    with pytest.raises(_lib.DibError):
        ops.topk_levels_hip(v, [3] * 33, [1] * 33, 4)
    with pytest.raises(_lib.DibError):
        ops.topk_levels_hip(v, [200], [10], 16)
