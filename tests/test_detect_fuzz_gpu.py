"""Randomised sweep of the detector-side kernels (csrc/dib_detect.hip, csrc/dib_topk.hip) against the tensor expressions they replace:
box matching / encoding / decoding over random image counts, ground-truth counts (0 included), candidate counts, thresholds, shared
and per-image candidates, duplicated and degenerate boxes; sorted top-k over random row lengths, level splits, k and score
quantisations (ties).  Bit for bit, as in tests/test_detect_gpu.py and tests/test_topk_gpu.py; DIB_FUZZ_CASES scales the sweep."""
import os

import pytest
import torch

from detectinblur_amd.models import detector_ops as ops

pytestmark = pytest.mark.gpu
CASES = int(os.environ.get("DIB_FUZZ_CASES", "12"))


def _rand_boxes(g, n, W, H, degenerate=0.0):
    xy = torch.rand(n, 2, generator=g) * torch.tensor([W, H])
    wh = torch.rand(n, 2, generator=g) ** 2 * torch.tensor([W / 2.0, H / 2.0])
    if degenerate:
        wh[torch.rand(n, generator=g) < degenerate] = 0.0
    return torch.cat((xy, xy + wh), dim=1)


@pytest.mark.parametrize("seed", range(CASES))
def test_match_encode_decode_random_cases(seed):
    g = torch.Generator().manual_seed(1000 + seed)
    N = int(torch.randint(1, 9, (1,), generator=g))
    M = int(torch.randint(1, 6000, (1,), generator=g))
    W, H = 1333.0, 800.0
    per_image = bool(seed % 2)
    gts = []
    for _ in range(N):
        n = int(torch.randint(0, 41, (1,), generator=g)) if seed % 5 else 0
        b = _rand_boxes(g, n, W, H, degenerate=0.1 if seed % 3 == 0 else 0.0)
        if n > 3 and seed % 4 == 0:
            b[n - 1] = b[0]                                            # duplicates: argmax ties
        gts.append(b.cuda())
    cand = (torch.stack([_rand_boxes(g, M, W, H, 0.05) for _ in range(N)]) if per_image else _rand_boxes(g, M, W, H, 0.05)).cuda()
    for i, b in enumerate(gts):                                        # exact hits
        if b.shape[0] and M > 2:
            (cand[i] if per_image else cand)[1] = b[0]
    lo = float(torch.rand(1, generator=g)) * 0.5
    hi = lo + float(torch.rand(1, generator=g)) * 0.4
    gt_cat, offs = ops.cat_boxes(gts)
    for allow in (True, False):
        matcher = ops.Matcher(hi, lo, allow_low_quality_matches=allow)
        got = ops.match_boxes_hip(matcher, gt_cat, offs, cand, shared=not per_image)
        for i, b in enumerate(gts):
            c = cand[i] if per_image else cand
            want = matcher(ops.box_iou(b, c)) if b.shape[0] else torch.full((M,), -1, dtype=torch.int64, device="cuda")
            assert torch.equal(got[i], want), (seed, allow, i, int((got[i] != want).sum()))
    coder = ops.BoxCoder((10.0, 10.0, 5.0, 5.0) if seed % 2 else (1.0, 1.0, 1.0, 1.0))
    match = torch.stack([torch.randint(-2, max(b.shape[0], 1), (M,), generator=g) for b in gts]).cuda()
    tg, mb = ops.encode_matched_hip(coder, gt_cat, offs, match, cand, shared=not per_image, want_targets=True, want_matched=True)
    for i, b in enumerate(gts):
        c = cand[i] if per_image else cand
        ref = b[match[i].clamp(min=0)] if b.shape[0] else torch.zeros_like(c)
        want = coder.encode(ref, c)
        assert torch.equal(mb[i], ref)
        assert torch.equal(tg[i].isnan(), want.isnan()) and torch.equal(tg[i].nan_to_num(1.0, 2.0, 3.0), want.nan_to_num(1.0, 2.0, 3.0)), (seed, i)
    if not per_image:
        deltas = (torch.randn(N * M, 4, generator=g) * torch.tensor([0.5, 0.5, 3.0, 3.0])).cuda()
        got = ops.decode_boxes_hip(coder, deltas, cand)
        want = coder.decode(deltas, torch.cat([cand] * N)).reshape(-1, 4)
        assert torch.equal(got.isnan(), want.isnan()) and torch.equal(got.nan_to_num(0.0, 1.0, 2.0), want.nan_to_num(0.0, 1.0, 2.0)), seed


@pytest.mark.parametrize("seed", range(CASES))
def test_topk_random_rows_levels_and_ties(seed):
    g = torch.Generator().manual_seed(2000 + seed)
    N = int(torch.randint(1, 5, (1,), generator=g))
    L = int(torch.randint(1, 7, (1,), generator=g))
    counts = [int(torch.randint(1, 70000 if l == 0 else 4000, (1,), generator=g)) for l in range(L)]
    K = int(torch.randint(1, 2049, (1,), generator=g))
    ks = [min(K, int(torch.randint(0, K + 1, (1,), generator=g))) for _ in counts]
    v = torch.randn(N, sum(counts), generator=g)
    q = (0, 4, 64, 1024)[seed % 4]
    if q:
        v = torch.round(v * q) / q                                       # ties, also across the k-th place
    if seed % 6 == 0:
        v[0, : counts[0] // 2] = float("-inf")
    v = v.cuda()
    s, ix, _, _ = ops.topk_levels_hip(v, counts, ks, K, want_index=True)
    off = 0
    for l, (c, k) in enumerate(zip(counts, ks)):
        k = min(k, c)
        ws, wi = torch.sort(v[:, off:off + c], dim=1, descending=True, stable=True)
        assert torch.equal(s[:, l, :k], ws[:, :k]) and torch.equal(ix[:, l, :k], wi[:, :k]), (seed, l)
        assert (s[:, l, k:] == float("-inf")).all()
        off += c
    boxes = torch.rand(N, sum(counts), 4, generator=g).cuda() * 100
    s1, _, b1, v1 = ops.topk_levels_hip(v, counts, ks, K, boxes, None, 0.0)
    s2, b2, v2 = ops.topk_levels_split_hip(v, counts, ks, K, boxes, None, 0.0)
    assert torch.equal(s1, s2) and torch.equal(b1, b2) and torch.equal(v1, v2)
