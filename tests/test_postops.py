"""the post-blur corruption chain of `manual_blur` (reference models/blur_functions.py:72-81) against
the reference's own output (tests/golden/postops.npz: the reference run on the CPU behind an identity PSF, numpy and
torch seeded per case): Gaussian noise + clamp, and the nearest-neighbour "block" down/up-sampling, including the draw
order on numpy's global stream.  On the GPU the block path must equal the CPU result exactly (index arithmetic only);
the noise draws come from the device generator there, so its statistics are checked instead."""
import math
import os

import numpy as np
import pytest
import torch

import gen_goldens as GG
from detectinblur_amd.models.blur_functions import _post_ops

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "postops.npz"))


def test_noise_and_block_match_the_reference_on_cpu():
    x, _ = GG.postop_input()
    for seed in GG.POSTOP_SEEDS:
        np.random.seed(seed); torch.manual_seed(seed)
        got = _post_ops(x.clone(), True, 0.01, False, False, None)
        assert np.array_equal(got.numpy(), G["noise_%d" % seed]), seed
        np.random.seed(seed); torch.manual_seed(seed)
        got = _post_ops(x.clone(), False, 0.001, True, False, None)
        assert np.array_equal(got.numpy(), G["block_%d" % seed]), seed
        np.random.seed(seed); torch.manual_seed(seed)
        got = _post_ops(x.clone(), True, 0.004, True, False, None)
        assert np.array_equal(got.numpy(), G["both_%d" % seed]), seed
        assert np.random.uniform() == G["rng_after_%d" % seed][0]          # same number of numpy draws, same order
    changed = [not np.array_equal(G["block_%d" % s], x.numpy()) for s in GG.POSTOP_SEEDS]
    assert any(changed) and not all(changed)                               # both arms of the coin flip are pinned


@pytest.mark.gpu
def test_block_on_gpu_equals_reference_and_noise_has_the_drawn_variance():
    x, _ = GG.postop_input()
    for seed in GG.POSTOP_SEEDS:
        np.random.seed(seed)
        got = _post_ops(x.clone().cuda(), False, 0.001, True, False, None)
        assert np.array_equal(got.cpu().numpy(), G["block_%d" % seed]), seed
    # noise: same numpy draw for the variance, device draws for the field
    big = torch.full((3, 400, 500), 0.5, device="cuda")
    np.random.seed(4)
    var = np.random.RandomState(4).uniform(0.00000001, 0.01)
    out = _post_ops(big.clone(), True, 0.01, False, False, None)
    d = (out - big).float()
    assert abs(float(d.mean())) < 4 * math.sqrt(var / d.numel()) + 1e-6
    assert abs(float(d.var()) / var - 1) < 0.02
    assert float(out.min()) >= 0 and float(out.max()) <= 1
    half = _post_ops(torch.rand(3, 64, 64, device="cuda").half(), True, 0.001, True, False, None)
    assert half.dtype == torch.float16 and tuple(half.shape) == (3, 64, 64)


@pytest.mark.gpu
def test_fused_block_pass_equals_torchs_two_interpolates_bit_for_bit():
    """csrc/dib_postops.hip composes the two nearest-neighbour maps by index arithmetic: the same result as the stock
    `interpolate(scale_factor=s)` + `interpolate(size=original)` on the device, over random sizes and scale factors,
    fp16 / fp32, 3-D and 2-D images."""
    from detectinblur_amd import blur_ops
    rs = np.random.RandomState(11)
    F = torch.nn.functional
    for k in range(60):
        C, H, W = (1, 3)[k % 2], int(rs.randint(5, 420)), int(rs.randint(5, 520))
        s = float(rs.uniform(0.6, 1.0)) if k % 7 else (0.6, 1.0 - 1e-9, 0.75, 0.9999999)[k % 4]
        dt = (torch.float16, torch.float32)[k % 3 == 0]
        x = torch.rand(C, H, W, device="cuda").to(dt)
        img = x[0] if (C == 1 and k % 4 == 1) else x
        want = F.interpolate(F.interpolate(img.reshape(1, -1, H, W), scale_factor=(s, s), mode="nearest"), size=(H, W), mode="nearest")
        got = blur_ops.post_ops(img, None, s)
        assert got.shape == img.shape and got.dtype == dt
        assert torch.equal(got.reshape(-1, H, W), want[0]), (k, C, H, W, s)
    big = torch.rand(3, 800, 1333, device="cuda").half()
    s = 0.8123
    want = F.interpolate(F.interpolate(big[None], scale_factor=(s, s), mode="nearest"), size=(800, 1333), mode="nearest")[0]
    assert torch.equal(blur_ops.post_ops(big, None, s), want)


@pytest.mark.gpu
def test_fused_noise_is_reproducible_travels_with_the_blocks_and_has_the_right_moments():
    from detectinblur_amd import blur_ops
    F = torch.nn.functional
    x = (torch.rand(3, 300, 420, device="cuda") * 0.6 + 0.2).half()
    torch.manual_seed(5)
    a = blur_ops.post_ops(x, 0.004, None)
    torch.manual_seed(5)
    b = blur_ops.post_ops(x, 0.004, None)
    c = blur_ops.post_ops(x, 0.004, None)
    assert torch.equal(a, b) and not torch.equal(a, c)                 # keyed by torch's host generator
    d = (a.float() - x.float())
    assert abs(float(d.mean())) < 4 * math.sqrt(0.004 / d.numel()) + 2e-5
    assert abs(float(d.var()) / 0.004 - 1) < 0.02
    z = (d / math.sqrt(0.004))
    assert abs(float((z ** 3).mean())) < 0.05 and abs(float((z ** 4).mean()) - 3.0) < 0.1      # skewness 0, kurtosis 3
    # no correlation between neighbours or channels (a counter-based field)
    assert abs(float((z[:, :, 1:] * z[:, :, :-1]).mean())) < 0.02 and abs(float((z[0] * z[1]).mean())) < 0.02
    # noise first, then blocks -- as the reference orders them: the fused pass == blocks of the noisy image
    torch.manual_seed(9)
    both = blur_ops.post_ops(x, 0.004, 0.7)
    torch.manual_seed(9)
    noisy = blur_ops.post_ops(x, 0.004, None)
    want = F.interpolate(F.interpolate(noisy[None], scale_factor=(0.7, 0.7), mode="nearest"), size=(300, 420), mode="nearest")[0]
    assert torch.equal(both, want)
    # clamp, fp32 images, and the rounding steps of torch's Half expression (sum of two Halves: results on the Half grid)
    edge = torch.cat([torch.zeros(1, 64, 64), torch.ones(1, 64, 64)]).cuda()
    out = blur_ops.post_ops(edge, 0.01, None)
    assert float(out.min()) == 0.0 and float(out.max()) == 1.0 and out.dtype == torch.float32
    assert 0.4 < float((out[0] > 0).float().mean()) < 0.6


@pytest.mark.gpu
def test_post_ops_through_the_drop_in_keep_the_references_numpy_draw_order():
    """`_post_ops` on the GPU (fused kernel) and with FUSE_POST_OPS off (stock torch ops) consume numpy's global stream
    identically -- variance, coin flip, scale factor, in the reference's order -- and agree on the block arm exactly."""
    from detectinblur_amd.models import blur_functions as BF
    x = torch.rand(3, 120, 160, device="cuda").half()
    for seed in range(8):
        outs, after = {}, {}
        try:
            for flag in (True, False):
                BF.FUSE_POST_OPS = flag
                np.random.seed(seed); torch.manual_seed(seed)
                outs[flag] = BF._post_ops(x.clone(), False, 0.001, True, False, None)
                after[flag] = np.random.uniform()
                np.random.seed(seed)
                BF._post_ops(x.clone(), True, 0.003, True, False, None)
                after[flag] = (after[flag], np.random.uniform())
        finally:
            BF.FUSE_POST_OPS = True
        assert torch.equal(outs[True], outs[False]) and after[True] == after[False]
