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
