"""JPEG round trip against outputs of the reference's models/jpeg/DiffJPEG
(tests/golden/jpeg.npz, generated on CPU by oracle/gen_goldens.py)."""
import numpy as np
import pytest
import torch

import gen_goldens as GG
from detectinblur_amd.models.jpeg import DiffJPEG, quality_to_factor


@pytest.mark.parametrize("q", GG.JPEG_QUALITIES)
def test_round_trip_matches_reference_on_cpu(golden, q):
    x = GG.jpeg_input()
    m = DiffJPEG(height=100, width=100, differentiable=False, quality=10)
    m.setRes(32, 48)
    m.setQuality(q)
    with torch.no_grad():
        out = m(x)
    want = golden.jpeg["jpeg_q%d" % q]
    assert out.shape == want.shape and out.dtype == torch.float32
    assert np.array_equal(out.numpy(), want)       # same torch ops, same order, same machine


def test_quality_factor_and_artifact_helper():
    assert quality_to_factor(10) == 5.0 and abs(quality_to_factor(75) - 0.5001) < 1e-12
    from detectinblur_amd.transforms import add_jpeg_artifact_to_image
    m = DiffJPEG(height=100, width=100, differentiable=False, quality=10)
    img = torch.rand(3, 37, 50).half()
    out = add_jpeg_artifact_to_image(img, m, 60)
    assert out.shape == img.shape and out.dtype == torch.float16
    assert float((out.float() - img.float()).abs().mean()) < 0.2       # recognisably the same picture


@pytest.mark.gpu
def test_round_trip_on_gpu_close_to_reference(golden):
    """On the GPU the tensordot kernels sum in another order: a coefficient can land on the other side of
    .5 and move one quantisation step.  Stated tolerance: mean abs 2e-3, max one luminance step / 255."""
    x = GG.jpeg_input().cuda()
    m = DiffJPEG(height=100, width=100, quality=10).cuda()
    m.setRes(32, 48)
    for q in GG.JPEG_QUALITIES:
        m.setQuality(q)
        with torch.no_grad():
            d = (m(x).cpu().numpy() - golden.jpeg["jpeg_q%d" % q])
        assert np.abs(d).mean() <= 2e-3 and np.abs(d).max() <= 121 * quality_to_factor(q) / 255 + 1e-3
