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


@pytest.mark.gpu
def test_fused_jpeg_kernel_close_to_the_module_path():
    """csrc/dib_jpeg.hip (pad + colour + 4:2:0 + DCT + quantise + inverse + crop in one launch) against the module-by-module
    path of the same helper, run on the CPU (the reference's arithmetic order): the tolerance the torch GPU path has --
    mean abs 2e-3, max one luminance quantisation step -- for ragged sizes, a multiple of 16 (a full extra macroblock of
    padding, reference transforms.py:471-472), Half and float inputs, low and high quality."""
    from detectinblur_amd import transforms as T
    m_cpu = DiffJPEG(height=100, width=100, differentiable=False, quality=10)
    m_gpu = DiffJPEG(height=100, width=100, differentiable=False, quality=10).cuda()
    rs = np.random.RandomState(3)
    for k, (shape, q) in enumerate((((3, 37, 50), 60), ((3, 64, 64), 25), ((3, 100, 131), 88.5), ((3, 17, 200), 10), ((3, 203, 160), 45))):
        # smooth + texture, so that high frequencies quantise to something
        yy, xx = np.meshgrid(np.linspace(0, 1, shape[1]), np.linspace(0, 1, shape[2]), indexing="ij")
        img = np.stack([0.5 + 0.4 * np.sin(6 * xx + c) * np.cos(4 * yy) for c in range(3)]) + rs.uniform(-0.08, 0.08, shape)
        x = torch.from_numpy(np.clip(img, 0, 1).astype(np.float32))
        x = x.half() if k % 2 == 0 else x
        want = T.add_jpeg_artifact_to_image(x, m_cpu, q)
        got = T.add_jpeg_artifact_to_image(x.cuda(), m_gpu, q)
        assert got.device.type == "cpu" and got.dtype == torch.float16 and got.shape == want.shape
        d = (got.float() - want.float()).abs()
        assert float(d.mean()) <= 2e-3 and float(d.max()) <= 121 * quality_to_factor(q) / 255 + 2e-3, (shape, q, float(d.mean()), float(d.max()))
        try:                                           # and against the module path on the GPU
            T.FUSE_JPEG = False
            mod = T.add_jpeg_artifact_to_image(x.cuda(), m_gpu, q)
        finally:
            T.FUSE_JPEG = True
        d2 = (got.float() - mod.float()).abs()
        assert float(d2.mean()) <= 2e-3 and float(d2.max()) <= 121 * quality_to_factor(q) / 255 + 2e-3
        assert float((got.float() - x.float()).abs().mean()) < 0.2
