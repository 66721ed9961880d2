"""GPU parity: box growth / clamping and the PSF rasteriser against goldens from the reference."""
import numpy as np
import pytest
import torch

import dib_oracle as O
import golden_inputs as GI

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("case", GI.box_cases(), ids=lambda c: c["name"])
def test_expand_targets_golden(golden, case):
    from detectinblur_amd import utils
    boxes, psf, shape = GI.make_box_case(case)
    tgt = [{"boxes": _dev(boxes)}]
    ret = utils.expand_targets(tgt, [{"blurring": True}], [_dev(psf)], [torch.zeros(shape, dtype=torch.float16, device="cuda")])
    assert ret is tgt
    assert np.array_equal(tgt[0]["boxes"].cpu().numpy().view(np.uint32), golden.boxes["boxes_" + case["name"]])


def test_expand_targets_skips_unblurred_and_refuses_256():
    from detectinblur_amd import utils
    b = _dev(np.array([[1, 2, 3, 4]], np.float32))
    tgt = [{"boxes": b.clone()}]
    utils.expand_targets(tgt, [{"blurring": False}], [torch.zeros(1, device="cuda")], [torch.zeros(3, 10, 10, device="cuda")])
    assert torch.equal(tgt[0]["boxes"], b)
    with pytest.raises(Exception, match="not 128 wide"):
        utils.expand_targets(tgt, [{"blurring": True}], [torch.ones(256, 256, device="cuda").half()],
                             [torch.zeros(3, 10, 10, device="cuda")])


def test_fix_bounding_box_squeeze_golden(golden):
    from detectinblur_amd import utils
    t = {"boxes": _dev(GI.make_squeeze_boxes())}
    assert utils.fix_bounding_box_squeeze(t, (3, 100, 150)) is t
    assert np.array_equal(t["boxes"].cpu().numpy().view(np.uint32), golden.boxes["boxes_squeeze"])


@pytest.mark.parametrize("param", GI.PARAMS)
def test_rasterizer_bit_exact_all_fractions(golden, param):
    """One batched call for the six exposure fractions of a blur type: float64 PSF after fit +
    centre + crop is bit-identical to the reference; so is the Half conversion."""
    from detectinblur_amd import blur_ops
    trajs, fracs = [], []
    for fi, frac in enumerate(GI.FRACTIONS):
        np.random.seed(GI.psf_seed(param, fi))
        O.trajectory(256, 2000, 96, param)
        x_re, x_im, _, _ = O.trajectory(256, 2000, 96, param)
        trajs.append(x_re + 1j * x_im)
        fracs.append(frac)
    p64, p16 = blur_ops.rasterize_psfs(np.stack(trajs), fracs, canvas=256, center=True)
    raw64, _ = blur_ops.rasterize_psfs(np.stack(trajs), fracs, canvas=256, center=False, want16=False)
    cen64, _ = blur_ops.rasterize_psfs(np.stack(trajs), fracs, canvas=256, center=True, out_n=256, want16=False)
    for fi in range(len(GI.FRACTIONS)):
        assert np.array_equal(raw64[fi].cpu().numpy(), GI.golden_psf(param, fi, "raw"))
        assert np.array_equal(cen64[fi].cpu().numpy(), GI.golden_psf(param, fi, "cen"))
        assert np.array_equal(p64[fi].cpu().numpy(), GI.golden_psf(param, fi, "crop"))
        assert np.array_equal(p16[fi].cpu().numpy().view(np.uint16), GI.golden_psf(param, fi, "half").view(np.uint16))


def test_rasterizer_small_canvas_vs_oracle():
    from detectinblur_amd import blur_ops
    np.random.seed(3)
    x_re, x_im, _, _ = O.trajectory(128, 700, 50, 0.005)
    for frac in (0.0004, 0.3, 1.0):
        want = O.psf_rasterize(x_re, x_im, [frac], 128)[0]
        got, _ = blur_ops.rasterize_psfs((x_re + 1j * x_im)[None], [frac], canvas=128, center=False, want16=False)
        assert np.array_equal(got[0].cpu().numpy(), want)
        gotc, _ = blur_ops.rasterize_psfs((x_re + 1j * x_im)[None], [frac], canvas=128, center=True, want16=False)
        assert np.array_equal(gotc[0].cpu().numpy(), O.psf_center(want))
