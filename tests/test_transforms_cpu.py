"""CPU tests of the dataset-side transforms against golden blur_dicts from the real reference:
same Python-`random` / numpy draw order, same PSFs (bit-exact float64), same statistics."""
import os
import random
import tempfile

import numpy as np
import pytest
import torch

import golden_inputs as GI


def _psf_from_golden(golden, key, rec):
    psf = np.zeros(rec["psf_shape"], dtype=rec["psf_dtype"])
    psf[golden.blurdict[key + "_r"].astype(int), golden.blurdict[key + "_c"].astype(int)] = golden.blurdict[key + "_v"]
    return psf


@pytest.mark.parametrize("mode", GI.blurimage_modes(), ids=lambda m: m["name"])
def test_blurimage_matches_reference(golden, mode, capsys):
    from detectinblur_amd.transforms import BlurImage
    kw = dict(mode["kwargs"])
    tmp = None
    if kw.get("use_stored_psfs"):
        tmp = tempfile.mkdtemp(prefix="dib_psfs_")
        kw["stored_psf_directory"] = tmp
    want = golden.meta["blurimage"][mode["name"]]
    random.seed(mode["seed"])
    np.random.seed(mode["seed"])
    bi = BlurImage(blur_image_in_transform=False, **kw)
    for call, rec in enumerate(want["records"]):
        if kw.get("use_stored_psfs"):
            st = random.getstate()
            pred = GI.predict_stored_draw(kw)
            random.setstate(st)
            if pred is not None:
                d = os.path.join(tmp, "P%dE%d" % (pred[0], pred[1]))
                os.makedirs(d, exist_ok=True)
                with open(os.path.join(d, "I%06d" % pred[2]), "wb") as f:
                    np.save(f, GI.stored_psf(*pred))
        img, tgt, bd = bi("IMG", {"t": call}, {})
        assert img == "IMG" and tgt == {"t": call}
        assert bd["blurring"] == rec["blurring"]
        if not rec["blurring"]:
            assert bd["psf"] == rec["psf"] and bd["theta_rad"] == 0 and bd["param_index"] is None
            continue
        psf = _psf_from_golden(golden, "bd_%s_%d" % (mode["name"], call), rec)
        assert bd["psf"].dtype == psf.dtype and bd["psf"].shape == psf.shape
        assert np.array_equal(bd["psf"], psf)
        assert float(bd["theta_rad"]).hex() == rec["theta_rad"]
        assert float(bd["scale_factor_lambda1"]).hex() == rec["scale_factor_lambda1"]
        assert float(bd["scale_factor_lambda2"]).hex() == rec["scale_factor_lambda2"]
        assert bd["param_index"] == rec["param_index"]
        assert bd["fraction_index"] == rec["fraction_index"]
        rr, cc = np.nonzero(psf > 0)
        assert bd["psf_extent"] == (rr.min(), rr.max(), cc.min(), cc.max())
        assert bd["psf_taps"] == len(rr)
        assert 1 <= bd["psf_segments"][1] <= bd["psf_segments"][0] <= len(rr)
    # both RNG streams end where the reference's end
    assert random.random() == want["next_random"]
    assert float(np.random.uniform()) == want["next_np"]


def test_preblurred_passthrough(golden):
    from detectinblur_amd.transforms import BlurImage
    _, _, bd = BlurImage(prob=1.0, blur_image_in_transform=False)("IMG", None, {"preBlurred": True})
    assert bd == golden.meta["blurimage"]["preblurred"]


def test_compose_totensor_flip():
    from detectinblur_amd import transforms as T
    img = (np.random.RandomState(0).random_sample((20, 30, 3)) * 255).astype(np.uint8)
    target = {"boxes": torch.tensor([[2., 3., 10., 12.], [0., 0., 30., 20.]])}
    random.seed(3)
    out, tgt, bd = T.Compose([T.ToTensor(), T.RandomHorizontalFlip(1.0)])(img, target, {"x": 1}, epoch_number=4)
    assert out.shape == (3, 20, 30) and out.dtype == torch.float32
    assert torch.equal(out, torch.from_numpy(img.transpose(2, 0, 1).copy()).float().div(255).flip(-1))
    assert torch.equal(tgt["boxes"], torch.tensor([[20., 3., 28., 12.], [0., 0., 30., 20.]]))
    assert bd == {"x": 1, "epoch_number": 4, "dryRun": False}


def test_cpu_blur_handler_matches_golden(golden):
    """--cpu_blur path: within one grey level of the reference run with the numpy cv2 shim."""
    from PIL import Image
    from detectinblur_amd.motion_blur.blur_image import BlurImageHandler
    h = BlurImageHandler(image_path=None, PSFs=[GI.make_fft_psf().astype(np.float32)], pillowImage=Image.fromarray(GI.make_fft_image()))
    assert h.blur_image()
    out = np.array(h.pilImageResult)
    want = golden.fft["fft_out"]
    assert out.shape == want.shape and np.abs(out.astype(int) - want.astype(int)).max() <= 1


def test_blurimage_cpu_mode_runs():
    from PIL import Image
    from detectinblur_amd.transforms import BlurImage
    random.seed(1)
    np.random.seed(1)
    img = Image.fromarray(GI.make_fft_image())
    out, _, bd = BlurImage(prob=1.0, low_exposure=True, blur_image_in_transform=True)(img, None, {})
    assert bd["blurring"] and out.size == img.size and out is not img
