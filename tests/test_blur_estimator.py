"""Blur-estimator labels (CPU) and the training / evaluation loops on the GPU blur."""
import numpy as np
import pytest
import torch

from detectinblur_amd import engine_blur_estimator as EB


def _bd(blurring, p=None, f=None, **kw):
    return dict({"blurring": blurring, "param_index": p, "fraction_index": f}, **kw)


def test_labels_16_way_and_lehe():
    dicts = [_bd(False), _bd(True, 0, 0), _bd(True, 1, 2), _bd(True, 2, 4), _bd(True, 0, 3), _bd(True, 2, 2),
             _bd(True, 1, 4, blur_est_label=3), _bd(False, blur_est_label=2)]
    t16 = EB.get_target_from_blur_dict(dicts[:6], torch.zeros(6, dtype=torch.long))
    assert t16.tolist() == [0, 1, 8, 15, 4, 13]
    t4 = EB.get_target_from_blur_dict_LEHE(dicts, torch.zeros(8, dtype=torch.long))
    assert t4.tolist() == [0, 0, 0, 3, 1, 0, 3, 2]


def test_accuracy_topk():
    out = torch.tensor([[0.1, 0.7, 0.2], [0.5, 0.2, 0.3], [0.2, 0.3, 0.5], [0.9, 0.04, 0.06]])
    tgt = torch.tensor([1, 2, 2, 1])
    a1, a2 = EB.accuracy(out, tgt, topk=(1, 2))
    assert float(a1) == 50.0 and float(a2) == 75.0


@pytest.mark.gpu
def test_estimator_train_and_eval_cli(tmp_path, capsys):
    from detectinblur_amd import train_blur_estimator as TB
    args = TB.build_parser().parse_args([
        "--synthetic", "--synthetic_images", "8", "--synthetic_size", "160", "224", "--blur_train", "--gpu_blur", "--LEHE_blur_seg",
        "--crop_images", "-b", "4", "--epochs", "1", "--early_stop", "2", "--lr", "0.01", "--print_freq", "1",
        "--output_dir", str(tmp_path / "est")])
    TB.main(args)
    text = capsys.readouterr().out
    assert "Blur estimator accuracy" in text and "loss" in text
    assert (tmp_path / "est" / "blur_estimator_0.pth").exists()


@pytest.mark.gpu
def test_resize_round_trip_blur_matches_manual_composition():
    """resize_images: interpolate to height 800, blur with the HIP path, interpolate back."""
    import torch.nn.functional as F
    from detectinblur_amd.models import blur_functions as BF
    g = torch.Generator().manual_seed(3)
    img = torch.rand(3, 120, 200, generator=g).half().cuda()
    psf = torch.zeros(128, 128, dtype=torch.float16)
    psf[60:66, 63] = 1.0; psf[63, 58:70] = 0.5
    psf = psf.cuda()
    imgs = [img.clone(), img.clone()]
    EB.blur_image_list(imgs, [{"blurring": True}, {"blurring": False}], [psf, psf], resize_images=True)
    up = F.interpolate(img.unsqueeze(0), size=(800, int(800 * 200 / 120)), mode="bilinear").squeeze(0)
    want = BF.manual_blur(up, psf / psf.sum())
    want = F.interpolate(want.unsqueeze(0), size=(120, 200), mode="bilinear").squeeze(0)
    assert torch.equal(imgs[0], want) and torch.equal(imgs[1], img)
