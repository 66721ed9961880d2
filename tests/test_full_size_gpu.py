"""configs[2] / [3] / [4] at their real size (3 x 800 x 1333, b = 8) through the real drivers on one GPU:
`train.main --blur_train --gpu_blur --use_stored_psfs --expand_target_boxes` (reference train.py:186-205, 294-322;
engine.py:74-158) and the `evaluate.main --use_ensemble --LEHE` sweep over all 15 cells (reference evaluate.py:299-370;
engine.py:284-392).  Size-independent checks: finite losses, a checkpoint, 12 COCO statistics per cell, and the first
batch that reaches `blur_image_list` compared with the oracle on an interior crop (a crop's reflect padding only
differs from the image's within 64 pixels of the crop's border)."""
import os

import numpy as np
import pytest
import torch

import dib_oracle as O

pytestmark = pytest.mark.gpu

CROP = (slice(None), slice(236, 492), slice(436, 756))        # 256 x 320, interior 128 x 192


def _tap_first_blur(monkeypatch):
    """Records the first `blur_image_list` call the engines make: inputs (crop), dicts, PSFs, outputs (crop)."""
    from detectinblur_amd.models import blur_functions as BF
    rec = {}
    real = BF.blur_image_list

    def tapped(images_GPU, blur_dicts, psfs_GPU, *a, **k):
        first = not rec
        if first:
            rec["shapes"] = [tuple(i.shape) for i in images_GPU]
            rec["dtypes"] = {i.dtype for i in images_GPU}
            rec["in"] = [i[CROP].cpu().numpy().copy() for i in images_GPU]
            rec["dicts"] = [{"blurring": bool(d["blurring"])} for d in blur_dicts]
            rec["psfs"] = [p.cpu().numpy().copy() for p in psfs_GPU]
            rec["tables"] = k.get("tables") is not None
        ret = real(images_GPU, blur_dicts, psfs_GPU, *a, **k)
        if first:
            rec["out"] = [i[CROP].cpu().numpy().copy() for i in images_GPU]
        rec["calls"] = rec.get("calls", 0) + 1
        return ret

    monkeypatch.setattr(BF, "blur_image_list", tapped)
    return rec


def _check_against_oracle(rec):
    want = [a.copy() for a in rec["in"]]
    O.blur_image_list(want, rec["dicts"], rec["psfs"])
    n_blurred = 0
    for got, w, d, before in zip(rec["out"], want, rec["dicts"], rec["in"]):
        if d["blurring"]:
            n_blurred += 1
            assert not np.array_equal(got, before)
            assert np.array_equal(got[:, 64:-64, 64:-64].view(np.uint16), w[:, 64:-64, 64:-64].view(np.uint16))
        else:
            assert np.array_equal(got.view(np.uint16), before.view(np.uint16))
    return n_blurred


def _psf_store(tmp_path, count):
    from detectinblur_amd.dataset_utils import generate_PSFs
    dest = str(tmp_path) + "/"
    generate_PSFs.main(generate_PSFs.get_parser().parse_args(["--destination_path", dest, "--num_workers", "1", "--total_num_psfs", str(count),
                                                              "--device", "cuda"]))
    return dest + "psfs"


def test_train_main_full_size_stored_psfs(tmp_path, monkeypatch, capsys):
    from detectinblur_amd import train
    store = _psf_store(tmp_path, 48)
    rec = _tap_first_blur(monkeypatch)
    out_dir = tmp_path / "weights"
    args = train.build_parser().parse_args([
        "--synthetic", "--synthetic_images", "48", "--synthetic_size", "800", "1333", "-b", "8", "-j", "2", "--epochs", "1",
        "--blur_train", "--gpu_blur", "--use_stored_psfs", "--stored_psf_directory", store, "--stored_psf_count", "48",
        "--param_index", "1", "--low_exposure", "--expand_target_boxes", "--early_stop", "3", "--lr", "0.002", "--print_freq", "1",
        "--output_dir", str(out_dir), "--tensorboard_path", str(tmp_path / "tb")])
    train.main(args)                   # exits the process on a non-finite loss (reference engine.py:145-148)
    text = capsys.readouterr().out
    assert rec["shapes"] == [(3, 800, 1333)] * 8 and rec["dtypes"] == {torch.float16} and rec["tables"]
    blurred = _check_against_oracle(rec)
    assert 1 <= blurred <= 8            # --low_exposure blurs 75 % of the images (reference train.py:139-140)
    assert all(p.shape == (128, 128) for p, d in zip(rec["psfs"], rec["dicts"]) if d["blurring"])
    # 5 training batches (early_stop 3 ends the epoch on its 5th iteration) + 4 blurred evaluation images
    assert rec["calls"] == 5 + 4
    assert text.count("Epoch: [0]") >= 4 and "loss" in text      # the iteration that breaks prints no progress line
    ck = torch.load(out_dir / "model_0.pth", map_location="cpu", weights_only=False)
    assert ck["epoch"] == 0 and all(torch.isfinite(v).all() for v in ck["model"].values() if v.is_floating_point())
    assert text.count("IoU metric: bbox") == 2         # the clean and the blurred evaluation pass (train.py:345-387)


def test_evaluate_main_full_size_ensemble_sweep(tmp_path, monkeypatch, capsys):
    from detectinblur_amd import evaluate
    rec = _tap_first_blur(monkeypatch)
    args = evaluate.build_parser().parse_args([
        "--synthetic", "--synthetic_images", "4", "--synthetic_size", "800", "1333", "-j", "2", "--use_ensemble", "--LEHE",
        "--use_blur_estimator", "--blur_eval", "--gpu_blur", "--expand_target_boxes", "--early_stop", "1",
        "--tensorboard_path", str(tmp_path / "tb")])
    results = evaluate.main(args)
    assert sorted(results) == sorted("P%dE%d" % (p, e) for p in (1, 2, 3) for e in range(5))
    for cell, out in results.items():
        stats = out.coco_eval["bbox"].stats
        assert len(stats) == 12 and all(np.isfinite(s) and -1.0 <= s <= 1.0 for s in stats), cell
        assert len(out["detections"]) == 2 and len(out["routes"]) == 2 and all(r in (0, 1, 2, 3) for r in out["routes"])
        for det in out["detections"].values():
            assert det["boxes"].shape[1] == 4 and torch.isfinite(det["boxes"]).all()
            assert (det["boxes"][:, 0] >= 0).all() and (det["boxes"][:, 2] <= 1333).all() and (det["boxes"][:, 3] <= 800).all()
        for boxes in out["targets"].values():           # expanded ground truth, xywh, inside the image
            assert (boxes[:, 2] > 0).all() and (boxes[:, 3] > 0).all()
    assert rec["calls"] == 30 and rec["shapes"] == [(3, 800, 1333)] and rec["tables"]
    assert _check_against_oracle(rec) == 1
    assert any(f.startswith("events.out.tfevents") for f in os.listdir(tmp_path / "tb"))


def test_two_runs_of_evaluate_main_print_identical_coco_stats(tmp_path):
    """Evaluation is reproducible on the GPU: the same `evaluate.main` command in two fresh processes -- native-size images (the
    resize path of the input transform), on-the-fly PSFs from seeded loader workers, HIP blur (standard and large window), box
    growth, eager and graphed trunk, RoI heads, COCO evaluator, all 15 sweep cells -- prints the same 180 statistic lines and
    produces the same detections and expanded ground truth bit for bit (round 3 could not: MIOpen's atomically accumulating
    kernels in the inference path, profiles/r4_nondeterminism.txt; reference engine.py:275-392).  (Two runs inside ONE process
    are not the claim: MIOpen's choice among equally ranked kernels depends on the workspace the allocator can spare at
    that moment.)"""
    from tests import _gpu_children
    from tests.test_ddp_gpu import _run_child
    argv = ["--synthetic", "--synthetic_images", "5", "--synthetic_size", "480", "640", "-j", "2", "--blur_eval", "--gpu_blur",
            "--expand_target_boxes", "--early_stop", "3"]
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    a = _run_child(_gpu_children.evaluate_main_digest, tmp_path / "a", argv)
    b = _run_child(_gpu_children.evaluate_main_digest, tmp_path / "b", argv)
    # What the two processes' kernel choices rested on is compared FIRST: when the outputs differ AND the choices did (a find step for a
    # shape the shipped data does not hold, ranked differently by the two processes; another MIOpen build ignoring the shipped files),
    # the failure is MIOpen / PyTorch choosing other kernels, not a defect of the evaluation path -- and it says so, instead of 180
    # differing statistic lines.  (tests/test_kernel_choices_gpu.py is where a stack the shipped data does not belong to fails.)
    ka, kb = a["kernel_choice"], b["kernel_choice"]
    drift = None
    if ka != kb or not ka["installed"] or ka["miopen_foreign_files"] or not ka["tunableop_validators_match"]:
        drift = ("the two processes did not rest on the same kernel choices (MIOpen find results appended to the private db: %s / %s bytes; "
                 "foreign db files %s; TunableOp validators match: %s): a kernel-choice drift, not a defect of the evaluation path -- %r vs %r"
                 % (ka["miopen_db_growth_bytes"], kb["miopen_db_growth_bytes"], ka["miopen_foreign_files"], ka["tunableop_validators_match"], ka, kb))
    same = a["stat_lines"] == b["stat_lines"] and all(a["cells"].get(c) == b["cells"].get(c) for c in a["cells"])
    assert same or drift is None, drift
    assert len(a["stat_lines"]) == 12 * 15 and a["stat_lines"] == b["stat_lines"]
    assert sorted(a["cells"]) == sorted(b["cells"]) and len(a["cells"]) == 15
    for cell in a["cells"]:
        assert a["cells"][cell] == b["cells"][cell], cell
        assert a["cells"][cell]["images"] == 4
    assert sum(c["boxes"] for c in a["cells"].values()) > 0


def test_pipelined_evaluation_equals_the_plain_loop(monkeypatch):
    """engine.evaluate on the GPU runs three images at once (heads of image i - 1, trunk of image i, blur / estimator of image
    i + 1 queued before anything is waited for; detections read from pinned memory an iteration later).  Same routes, same
    detections bit for bit, same COCO statistics as the plain one-image-at-a-time loop (DIB_NO_PIPELINE=1) -- ensemble + estimator at
    800 x 1333, expanded boxes, 7 images (odd: the pipeline's tail), and once more with early_stop."""
    import contextlib
    import io
    from torch import nn
    from detectinblur_amd import engine, utils
    from detectinblur_amd.coco_utils import SyntheticCocoDetection
    from detectinblur_amd.models.blur_estimator import resnet18
    from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
    from detectinblur_amd.train import get_transform
    dev = torch.device("cuda", 0)
    torch.manual_seed(1337)
    ens = [fasterrcnn_resnet50_fpn(num_classes=91, pretrained=False, pretrained_backbone=False).to(dev).eval() for _ in range(2)]
    est = resnet18(); est.fc = nn.Linear(512, 4); est = est.to(dev).eval()
    with contextlib.redirect_stdout(io.StringIO()):
        tf = get_transform(False, blur=True, blur_type=0.001, blur_ratio=1, blur_exposure=0.5)
    ds = SyntheticCocoDetection(num_images=7, size=(800, 1333), transforms=tf)

    class L(list):
        dataset = ds

    batches = L(utils.collate_fn([ds[i]]) for i in range(7))
    kw = dict(blurring_images=True, gpu_blur=True, expand_target_boxes=True, use_ensemble=True, ensemble_models=ens + ens, blur_estimator=est, LEHE=True)
    out = {}
    for stop in (None, 3):
        for plain in ("", "1"):
            if plain:
                monkeypatch.setenv("DIB_NO_PIPELINE", "1")
            else:
                monkeypatch.delenv("DIB_NO_PIPELINE", raising=False)
            with contextlib.redirect_stdout(io.StringIO()):
                r = engine.evaluate(None, batches, dev, early_stop=stop, **kw)
            out[(stop, plain)] = r
        a, b = out[(stop, "")], out[(stop, "1")]
        assert a.routes == b.routes and len(a.routes) == (7 if stop is None else stop + 1)
        assert list(a.detections) == list(b.detections) and len(a.detections) == len(a.routes)
        for k in a.detections:
            for f in ("boxes", "scores", "labels"):
                assert a.detections[k][f].device.type == "cpu" and torch.equal(a.detections[k][f], b.detections[k][f]), (k, f)
            assert torch.equal(a.targets[k], b.targets[k])
        assert np.array_equal(np.asarray(a.coco_stats), np.asarray(b.coco_stats))
        assert sum(v["scores"].numel() for v in a.detections.values()) > 50


def test_split_forward_pass_equals_forward_and_survives_odd_cases(monkeypatch):
    """GeneralizedRCNN.launch_trunk / launch_heads / finish (what the pipelined evaluation calls) against `forward` on the same
    image; then with the detection kernels switched off (the heads finish through the tensor path) and with an image whose RPN
    keeps no proposal at all (one padding row, no detection)."""
    from detectinblur_amd.models import detector_ops as ops
    from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
    torch.manual_seed(0)
    m = fasterrcnn_resnet50_fpn(num_classes=91, pretrained=False, pretrained_backbone=False).cuda().eval()
    m.graph_inference = True
    img = torch.rand(3, 800, 1088, device="cuda")
    means, stds = np.tile([0.485, 0.456, 0.406], (1, 1)), np.tile([0.229, 0.224, 0.225], (1, 1))

    def split():
        h = m.launch_trunk([img], killWarp=True, newMeans=means, newSTDs=stds)
        torch.cuda.synchronize()
        assert m.launch_heads(h) is h
        torch.cuda.synchronize()
        return m.finish(h)[0]

    with torch.no_grad():
        for _ in range(2):
            want = m([img], killWarp=True, newMeans=means, newSTDs=stds)[0]      # second sighting: the trunk graph is captured
        got = split()
        for k in ("boxes", "scores", "labels"):
            assert got[k].device.type == "cpu" and torch.equal(got[k], want[k].cpu()), k
        assert got["scores"].numel() > 10
        monkeypatch.setattr(ops, "HIP_BOXES", False)
        plain = split()
        monkeypatch.setattr(ops, "HIP_BOXES", True)
        for k in ("boxes", "scores", "labels"):
            assert torch.equal(plain[k], got[k]), k
        # no proposal survives: the RPN's minimum size above every box
        monkeypatch.setattr(m.rpn, "min_size", 1e9)
        m.__dict__.pop("_trunk_graphs")                          # the captured trunk has the old value in its kernel arguments
        empty = split()
        assert empty["boxes"].shape == (0, 4) and empty["scores"].numel() == 0 and empty["labels"].dtype == torch.int64
    assert m.launch_trunk([img.cpu()], killWarp=True) is None and m.train().launch_trunk([img], killWarp=True) is None
