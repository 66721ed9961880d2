"""GPU integration: one blurred training step and one blurred evaluation pass through the engines
(`--blur_train --gpu_blur --expand_target_boxes`), exactly the call sequence of reference engine.py:74-158."""
import numpy as np
import pytest
import torch

import dib_oracle as O

pytestmark = pytest.mark.gpu


def _model():
    from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
    torch.manual_seed(0)
    return fasterrcnn_resnet50_fpn(pretrained=False, pretrained_backbone=False, num_classes=91, min_size=160, max_size=224,
                                   rpn_pre_nms_top_n_train=400, rpn_post_nms_top_n_train=200, rpn_post_nms_top_n_test=100).cuda()


def _loader(train, n=4):
    import random
    from detectinblur_amd import utils
    from detectinblur_amd.coco_utils import SyntheticCocoDetection
    from detectinblur_amd.train import get_transform
    random.seed(3); np.random.seed(3)
    tf = get_transform(train, blur=True, blur_type=0.005, blur_ratio=1.0, low_exposure=True)
    ds = SyntheticCocoDetection(num_images=n, size=(150, 210), boxes_per_image=4, transforms=tf)
    return torch.utils.data.DataLoader(ds, batch_size=2 if train else 1, collate_fn=utils.collate_fn)


def test_blurred_train_step_and_eval():
    from detectinblur_amd.engine import evaluate, train_one_epoch
    m = _model()
    opt = torch.optim.SGD(m.parameters(), lr=0.002, momentum=0.9)
    w0 = m.backbone.body.conv1.weight.detach().clone()
    log = train_one_epoch(m, opt, _loader(True), torch.device("cuda"), epoch=0, print_freq=1, blur_train=True, gpu_blur=True,
                          expand_target_boxes=True, use_custom_image_norm=True, early_stop=None)
    assert log.meters["loss"].count == 2 and np.isfinite(log.meters["loss"].global_avg)
    assert not torch.equal(w0, m.backbone.body.conv1.weight.detach())
    out = evaluate(m, _loader(False), torch.device("cuda"), blurring_images=True, gpu_blur=True, expand_target_boxes=True)
    assert len(out["detections"]) == 4
    assert out["coco_stats"] is not None and len(out["coco_stats"]) == 12 and all(-1.0 <= v <= 1.0 for v in out["coco_stats"])


def test_engine_blur_equals_oracle_on_loader_batch():
    """What the engine feeds the detector is bit-identical to the oracle's blur of the same batch."""
    from detectinblur_amd import engine, utils
    from detectinblur_amd.models import blur_functions
    images_CPU, targets, blur_dicts = next(iter(_loader(True)))
    imgs, tg, psfs, *_ = engine._to_device(images_CPU, targets, blur_dicts, torch.device("cuda"), True)
    want_imgs = [i.half().numpy().copy() for i in images_CPU]
    want_psfs = [O.to_half_like_torch(bd["psf"]) for bd in blur_dicts]
    O.blur_image_list(want_imgs, blur_dicts, want_psfs)
    want_boxes = [O.expand_boxes(t["boxes"].numpy(), p, i.shape[1], i.shape[2]) for t, p, i in zip(targets, want_psfs, images_CPU)]
    blur_functions.blur_image_list(imgs, blur_dicts, psfs_GPU=psfs)
    tg = utils.expand_targets(tg, blur_dicts, psfs, imgs)
    for g, w in zip(imgs, want_imgs):
        assert np.array_equal(g.cpu().numpy().view(np.uint16), w.view(np.uint16))
    for t, w in zip(tg, want_boxes):
        assert np.array_equal(t["boxes"].cpu().numpy(), w)


def test_ensemble_routing_runs():
    from torch import nn
    from detectinblur_amd.engine import evaluate
    from detectinblur_amd.models.blur_estimator import resnet18
    nets = [_model() for _ in range(2)] * 2
    est = resnet18(); est.fc = nn.Linear(512, 4); est = est.cuda()
    out = evaluate(None, _loader(False, 2), torch.device("cuda"), blurring_images=True, gpu_blur=True, expand_target_boxes=True,
                   use_ensemble=True, ensemble_models=nets, blur_estimator=est, LEHE=True)
    assert len(out["routes"]) == 2 and all(r in (0, 1, 2, 3) for r in out["routes"])


def test_train_cli_with_stored_psfs_end_to_end(tmp_path, capsys):
    """BASELINE configs[3] in miniature: `train.py --blur_train --gpu_blur --use_stored_psfs` on a store
    written by the store builder (2 PSFs per directory), two iterations, finite losses, checkpoint saved."""
    from detectinblur_amd import train
    from detectinblur_amd.dataset_utils import generate_PSFs as G
    dest = str(tmp_path) + "/"
    G.main(G.get_parser().parse_args(["--destination_path", dest, "--num_workers", "1", "--total_num_psfs", "2", "--packed"]))
    out = str(tmp_path / "run")
    args = train.build_parser().parse_args([
        "--synthetic", "--synthetic_images", "4", "--synthetic_size", "160", "224", "--blur_train", "--gpu_blur",
        "--use_stored_psfs", "--stored_psf_directory", dest + "psfs", "--stored_psf_count", "2", "--param_index", "1",
        "--low_exposure", "--expand_target_boxes", "--use_custom_image_norm", "-b", "2", "--epochs", "1", "--early_stop", "2",
        "--lr", "0.002", "--print_freq", "1", "--output_dir", out, "--tensorboard_path", str(tmp_path / "tb")])
    train.main(args)
    text = capsys.readouterr().out
    assert "loss_classifier" in text and "nan" not in text.lower().split("namespace")[-1]


def test_evaluate_cli_ensemble_sweep_end_to_end(capsys, tmp_path):
    """BASELINE configs[4] in miniature: the (type x exposure) sweep through 4 detectors + estimator."""
    from detectinblur_amd import evaluate as E
    args = E.build_parser().parse_args(["--synthetic", "--synthetic_images", "2", "--synthetic_size", "160", "224", "--use_ensemble",
                                      "--LEHE", "--use_blur_estimator", "--blur_eval", "--gpu_blur", "--expand_target_boxes",
                                      "--early_stop", "1", "--tensorboard_path", str(tmp_path / "tb")])
    res = E.main(args)
    text = capsys.readouterr().out
    assert text.count("routes") == 15           # 3 blur types x 5 exposures
    assert len(res) == 15 and res["P1E0"].coco_eval["bbox"].stats.shape == (12,)
    from detectinblur_amd import tb_writer
    import os
    ev = os.path.join(str(tmp_path / "tb"), os.listdir(str(tmp_path / "tb"))[0])
    if not ev.endswith(".v2"):
        tags = {(t, s) for t, s, _ in tb_writer.read_scalars(ev)}
        assert ("P1/AccuraciesSweep", 1) in tags and ("P3/recall", 5) in tags      # reference evaluate.py:358-368


def test_train_cli_with_full_corruption_chain(tmp_path, capsys):
    """--add_noise --add_block --add_jpeg_artefacts behind the GPU blur, two iterations."""
    from detectinblur_amd import train
    args = train.build_parser().parse_args([
        "--synthetic", "--synthetic_images", "4", "--synthetic_size", "160", "224", "--blur_train", "--gpu_blur",
        "--param_index", "2", "--high_exposure", "--expand_target_boxes", "--add_noise", "--add_block", "--add_jpeg_artefacts",
        "-b", "2", "--epochs", "1", "--early_stop", "2", "--lr", "0.002", "--print_freq", "1", "--output_dir", str(tmp_path / "run"),
        "--tensorboard_path", str(tmp_path / "tb")])
    train.main(args)
    text = capsys.readouterr().out
    assert "loss_classifier" in text and "Loss is" not in text


def test_degenerate_boxes_raise_on_the_gpu_without_draining_the_queue():
    """The degenerate-box flag travels to pinned host memory behind the transform and is read at the end of forward
    (reference models/generalized_rcnn.py:119-129: same ValueError, same text)."""
    m = _model()
    m.train()
    imgs = [torch.rand(3, 150, 210, device="cuda"), torch.rand(3, 150, 210, device="cuda")]
    good = {"boxes": torch.tensor([[10., 20., 60., 70.]], device="cuda"), "labels": torch.tensor([3], device="cuda")}
    bad = {"boxes": torch.tensor([[5., 5., 50., 60.], [30., 40., 30., 90.]], device="cuda"), "labels": torch.tensor([1, 2], device="cuda")}
    means = np.tile([0.485, 0.456, 0.406], (2, 1)); stds = np.tile([0.229, 0.224, 0.225], (2, 1))
    with pytest.raises(ValueError, match=r"positive height and width. Found invaid box \[.*\] for target at index 1"):
        m(imgs, [good, bad], newMeans=means, newSTDs=stds)
    losses = m(imgs, [good, good], newMeans=means, newSTDs=stds)          # and the flag is clean again on the next call
    assert all(torch.isfinite(v) for v in losses.values())


def test_graphed_inference_equals_eager_inference():
    """engine.evaluate runs the detector's static trunk (backbone + FPN + RPN head + proposal filtering) as a HIP graph per
    input shape: the recorded launches of the eager code.  The trunk's outputs (pyramid features, the best proposals) equal the
    eager ones to the ~1e-6 by which two eager passes differ (MIOpen's kernel choice varies between calls); the detections
    of a random-init head sit on score / NMS thresholds, so their count may differ by a few and the best ones must agree --
    for two shapes, repeated replays and changing image sizes inside one padded shape."""
    m = _model().eval()
    torch.manual_seed(1)
    batches = [[torch.rand(3, 150, 210, device="cuda")], [torch.rand(3, 150, 210, device="cuda")], [torch.rand(3, 140, 200, device="cuda")],
               [torch.rand(3, 120, 224, device="cuda"), torch.rand(3, 150, 190, device="cuda")], [torch.rand(3, 150, 210, device="cuda")]]
    means = lambda n: np.tile([0.485, 0.456, 0.406], (n, 1))          # noqa: E731
    stds = lambda n: np.tile([0.229, 0.224, 0.225], (n, 1))           # noqa: E731
    with torch.no_grad():
        eager = [m(list(b), newMeans=means(len(b)), newSTDs=stds(len(b))) for b in batches]
        m.graph_inference = True
        graphed = [m(list(b), newMeans=means(len(b)), newSTDs=stds(len(b))) for b in batches]
        again = [m(list(b), newMeans=means(len(b)), newSTDs=stds(len(b))) for b in batches]
        assert len(m._trunk_graphs.graphs) == 2 and all(g is not None for g in m._trunk_graphs.graphs.values())     # captured, not eager
        # the trunk itself: graph replay vs the eager function on the same padded batch
        for b in (batches[0], batches[3]):
            imgs, _ = m.transform(list(b), None, means(len(b)), stds(len(b)))
            m._sizes[len(b)].copy_(torch.tensor([[float(s[1]), float(s[0])] for s in imgs.image_sizes], device="cuda"))
            want = [t.clone() for t in m._trunk(imgs.tensors)]
            got = m._trunk_graphs(imgs.tensors)
            for w, g in zip(want[:-3], got[:-3]):                                   # pyramid features
                assert torch.allclose(g, w, rtol=0, atol=1e-4 * float(w.abs().max()))
            nw, ng = want[-1].tolist(), got[-1].tolist()                            # proposals kept after NMS, best first
            for i in range(len(b)):
                assert abs(nw[i] - ng[i]) <= 5 and ng[i] > 10
                assert torch.allclose(got[-3][i, :10], want[-3][i, :10], atol=1e-2) and torch.allclose(got[-2][i, :10], want[-2][i, :10], atol=1e-4)
    for e, g, a in zip(eager, graphed, again):
        for de, dg, da in zip(e, g, a):
            for other in (de, da):
                assert abs(len(dg["boxes"]) - len(other["boxes"])) <= 5
                k = min(3, len(dg["boxes"]), len(other["boxes"]))
                assert torch.allclose(dg["scores"][:k], other["scores"][:k], atol=1e-4)
    assert len(eager[0][0]["boxes"]) > 0
