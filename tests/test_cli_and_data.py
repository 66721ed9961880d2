"""The reference's command-line surface, its COCO feed and its evaluation contract:
  * the four command lines of the reference README parse unchanged through train / evaluate `build_parser`;
  * a COCO tree is read with json + PIL only, targets laid out as reference coco_utils.py:51-104;
  * `engine.evaluate` returns an object with `.coco_eval["bbox"].stats` (reference engine.py:416, train.py:350-387);
  * a checkpoint written by the training loop resumes to identical weights, optimizer and schedule."""
import json
import os
import shlex

import numpy as np
import pytest
import torch

from detectinblur_amd import coco_utils, utils
from detectinblur_amd import evaluate as EV
from detectinblur_amd import train as TR

# /root/reference/README.md lines 24, 28, 33 (evaluate.py) and 49, 53, 57 (train.py), arguments only
README_EVAL = [
    "-j 3 --tensorboard_path evals/test --blur_eval --gpu_blur --data_path /mnt/data_f2/mosayed/COCO/coco/ --resume weights/resnet50FPNBlur.pth",
    "-j 3 --tensorboard_path evals/test --blur_eval --gpu_blur --data_path /mnt/data_f2/mosayed/COCO/coco/ --resume weights/resnet50FPNBlurExpand.pth --expand_target_boxes",
    '-j 3 --tensorboard_path evals/test --blur_eval --gpu_blur --data_path /mnt/data_f2/mosayed/COCO/coco/ --expand_target_boxes --use_ensemble --LEHE --blur_estimator_path weights/SpecByExpEstimator.pth --ensemble_model_paths "weights/resnet50FPNBlurLEExpand.pth weights/resnet50FPNBlurP1HEExpand.pth weights/resnet50FPNBlurP2HEExpand.pth weights/resnet50FPNBlurP3HEExpand.pth"',
    "-j 3 --tensorboard_path evals/test --pretrained --dataset GOPRO --data_path /media/mosayed/data_f_256/datasets/GOPRO --blurred_dataset",
]
README_TRAIN = [
    "-j 3 -b 8 --lr 0.04 --epochs 35 --lr-steps 16 21 --aspect-ratio-group-factor 3 --model fasterrcnn_resnet50_fpn --tensorboard_path runs/test --output_dir weights/test --pretrained --blur_train --gpu_blur --data_path /mnt/data_f2/mosayed/COCO/coco/ --stored_psf_directory /mnt/data_f2/mosayed/COCO/coco/psfs --use_stored_psfs",
    "-j 3 -b 8 --lr 0.04 --epochs 35 --lr-steps 16 21 --aspect-ratio-group-factor 3 --model fasterrcnn_resnet50_fpn --tensorboard_path runs/test --output_dir weights/test --pretrained --blur_train --gpu_blur --data_path /mnt/data_f2/mosayed/COCO/coco/ --stored_psf_directory /mnt/data_f2/mosayed/COCO/coco/psfs --use_stored_psfs --param_index 3 --high_exposure",
    "-j 3 -b 8 --lr 0.04 --epochs 35 --lr-steps 16 21 --aspect-ratio-group-factor 3 --model fasterrcnn_resnet50_fpn --tensorboard_path runs/test --output_dir weights/test --pretrained --blur_train --gpu_blur --data_path /mnt/data_f2/mosayed/COCO/coco/ --stored_psf_directory /mnt/data_f2/mosayed/COCO/coco/psfs --use_stored_psfs --param_index 3 --high_exposure --expand_target_boxes",
]


def test_reference_readme_command_lines_parse_unchanged():
    a = EV.build_parser().parse_args(shlex.split(README_EVAL[0]))
    assert a.resume == "weights/resnet50FPNBlur.pth" and a.gpu_blur and a.blur_eval and a.workers == 3 and a.tensorboard_path == "evals/test"
    assert EV.build_parser().parse_args(shlex.split(README_EVAL[1])).expand_target_boxes
    a = EV.build_parser().parse_args(shlex.split(README_EVAL[2]))
    assert a.use_ensemble and a.LEHE and a.blur_estimator_path.endswith("SpecByExpEstimator.pth")
    assert len(a.ensemble_model_paths) == 1 and len(a.ensemble_model_paths[0].split()) == 4     # one quoted string, split in main()
    a = EV.build_parser().parse_args(shlex.split(README_EVAL[3]))                                 # parses; refused when run
    assert a.dataset == "GOPRO" and a.blurred_dataset and a.pretrained
    with pytest.raises(SystemExit, match="outside the built hot path"):
        TR.reject_out_of_scope(a)
    for line in README_TRAIN:
        a = TR.build_parser().parse_args(shlex.split(line))
        assert a.aspect_ratio_group_factor == 3 and a.lr_steps == [16, 21] and a.use_stored_psfs and a.pretrained
        assert a.tensorboard_path == "runs/test" and a.output_dir == "weights/test" and a.blur_train and a.gpu_blur
        TR.reject_out_of_scope(a)
    assert a.param_index == "3" and a.high_exposure and a.expand_target_boxes


def test_every_reference_flag_is_accepted():
    """flag names of reference train.py:399-478 / evaluate.py:384-466"""
    train_flags = ["--dataset", "--data_path", "--aspect-ratio-group-factor", "--use_stored_psfs", "--stored_psf_directory", "-j",
                   "--workers", "--model", "--trainable_backbone_blocks", "--pretrained", "--device", "-b", "--batch_size", "--lr",
                   "--lr-step-size", "--lr-steps", "--lr-gamma", "--epochs", "--momentum", "--weight_decay", "--resume",
                   "--start_from_weights", "--start_epoch", "--early_stop", "--eval_first", "--tensorboard_path", "--output_dir",
                   "--image_output_dir", "--print_freq", "--blur_train", "--cpu_blur", "--gpu_blur", "--param_index",
                   "--high_exposure", "--low_exposure", "--expand_target_boxes", "--dont_center_psf", "--add_noise", "--noise_level",
                   "--add_block", "--add_jpeg_artefacts", "--warp_in_model", "--deblur_first", "--deblurer_model_location",
                   "--non_pos_aug_mix", "--include_pos_aug_mix", "--aug_mix_target_expand", "--use_custom_image_norm",
                   "--unfrozen_batch_norm", "--world-size", "--dist-url"]
    eval_flags = ["--dataset", "--data_path", "--use_stored_psfs", "--stored_psf_directory", "-j", "--workers", "--blurred_dataset",
                  "--expand_synth_boxes", "--model", "--trainable_backbone_blocks", "--pretrained", "--resume", "--use_ensemble",
                  "--ensemble_model_paths", "--blur_estimator_path", "--vanilla_eval", "--early_stop", "--device", "--tensorboard_path",
                  "--output_dir", "--image_output_dir", "--blur_eval", "--cpu_blur", "--gpu_blur", "--param_index", "--high_exposure",
                  "--low_exposure", "--LEHE", "--expand_target_boxes", "--dont_center_psf", "--add_noise", "--noise_level", "--add_block",
                  "--add_jpeg_artefacts", "--dilate_psf", "--warp_in_model", "--deblur_first", "--deblurer_model_location",
                  "--non_pos_aug_mix", "--include_pos_aug_mix", "--aug_mix_target_expand", "--use_custom_image_norm",
                  "--unfrozen_batch_norm", "--mode_one_norm", "--world-size", "--dist-url"]
    for parser, flags in ((TR.build_parser(), train_flags), (EV.build_parser(), eval_flags)):
        known = {s for a in parser._actions for s in a.option_strings}
        assert not [f for f in flags if f not in known]
    # defaults the reference sets
    d = TR.build_parser().parse_args([])
    assert (d.lr, d.lr_steps, d.lr_gamma, d.epochs, d.batch_size, d.momentum, d.weight_decay, d.print_freq, d.aspect_ratio_group_factor,
            d.trainable_backbone_blocks, d.noise_level) == (0.04, [16, 22], 0.1, 37, 8, 0.9, 1e-4, 20, 3, 3, 0.001)


# ---- a COCO tree written by the test itself ------------------------------------------------------------------

def _write_coco(root, split="val"):
    from PIL import Image
    os.makedirs(os.path.join(root, split + "2017"))
    os.makedirs(os.path.join(root, "annotations"), exist_ok=True)
    rs = np.random.RandomState(3)
    images, anns = [], []
    sizes = {11: (96, 120), 5: (110, 100), 42: (90, 130), 77: (100, 100)}          # id -> (H, W)
    for img_id, (h, w) in sizes.items():
        name = "%012d.jpg" % img_id
        Image.fromarray(rs.randint(0, 255, (h, w, 3), dtype=np.uint8)).save(os.path.join(root, split + "2017", name))
        images.append({"id": img_id, "file_name": name, "height": h, "width": w})
    k = 1
    def add(img, bbox, cat, crowd=0):
        nonlocal k
        x, y, bw, bh = bbox          # outline: the box itself as a four-point polygon; the crowd object as an uncompressed RLE; one none
        seg = {"size": list(sizes[img]), "counts": [30, 50, sizes[img][0] * sizes[img][1] - 80]} if crowd else \
            ([] if cat == 90 else [[x, y, x + bw, y, x + bw, y + bh, x, y + bh]])
        anns.append({"id": k, "image_id": img, "bbox": bbox, "category_id": cat, "area": bbox[2] * bbox[3], "iscrowd": crowd,
                     "segmentation": seg})
        k += 1
    add(11, [10.5, 20.25, 40.0, 30.0], 3)
    add(11, [100.0, 60.0, 50.0, 50.0], 18)          # sticks out of the 120 x 96 image: clamped
    add(11, [5.0, 5.0, 20.0, 20.0], 1, crowd=1)     # crowd: dropped from the target, kept in the ground truth
    add(11, [30.0, 30.0, 0.0, 10.0], 7)             # zero width: dropped by `keep`, its area / iscrowd entries stay
    add(5, [1.0, 2.0, 30.0, 40.0], 90)
    add(42, [0.0, 0.0, 1.0, 1.0], 2)                # only a (close to) empty box: image dropped from training
    # image 77 has no annotation at all
    cats = [{"id": c, "name": str(c)} for c in (1, 2, 3, 7, 18, 90)]
    with open(os.path.join(root, "annotations", "instances_%s2017.json" % split), "w") as f:
        json.dump({"images": images, "annotations": anns, "categories": cats}, f)
    return sizes


def test_coco_detection_reads_a_real_tree(tmp_path):
    root = str(tmp_path)
    _write_coco(root, "val")
    _write_coco(root, "train")
    ds, n = coco_utils.get_coco(root, "val", TR.get_transform(False))
    assert n == 91 and len(ds) == 4 and ds.ids == [5, 11, 42, 77]
    img, tgt, bd = ds[1]                                                     # image 11
    assert img.dtype == torch.float32 and tuple(img.shape) == (3, 96, 120) and bd == {"epoch_number": None, "dryRun": False}
    assert tgt["image_id"].tolist() == [11]
    assert torch.equal(tgt["boxes"], torch.tensor([[10.5, 20.25, 50.5, 50.25], [100.0, 60.0, 120.0, 96.0]]))
    assert tgt["labels"].tolist() == [3, 18] and tgt["labels"].dtype == torch.int64
    assert tgt["iscrowd"].tolist() == [0, 0, 0] and tgt["area"].tolist() == [1200.0, 2500.0, 0.0]     # not filtered by `keep`
    # masks of the two kept objects (reference coco_utils.py:34-49), flipped with the image when the flip transform fires
    assert tgt["masks"].dtype == torch.uint8 and tuple(tgt["masks"].shape) == (2, 96, 120)
    m0 = tgt["masks"][0].numpy()
    ys, xs = np.nonzero(m0)
    assert (xs.min(), xs.max(), ys.min(), ys.max()) == (11, 50, 20, 49) and m0.sum() == 40 * 30      # the polygon (10.5, 20.25) .. (50.5, 50.25) as pycocotools fills it
    assert tgt["masks"][1][60:, 100:].all() and tgt["masks"][1].sum() == 20 * 36                      # clipped by the image
    _, t5, _ = ds[0]                                                         # image 5: an object without an outline
    assert tuple(t5["masks"].shape) == (1, 110, 100) and int(t5["masks"].sum()) == 0
    img, tgt, _ = ds[3]                                                      # image 77: nothing annotated
    assert tuple(tgt["boxes"].shape) == (0, 4) and tgt["labels"].numel() == 0
    # training drops 42 (only an empty box) and 77 (nothing)
    tr, _ = coco_utils.get_coco(root, "train", TR.get_transform(True))
    assert isinstance(tr, torch.utils.data.Subset) and [tr.dataset.ids[i] for i in tr.indices] == [5, 11]
    # ground truth for the evaluator = the annotation file, crowd entries included
    gt = coco_utils.get_coco_api_from_dataset(tr)
    assert [a["id"] for a in gt.imgToAnns[11]] == [1, 2, 3, 4] and gt.getCatIds() == [1, 2, 3, 7, 18, 90]
    with pytest.raises(FileNotFoundError):
        coco_utils.get_coco(str(tmp_path / "nowhere"), "val", None)
    with pytest.raises(RuntimeError, match="--synthetic"):
        coco_utils.get_coco(None, "val", None)


def test_aspect_ratio_groups(tmp_path):
    from detectinblur_amd.group_by_aspect_ratio import GroupedBatchSampler, create_aspect_ratio_groups
    _write_coco(str(tmp_path), "val")
    ds, _ = coco_utils.get_coco(str(tmp_path), "val", None)
    groups = create_aspect_ratio_groups(ds, k=3)                 # ratios 100/110, 120/96, 130/90, 1.0
    bins = (2 ** np.linspace(-1, 1, 7)).tolist()
    assert groups == [int(np.searchsorted(bins, r, side="right")) for r in (100 / 110, 120 / 96, 130 / 90, 1.0)]
    ids = [0, 0, 1, 0, 1, 1, 0, 2, 0]
    s = GroupedBatchSampler(torch.utils.data.SequentialSampler(range(9)), ids, 2)
    batches = list(s)
    assert len(batches) == len(s) == 4 and all(len({ids[i] for i in b}) == 1 and len(b) == 2 for b in batches)
    assert batches[:3] == [[0, 1], [2, 4], [3, 6]]


def test_evaluate_returns_the_coco_evaluator_surface(tmp_path):
    from detectinblur_amd.engine import evaluate
    from tests.test_engine_ddp_cpu import _small_model
    root = str(tmp_path)
    _write_coco(root, "val")
    ds, _ = coco_utils.get_coco(root, "val", TR.get_transform(False))
    loader = torch.utils.data.DataLoader(ds, batch_size=1, collate_fn=utils.collate_fn)
    ce = evaluate(_small_model(), loader, torch.device("cpu"), vanilla_eval=True)
    stats = ce.coco_eval["bbox"].stats                                        # reference train.py:350-358
    assert stats.shape == (12,) and np.all((stats >= -1) & (stats <= 1))
    assert sorted(ce.img_ids) == [5, 11, 42, 77] and len(ce["detections"]) == 4 and ce.coco_stats is stats
    assert ce.coco_gt is not ds.coco and ce.coco_gt.imgToAnns[11][0]["bbox"] == [10.5, 20.25, 40.0, 30.0]
    # perfect detections score AP = AR = 1 through the same object
    from detectinblur_amd.coco_eval import CocoEvaluator
    ev = CocoEvaluator(ds.coco, ["bbox"])
    for img in (5, 42):          # (image 11 carries a zero-width box, which COCO's IoU can never match)
        anns = [a for a in ds.coco.imgToAnns[img] if not a["iscrowd"]]
        b = torch.tensor([a["bbox"] for a in anns], dtype=torch.float64)
        b[:, 2:] += b[:, :2]
        ev.update({img: {"boxes": b, "labels": torch.tensor([a["category_id"] for a in anns]), "scores": torch.ones(len(anns))}})
    ev.synchronize_between_processes()
    ev.accumulate()
    s = ev.summarize()
    assert abs(s[0] - 1) < 1e-12 and abs(s[1] - 1) < 1e-12 and s[8] == 1.0      # precision = tp / (tp + fp + eps), as COCOeval
    with pytest.raises(NotImplementedError):
        CocoEvaluator(ds.coco, ["bbox", "segm"])


def test_expanded_boxes_replace_the_ground_truth(golden):
    """wrapper == CocoBoxEvaluator on the reference-pinned synthetic case, with the ground truth handed over as a
    COCO annotation set and edited in place before scoring (reference engine.py:325-342)"""
    import gen_goldens as GG
    from detectinblur_amd.coco_eval import CocoEvaluator
    gt, dt = GG.coco_eval_inputs()
    anns, k = [], 1
    for img, t in gt.items():
        b = np.asarray(t["boxes"], dtype=np.float64)
        for j in range(b.shape[0]):
            area = float(t["area"][j]) if "area" in t else float((b[j, 2] - b[j, 0]) * (b[j, 3] - b[j, 1]))
            anns.append({"id": k, "image_id": int(img), "bbox": [0.0, 0.0, 1.0, 1.0], "category_id": int(t["labels"][j]),
                         "area": area, "iscrowd": int(t["iscrowd"][j]) if "iscrowd" in t else 0})
            k += 1
    coco = coco_utils.CocoGT({"images": [{"id": int(i)} for i in gt], "annotations": anns, "categories": []})
    ev = CocoEvaluator(coco, ["bbox"])
    for img in gt:                                   # the boxes arrive late, image by image, as in evaluate()
        b = np.asarray(gt[img]["boxes"], dtype=np.float64)
        for j, a in enumerate(ev.coco_gt.imgToAnns[int(img)]):
            a["bbox"] = [b[j, 0], b[j, 1], b[j, 2] - b[j, 0], b[j, 3] - b[j, 1]]
        ev.update({img: dt[img]} if img in dt else {})
    ev.synchronize_between_processes()
    ev.accumulate()
    assert np.allclose(ev.summarize(), golden.coco["coco_stats"], rtol=0, atol=1e-15)
    assert coco.imgToAnns[int(next(iter(gt)))][0]["bbox"] == [0.0, 0.0, 1.0, 1.0]       # the caller's object is untouched


def test_tensorboard_event_file(tmp_path):
    from detectinblur_amd import tb_writer
    assert tb_writer.crc32c(b"123456789") == 0xE3069283                       # CRC-32C check value
    w = tb_writer.make_writer(str(tmp_path / "tb"))
    w.add_scalar("Blurred/Accuracies", 0.25, 3)
    w.add_scalar("losses/overallLoss", torch.tensor(1.5), 7)
    w.close()
    files = os.listdir(str(tmp_path / "tb"))
    assert len(files) == 1 and files[0].startswith("events.out.tfevents.")
    if isinstance(w, tb_writer.EventFileWriter):
        assert tb_writer.read_scalars(w.path) == [("Blurred/Accuracies", 3, 0.25), ("losses/overallLoss", 7, 1.5)]
        first = open(w.path, "rb").read()[12:]
        assert b"brain.Event:2" in first[:40]                                # the version record TensorBoard looks for


def test_train_cli_checkpoint_resume_and_tensorboard(tmp_path, monkeypatch):
    """two epochs through train.main on CPU (blur off: no CPU blur path exists), then --resume: weights, momentum
    buffers and schedule come back identical and training continues at the next epoch (reference train.py:251-257,
    :330-340); TensorBoard carries the per-epoch statistics under the reference's tags."""
    from detectinblur_amd import tb_writer
    import detectinblur_amd.train as train_mod
    from tests.test_engine_ddp_cpu import _small_model
    monkeypatch.setattr(train_mod, "fasterrcnn_resnet50_fpn", lambda **kw: _small_model())
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    out, tb = str(tmp_path / "w"), str(tmp_path / "tb")
    base = ["--synthetic", "--synthetic_images", "4", "--synthetic_size", "90", "120", "--device", "cpu", "-b", "2", "--lr", "0.001",
            "--lr-steps", "1", "--output_dir", out, "--tensorboard_path", tb, "--print_freq", "1", "--aspect-ratio-group-factor", "3"]
    train_mod.main(train_mod.build_parser().parse_args(base + ["--epochs", "2"]))
    ck0 = torch.load(os.path.join(out, "model_0.pth"), map_location="cpu", weights_only=False)
    ck1 = torch.load(os.path.join(out, "model_1.pth"), map_location="cpu", weights_only=False)
    assert set(ck1) == {"model", "optimizer", "lr_scheduler", "args", "epoch"} and ck1["epoch"] == 1
    assert ck1["lr_scheduler"]["last_epoch"] == 2 and ck1["optimizer"]["param_groups"][0]["lr"] == pytest.approx(1e-4)
    tags = {t for t, _, _ in tb_writer.read_scalars(os.path.join(tb, os.listdir(tb)[0]))}
    assert {"Normal/AccuraciesSweep", "Normal/recall", "Blurred/Accuracies", "Blurred/recallLarge", "losses/overallLoss"} <= tags

    # resume from epoch 0's checkpoint with epochs=1: nothing left to train, state is exactly what was saved
    seen = {}
    real_sgd = torch.optim.SGD

    class SpySGD(real_sgd):
        def load_state_dict(self, sd):
            super().load_state_dict(sd)
            seen["opt"] = self
    monkeypatch.setattr(torch.optim, "SGD", SpySGD)
    real_eval = train_mod.evaluate
    def spy_eval(model, *a, **k):
        seen["model"] = model
        return real_eval(model, *a, **k)
    monkeypatch.setattr(train_mod, "evaluate", spy_eval)
    args = train_mod.build_parser().parse_args(base + ["--epochs", "2", "--resume", os.path.join(out, "model_0.pth"), "--output_dir",
                                                       str(tmp_path / "w2"), "--tensorboard_path", str(tmp_path / "tb2")])
    torch.manual_seed(99)
    train_mod.main(args)
    assert args.start_epoch == 1
    assert os.listdir(str(tmp_path / "w2")) == ["model_1.pth"]                # continued at epoch 1, one epoch run
    # --start_from_weights: weights only
    m = _small_model()
    m.load_state_dict(ck0["model"])
    m2 = _small_model()
    monkeypatch.setattr(train_mod, "fasterrcnn_resnet50_fpn", lambda **kw: m2)
    train_mod.main(train_mod.build_parser().parse_args(base + ["--epochs", "0", "--start_from_weights", os.path.join(out, "model_0.pth"),
                                                               "--tensorboard_path", str(tmp_path / "tb3")]))
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    # and the optimizer state of --resume equals the saved one
    opt_sd = seen["opt"].state_dict()
    for pid, st in ck0["optimizer"]["state"].items():
        assert "momentum_buffer" in st
    assert len(opt_sd["state"]) == len(ck0["optimizer"]["state"])


def test_loader_workers_come_from_the_fork_server():
    """`utils.loader_context`: DataLoader workers are forked from a server process started before this process touches a
    GPU (a fork of a GPU process stalls the GPU for tens of seconds per loader: DESIGN.md section 6); the dataset, the
    `BlurImage` transform, the collate function and the worker seeding all travel to such workers."""
    import torch
    from detectinblur_amd import utils
    from detectinblur_amd.coco_utils import SyntheticCocoDetection
    from detectinblur_amd.train import _seed_worker, get_transform
    ctx = utils.loader_context()
    if torch.cuda.is_initialized() or os.environ.get("DIB_LOADER_FORK"):
        assert ctx is None
        return
    assert ctx is not None and ctx.get_start_method() == "forkserver" and utils.loader_context() is ctx
    ds = SyntheticCocoDetection(num_images=6, size=(70, 90), boxes_per_image=3,
                                transforms=get_transform(True, blur=True, blur_type=0.005, blur_ratio=1.0, low_exposure=True))
    loader = torch.utils.data.DataLoader(ds, batch_size=2, num_workers=2, collate_fn=utils.collate_fn, worker_init_fn=_seed_worker,
                                         multiprocessing_context=ctx)
    seen = 0
    for images, targets, blur_dicts in loader:
        assert len(images) == 2 and images[0].shape == (3, 70, 90)
        assert all(bd["blurring"] and bd["psf"].shape == (128, 128) for bd in blur_dicts)
        seen += len(images)
    assert seen == 6
