"""Bodies of the GPU tests that need a process of their own (an RCCL process group has to be created by a process that has not
been doing other GPU work, and a GPU process must not exec): run through the session's fork server
(detectinblur_amd.utils.loader_context, started before pytest touches the GPU), they write their result as JSON to `out_path`
(or the traceback to `out_path + ".err"`)."""
import json
import os
import socket
import sys
import traceback


def _free_port():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _one_rank_env():
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      LOCAL_WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def _guarded(fn, out_path, args):
    try:
        res = fn(*args)
        with open(out_path, "w") as f:
            json.dump(res, f)
    except BaseException:       # noqa: BLE001 -- the parent reads the traceback from the file
        with open(out_path + ".err", "w") as f:
            f.write(traceback.format_exc())
        raise


def _ddp_one_rank(fused):
    """DDP(model, gradient_as_bucket_view=True, broadcast_buffers=False) over a ONE-RANK RCCL group against the bare model (same
    weights, same inputs, same sampler draws), two consecutive steps with momentum SGD: per step the worst relative error of
    any gradient tensor, the run-to-run noise of the bare model itself (a second bare copy), and how far step 1's gradients
    are from step 0's (the comparison's sensitivity)."""
    _one_rank_env()
    import copy
    import tempfile
    os.environ["MIOPEN_USER_DB_PATH"] = tempfile.mkdtemp(prefix="dib_ddp_miopen_")     # keep deterministic-mode choices out of the shipped db
    import numpy as np
    import torch
    import torch.distributed as dist
    # MIOpen's default kernels for this detector's small-M convolutions accumulate with atomics (profiles/r4_nondeterminism.txt): two
    # forward passes of the SAME model differ in the last bits, proposals and sampled RoIs flip, and two bare models' gradients
    # differ by per cents (measured: 3 % on conv1.weight) -- no yardstick for DDP.  With MIOpen's deterministic attribute
    # (slow reference kernels; fine at 320 x 480) forward decisions are reproducible and what is left is ~1e-6 of atomics in
    # backward kernels.
    torch.backends.cudnn.deterministic = True
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method="env://", device_id=dev)
    from detectinblur_amd.models import backbone as B
    from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
    if not fused:
        B.FUSE_EPILOGUE = B.BLOCK_ENTRY = False
    torch.manual_seed(0)
    model = fasterrcnn_resnet50_fpn(pretrained=False, pretrained_backbone=False, num_classes=91, min_size=320, max_size=480).to(dev)
    bare, bare2 = copy.deepcopy(model), copy.deepcopy(model)
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], broadcast_buffers=False, gradient_as_bucket_view=True)
    nets = {"ddp": ddp, "bare": bare, "bare2": bare2}
    cores = {"ddp": model, "bare": bare, "bare2": bare2}
    opts = {k: torch.optim.SGD([p for p in cores[k].parameters() if p.requires_grad], lr=0.002, momentum=0.9) for k in nets}
    for n in nets.values():
        n.train()
    means, stds = np.tile([0.485, 0.456, 0.406], (2, 1)), np.tile([0.229, 0.224, 0.225], (2, 1))
    g = torch.Generator().manual_seed(7)
    steps, prev = [], None
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    for step in range(2):
        imgs = [torch.rand(3, 320, 480, generator=g).to(dev), torch.rand(3, 300, 440, generator=g).to(dev)]
        tg = [{"boxes": torch.tensor([[30., 40., 200., 260.], [100., 20., 300., 180.]], device=dev), "labels": torch.tensor([3, 17], device=dev)},
              {"boxes": torch.tensor([[10., 50., 150., 290.]], device=dev), "labels": torch.tensor([44], device=dev)}]
        grads, losses = {}, {}
        for k in ("ddp", "bare", "bare2"):
            torch.manual_seed(100 + step)
            loss = sum(nets[k]([i.clone() for i in imgs], [{a: b.clone() for a, b in t.items()} for t in tg], newMeans=means, newSTDs=stds).values())
            opts[k].zero_grad()
            loss.backward()
            grads[k] = [p.grad.detach().clone() for p in cores[k].parameters() if p.requires_grad]
            losses[k] = float(loss.detach())
        rel = lambda a, b: float((a - b).norm()) / (float(b.norm()) + 1e-12)      # noqa: E731
        err = [rel(a, b) for a, b in zip(grads["ddp"], grads["bare"])]
        noise = [rel(a, b) for a, b in zip(grads["bare2"], grads["bare"])]
        rec = {"step": step, "loss": losses, "ddp_vs_bare_max": max(err), "ddp_vs_bare_worst": names[int(np.argmax(err))],
               "bare_vs_bare_max": max(noise), "bare_vs_bare_worst": names[int(np.argmax(noise))], "tensors": len(err),
               "all_finite": all(bool(torch.isfinite(x).all()) for x in grads["ddp"])}
        if prev is not None:
            rec["vs_previous_step_min"] = min(rel(a, b) for a, b in zip(grads["ddp"], prev))
        prev = grads["ddp"]
        steps.append(rec)
        for k in nets:
            opts[k].step()
    wdiff = max(float((a - b).norm()) / (float(b.norm()) + 1e-12) for a, b in zip(model.parameters(), bare.parameters()))
    dist.barrier()
    dist.destroy_process_group()
    return {"fused": fused, "steps": steps, "weights_ddp_vs_bare_max": wdiff, "device": torch.cuda.get_device_name(0)}


def _bench_forced_dist(argv):
    """bench.py's N > 1 branch with ONE rank (DIB_BENCH_FORCE_DIST=1): RCCL process group, collectives around every timed block,
    the detector under the DDP wrapper, distributed engine mode, the sharded sweep with its merges.  Returns the JSON line."""
    _one_rank_env()
    os.environ["DIB_BENCH_FORCE_DIST"] = "1"
    import contextlib
    import io
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    sys.argv = ["bench.py"] + list(argv)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    lines = [l for l in buf.getvalue().splitlines() if l.startswith("{")]
    assert len(lines) == 1, buf.getvalue()[-2000:]
    return json.loads(lines[0])


def _evaluate_main_digest(argv):
    """`python -m detectinblur_amd.evaluate <argv>` in this (fresh) process: per sweep cell the twelve COCO statistics and a
    SHA-256 over every detection (image id, boxes, scores, labels) and every expanded ground-truth box, bit for bit."""
    import contextlib
    import hashlib
    import io
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from detectinblur_amd import evaluate
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res = evaluate.main(evaluate.build_parser().parse_args(list(argv)))
    out = {}
    for cell in sorted(res):
        r = res[cell]
        h = hashlib.sha256()
        n = 0
        for k in sorted(r["detections"]):
            h.update(np.int64(k).tobytes())
            for f in ("boxes", "scores", "labels"):
                h.update(np.ascontiguousarray(r["detections"][k][f].numpy()).tobytes())
            h.update(np.ascontiguousarray(r["targets"][k].numpy()).tobytes())
            n += len(r["detections"][k]["boxes"])
        out[cell] = {"stats": [float(x) for x in r.coco_eval["bbox"].stats], "sha256": h.hexdigest(), "images": len(r["detections"]), "boxes": n}
    lines = [l for l in buf.getvalue().splitlines() if "Average Precision" in l or "Average Recall" in l]
    # what the kernel choice of this process rested on: the shipped find-db / TunableOp data (evaluate.main installs them) and
    # whatever MIOpen appended to its private copy for shapes the data does not hold -- records of a find step run HERE, which a
    # second process may rank differently
    from detectinblur_amd import kernel_choices
    rep = kernel_choices.report()
    h = hashlib.sha256()
    d = rep.get("miopen_user_db")
    if d and os.path.isdir(d):
        for f in sorted(os.listdir(d)):
            if f.endswith(".lock") or f.endswith(".time") or not os.path.isfile(os.path.join(d, f)):
                continue
            with open(os.path.join(d, f), "rb") as fh:
                h.update(f.encode() + b"\0" + b"\n".join(sorted(fh.read().splitlines())))
    choice = {k: rep.get(k) for k in ("installed", "miopen_foreign_files", "miopen_db_growth_bytes", "tunableop_validators_match",
                                      "tunableop_entries_loaded", "tunableop_shipped_entries")}
    choice["miopen_db_sha256"] = h.hexdigest()
    return {"cells": out, "stat_lines": lines, "kernel_choice": choice}


def evaluate_main_digest(out_path, argv):
    _guarded(_evaluate_main_digest, out_path, (argv,))


def _driver_under_one_rank_rccl(which, argv):
    """`train.main` / `evaluate.main` in distributed mode over a ONE-RANK RCCL process group (RANK / WORLD_SIZE / LOCAL_RANK set as
    torchrun sets them): utils.init_distributed_mode picks "nccl", the model is wrapped in DistributedDataParallel, the samplers
    are DistributedSamplers, metrics and COCO records go through the RCCL collectives of utils.all_gather / reduce_dict."""
    _one_rank_env()
    import contextlib
    import io
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch
    import torch.distributed as dist
    buf = io.StringIO()
    out = {}
    with contextlib.redirect_stdout(buf):
        if which == "train":
            from detectinblur_amd import train
            train.main(train.build_parser().parse_args(list(argv)))
        else:
            from detectinblur_amd import evaluate
            res = evaluate.main(evaluate.build_parser().parse_args(list(argv)))
            out["cells"] = {k: {"stats": [float(x) for x in v.coco_eval["bbox"].stats], "images": len(v["detections"])} for k, v in res.items()}
    out["backend"] = dist.get_backend() if dist.is_initialized() else None
    out["world"] = dist.get_world_size() if dist.is_initialized() else None
    text = buf.getvalue()
    out["distributed_line"] = "| distributed init (rank 0)" in text
    out["loss_lines"] = len([l for l in text.splitlines() if "loss_classifier" in l])
    out["stat_lines"] = len([l for l in text.splitlines() if "Average Precision" in l])
    out["tail"] = text[-600:]
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return out


def driver_under_one_rank_rccl(out_path, which, argv):
    _guarded(_driver_under_one_rank_rccl, out_path, (which, argv))


def ddp_one_rank(out_path, fused):
    _guarded(_ddp_one_rank, out_path, (fused,))


def bench_forced_dist(out_path, argv):
    _guarded(_bench_forced_dist, out_path, (argv,))
