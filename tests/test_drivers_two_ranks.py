"""configs[3] / configs[4]'s collective paths through the REAL drivers, two ranks over gloo on the CPU:
`python -m torch.distributed.run --nproc-per-node 2 <worker> train|evaluate <driver flags>` where the worker only
parses the flags with the driver's own parser and calls `train.main` / `evaluate.main` (plus a tap on the collate
function to see which images each rank was dealt).  What runs: `init_distributed_mode`, `DistributedSampler` +
`GroupedBatchSampler` (reference train.py:186-205), DDP (:238-241), `save_on_master` (:332-339),
`CocoEvaluator.synchronize_between_processes` + `all_gather` of the shards (reference engine.py:406-414,
utils.py:536-576).  No 8-GPU node is available to this project: this is the evidence for the N > 1 paths."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import torch
from detectinblur_amd import utils
mode, outdir, argv = sys.argv[1], sys.argv[2], sys.argv[3:]
rank = int(os.environ.get("RANK", "0"))
batches = []
_collate = utils.collate_fn
def tapped(batch):
    out = _collate(batch)
    batches.append([int(t["image_id"]) for t in out[1]])
    return out
utils.collate_fn = tapped            # the drivers read utils.collate_fn when they build their loaders
result = {}
if mode == "train":
    from detectinblur_amd import train
    train.main(train.build_parser().parse_args(argv))
else:
    from detectinblur_amd import evaluate
    res = evaluate.main(evaluate.build_parser().parse_args(argv))
    for k, v in res.items():
        result[k] = {"stats": [float(x) for x in v.coco_eval["bbox"].stats], "own_ids": sorted(int(i) for i in v["detections"]),
                     "merged_ids": [int(i) for i in v.img_ids], "routes": list(v["routes"])}
with open(os.path.join(outdir, "rank%%d.json" %% rank), "w") as f:
    json.dump({"rank": rank, "world": int(os.environ.get("WORLD_SIZE", "1")), "batches": batches, "result": result}, f)
'''

SMALL = ["--device", "cpu", "--synthetic", "--synthetic_size", "96", "128", "--min_size", "96", "--max_size", "128", "-j", "0"]


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(tmp_path, nproc, mode, flags, tag):
    out = tmp_path / tag
    out.mkdir()
    worker = tmp_path / "worker.py"
    worker.write_text(_WORKER % {"root": ROOT})
    env = dict(os.environ, OMP_NUM_THREADS="2", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % nproc, "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), str(worker), mode, str(out)] + flags
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-4000:]
    ranks = []
    for k in range(nproc):
        with open(out / ("rank%d.json" % k)) as f:
            ranks.append(json.load(f))
    return ranks, r.stdout


def test_train_main_two_ranks_gloo(tmp_path):
    """`train.main` with DDP over gloo: --blur_train --cpu_blur (the blur runs in the loaders' transform, as the
    reference's --cpu_blur does), aspect-ratio grouped batches of 2 over a DistributedSampler, one epoch, then the
    clean and the blurred evaluation pass with their all_gather merges."""
    n = 8
    ck = tmp_path / "weights"
    ranks, text = _launch(tmp_path, 2, "train", SMALL + ["--synthetic_images", str(n), "-b", "2", "--epochs", "1", "--lr", "0.001",
                                                         "--blur_train", "--cpu_blur", "--param_index", "1", "--low_exposure",
                                                         "--output_dir", str(ck), "--tensorboard_path", str(tmp_path / "tb"),
                                                         "--print_freq", "1"], "train")
    assert [r["world"] for r in ranks] == [2, 2]
    train_batches = [[b for b in r["batches"] if len(b) == 2] for r in ranks]
    eval_batches = [[b[0] for b in r["batches"] if len(b) == 1] for r in ranks]
    # training: each rank sees its own half of the DistributedSampler's permutation, the same number of batches
    assert len(train_batches[0]) == len(train_batches[1]) == n // 2 // 2
    seen = [sorted(i for b in tb for i in b) for tb in train_batches]
    assert not set(seen[0]) & set(seen[1]) and sorted(seen[0] + seen[1]) == list(range(n))
    # the two evaluation passes (clean, blurred): shards disjoint, complete, equally long
    for r in (0, 1):
        assert len(eval_batches[r]) == 2 * (n // 2)
    clean = [eb[:n // 2] for eb in eval_batches]
    assert not set(clean[0]) & set(clean[1]) and sorted(clean[0] + clean[1]) == list(range(n))
    # exactly one checkpoint, written by rank 0 (reference train.py:332-339), loadable, with the reference's keys
    assert sorted(os.listdir(ck)) == ["model_0.pth"]
    import torch
    state = torch.load(ck / "model_0.pth", map_location="cpu", weights_only=False)
    assert sorted(state) == ["args", "epoch", "lr_scheduler", "model", "optimizer"] and state["epoch"] == 0
    assert not any(k.startswith("module.") for k in state["model"])            # model_without_ddp's state dict
    # rank-0-only printing (reference utils.py:719-731): the epoch header appears once
    assert text.count("Epoch: [0]") >= 1 and "Training time" in text
    assert any(f.startswith("events.out.tfevents") for f in os.listdir(tmp_path / "tb"))


@pytest.mark.parametrize("n", [5, 1])
def test_evaluate_main_two_ranks_equals_one_rank(tmp_path, n):
    """`evaluate.main --vanilla_eval` on an odd image count (the sampler pads: one image is scored twice and merged
    by id) and on a single image (rank 1's shard is only padding): nothing hangs, every rank ends up with every
    image, and the merged COCO statistics equal those of a one-rank run of the same command."""
    flags = SMALL + ["--synthetic_images", str(n), "--vanilla_eval", "--tensorboard_path", ""]
    one, _ = _launch(tmp_path, 1, "evaluate", flags, "one")
    two, _ = _launch(tmp_path, 2, "evaluate", flags, "two")
    ref = one[0]["result"]["Clean"]
    assert ref["merged_ids"] == list(range(n)) and ref["own_ids"] == list(range(n))
    own = [r["result"]["Clean"]["own_ids"] for r in two]
    assert sorted(set(own[0]) | set(own[1])) == list(range(n))
    assert len(own[0]) + len(own[1]) == n + (n % 2)                 # one padded duplicate when n is odd
    for r in two:
        got = r["result"]["Clean"]
        assert got["merged_ids"] == list(range(n))
        assert got["stats"] == ref["stats"]
    assert len(ref["stats"]) == 12


def test_evaluate_sweep_two_ranks_ensemble(tmp_path):
    """The 15-cell sweep of configs[4] (`--use_ensemble`, oracle routing, `--cpu_blur` because this host has no GPU),
    2 ranks, 1 image per cell and rank (`--early_stop 0`): all cells complete and merge.  (No --expand_target_boxes
    here: box growth is a HIP kernel and this package has no CPU path; the GPU tests run it.)"""
    flags = SMALL + ["--synthetic_images", "4", "--use_ensemble", "--blur_eval", "--cpu_blur", "--early_stop", "0",
                     "--tensorboard_path", str(tmp_path / "tb")]
    two, text = _launch(tmp_path, 2, "evaluate", flags, "sweep")
    cells = ["P%dE%d" % (p, e) for p in (1, 2, 3) for e in range(5)]
    for r in two:
        assert sorted(r["result"]) == sorted(cells)
    for c in cells:
        a, b = two[0]["result"][c], two[1]["result"][c]
        assert len(a["own_ids"]) == 1 and len(b["own_ids"]) == 1 and not set(a["own_ids"]) & set(b["own_ids"])
        assert a["merged_ids"] == b["merged_ids"] == sorted(a["own_ids"] + b["own_ids"])
        assert a["stats"] == b["stats"] and len(a["stats"]) == 12
        # oracle routing: blur type P -> net P for every exposure but the shortest of the sweep (1/25 bins to
        # fraction_index 0, not -1: reference transforms.py:441-446), so net index == P
        assert set(a["routes"]) == {int(c[1])}
    assert text.count("##### P") >= 15
