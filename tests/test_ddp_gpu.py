"""What one GPU can show of the N > 1 path (reference train.py:186-205, 238-241; utils.py:579-603; engine.py:151-153, 406-414):
the detector's gradients under DDP over an RCCL process group equal the bare model's, with every fused autograd node on (custom
in-place epilogues, entry nodes, bucket-view gradients) and with them off; and bench.py's whole distributed branch runs
with one rank and reports every field an 8-rank line would.  Each body runs in a process of its own (tests/_gpu_children.py)
started by the session's fork server."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu


def _run_child(target, tmp_path, *args, timeout=1500):
    from detectinblur_amd import utils
    ctx = utils.loader_context()
    if ctx is None:
        pytest.skip("no fork server (the GPU was initialised before the test session could start one)")
    out = str(tmp_path / "child.json")
    p = ctx.Process(target=target, args=(out,) + args)
    p.start()
    p.join(timeout)
    if p.is_alive():
        p.kill()
        p.join()
        pytest.fail("child timed out")
    if os.path.exists(out + ".err"):
        pytest.fail(open(out + ".err").read()[-4000:])
    assert p.exitcode == 0 and os.path.exists(out), p.exitcode
    return json.load(open(out))


@pytest.mark.parametrize("fused", [True, False], ids=["fused_nodes_on", "plain_autograd_graph"])
def test_ddp_gradients_equal_the_bare_model_over_one_rank_rccl(tmp_path, fused):
    """Two consecutive momentum-SGD steps.  The DDP-wrapped model and the bare model see the same batch and the same sampler
    draws; with one rank the all-reduced mean IS the local gradient, so every gradient tensor must agree -- to 1e-5 of its norm in
    the first step (MIOpen in deterministic mode: see tests/_gpu_children.py), to the ReLU-flip bound in the second; a second
    bare copy gives the yardstick (what two runs WITHOUT DDP differ by); the second step's gradients are far (> 1e-2) from the
    first's, so the comparison can fail."""
    from tests import _gpu_children
    r = _run_child(_gpu_children.ddp_one_rank, tmp_path, fused)
    print(json.dumps(r))
    assert r["fused"] is fused and len(r["steps"]) == 2
    for s in r["steps"]:
        assert s["all_finite"] and s["tensors"] > 80
        # step 0 (identical weights, deterministic forward): 1e-5 of every tensor's norm (measured 1.5e-6, the same as two bare
        # runs).  step 1: the copies' weights differ by ~1e-6 after the first update, a few ReLU masks flip (measured
        # 5e-4..2e-3, again the same between two bare runs): 5e-3, or three times the bare models' own spread
        bound = 1e-5 if s["step"] == 0 else max(5e-3, 3.0 * s["bare_vs_bare_max"])
        assert s["ddp_vs_bare_max"] <= bound, s
        assert s["bare_vs_bare_max"] <= (1e-5 if s["step"] == 0 else 2e-2), s         # the yardstick itself
        assert abs(s["loss"]["ddp"] - s["loss"]["bare"]) <= 1e-5 * abs(s["loss"]["bare"]), s
    assert r["steps"][1]["vs_previous_step_min"] > 1e-2
    assert r["weights_ddp_vs_bare_max"] <= 1e-4


def test_bench_distributed_branch_with_one_rank(tmp_path):
    """`DIB_BENCH_FORCE_DIST=1`: the line a 1-rank run of bench.py's N > 1 code prints carries what an 8-rank line would --
    RCCL's own rank count, per-rank step times, the DDP train step with its RCCL time and overlap fields, the loader-fed
    epoch in distributed mode, the sharded sweep with the merge inside the clock."""
    from tests import _gpu_children
    argv = ["--gpus", "1", "--steps", "5", "--warmup", "5", "--repeats", "3", "--train-steps", "2", "--train-warmup", "2", "--e2e-steps", "2",
            "--e2e-warmup", "1", "--sweep-images", "2", "--no-cpu-baseline", "--cold-sets", "0", "--graph-steps", "4"]
    d = _run_child(_gpu_children.bench_forced_dist, tmp_path, argv, timeout=2400)
    assert d["n_gpus"] == 1 and d["world_size_seen_by_rccl"] == 1 and d["rank_devices"] == [0]
    assert d["value"] > 1e4 and len(d["ms_per_step_by_rank"]) == 1 and d["ms_per_step_rank_min_max"][0] > 0
    assert d["roofline"]["frac"] > 0.1 and d["graph"]["ms_per_step"] is not None
    t = d["train_step"]
    assert t["parallelism"] == "ddp1" and t["loss_finite"] and t["value"] > 10
    for k in ("rccl_ms_per_step", "rccl_overlapped_frac", "rccl_kernels_per_step", "conv_kernel_ms_per_step", "conv_frac_of_fp32_mfma"):
        assert k in t, k
    assert d["train_e2e"]["value"] > 5
    sw = d["eval_sweep"]
    assert sw["n_gpus"] == 1 and sw["sharding"].startswith("DistributedSampler") and len(sw["cells"]) == 15
    assert set(sw["images_per_s_with_merge"]) == set(sw["images_per_s"]) and all(v > 0 for v in sw["images_per_s_with_merge"].values())


def test_train_main_and_evaluate_main_over_one_rank_rccl(tmp_path):
    """configs[3] / configs[4] as the reference launches them (`python -m torch.distributed.launch ... train.py`, reference
    train.py:186-205; evaluate.py:331-332) with the one rank a 1-GPU box has: the REAL drivers at 800 x 1333 in distributed
    mode on the RCCL backend -- DDP-wrapped detector fed by a DistributedSampler over stored PSFs, blurred and clean evaluation
    passes with their all_gather merges, checkpoint written by rank 0; then the 15-cell sweep of `evaluate.main` with its
    DistributedSampler and per-cell merges.  (Two gloo ranks run the same drivers on the CPU: tests/test_drivers_two_ranks.py.)"""
    import torch
    from tests import _gpu_children
    from tests.test_full_size_gpu import _psf_store
    store = _psf_store(tmp_path, 32)
    out_dir = tmp_path / "weights"
    (tmp_path / "t").mkdir(); (tmp_path / "e").mkdir()
    r = _run_child(_gpu_children.driver_under_one_rank_rccl, tmp_path / "t", "train", [
        "--synthetic", "--synthetic_images", "32", "--synthetic_size", "800", "1333", "-b", "8", "-j", "2", "--epochs", "1",
        "--blur_train", "--gpu_blur", "--use_stored_psfs", "--stored_psf_directory", store, "--stored_psf_count", "32",
        "--param_index", "1", "--low_exposure", "--expand_target_boxes", "--early_stop", "2", "--lr", "0.002", "--print_freq", "1",
        "--output_dir", str(out_dir)], timeout=2400)
    assert r["backend"] == "nccl" and r["world"] == 1 and r["distributed_line"], r["tail"]
    assert r["loss_lines"] >= 2 and r["stat_lines"] >= 2, r["tail"]
    ck = torch.load(out_dir / "model_0.pth", map_location="cpu", weights_only=False)
    assert not any(k.startswith("module.") for k in ck["model"]) and len(ck["model"]) == 295
    assert all(torch.isfinite(v).all() for v in ck["model"].values() if v.is_floating_point())
    e = _run_child(_gpu_children.driver_under_one_rank_rccl, tmp_path / "e", "evaluate", [
        "--synthetic", "--synthetic_images", "3", "--synthetic_size", "800", "1333", "-j", "2", "--blur_eval", "--gpu_blur",
        "--expand_target_boxes", "--early_stop", "1", "--resume", str(out_dir / "model_0.pth")], timeout=2400)
    assert e["backend"] == "nccl" and e["world"] == 1 and len(e["cells"]) == 15, e["tail"]
    assert all(c["images"] == 2 and len(c["stats"]) == 12 for c in e["cells"].values())
