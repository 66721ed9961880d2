"""bench.py's launch contract: `--gpus N` really runs N ranks, and a rank count that
was not run is never reported.  CPU only: `--dry-run-gloo` counts the ranks over gloo and skips the blur."""
import json
import os
import subprocess
import sys


def _free_port():
    """a port nobody listens on right now: two test sessions on one machine must not meet on a fixed one"""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=600)


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out            # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_gpus_2_launches_two_ranks():
    r = _run(["--gpus", "2", "--dry-run-gloo"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["world_size_seen_by_backend"] == 2 and d["gpus_flag"] == 2
    assert sorted(x["rank"] for x in d["ranks"]) == [0, 1]
    assert len({x["pid"] for x in d["ranks"]}) == 2       # two processes, not one process counted twice
    assert os.getpid() not in {x["pid"] for x in d["ranks"]}


def test_gpus_1_stays_in_process():
    d = _json_line(_run(["--gpus", "1", "--dry-run-gloo"]).stdout)
    assert d["n_gpus"] == 1 and len(d["ranks"]) == 1


def test_world_size_mismatch_is_refused():
    r = _run(["--gpus", "8", "--dry-run-gloo"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_under_torchrun_the_driver_form():
    """the driver's own N>1 command line: torch.distributed.run ... bench.py --gpus N"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run-gloo"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout)["n_gpus"] == 2
