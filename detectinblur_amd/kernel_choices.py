"""Shipped kernel-choice data for the detector's library kernels (MIOpen convolutions, hipBLASLt / rocBLAS GEMMs), opt-in.

`use_shipped_kernel_choices()` is what the drivers (train.py, evaluate.py, train_blur_estimator.py: the place of the
reference's `main`, train.py:89) and bench.py call before their first convolution.  Importing the package does NOT touch the
process environment any more: a drop-in library has no business exporting MIOPEN_* / PYTORCH_TUNABLEOP_* behind its caller's
back (round-4 review).  `report()` says what the running stack actually did with the data.

MIOpen picks each convolution's kernel by timing every applicable solver the first time it sees a shape (~90 s for this detector
at b = 8 x 800 x 1344 on a fresh machine) and remembers the result in a "user find-db"; next to it sits the user PERF-db
(`*.udb.txt`) of a tuning run (MIOPEN_FIND_ENFORCE=SEARCH over the train step's convolutions: 100.0 -> 94.3 ms per step).
`miopen_db/` ships both for every shape the bench, the drivers and the GPU tests meet on gfx950, for the ~50 padded batch-1 input
shapes real COCO images reach in evaluation (`--fill --shapes coco-eval`, below) and for the padded b = 8 batch shapes of COCO training
(800 x 800..1344 and 800..1344 x 800 in steps of 32, a few more: `--fill --shapes coco-train`) -- a shape without a record costs its
first step 67-118 s of MIOpen's find (profiles/r5_new_batch_shapes.txt), and the find mode that skips it (MIOPEN_FIND_MODE=2)
runs the step 18 x slower.  What MIOpen learns beyond the shipped records stays in the private copy and goes with the process: a
user who wants to keep it sets MIOPEN_USER_DB_PATH (respected, see below).  The process works on a PRIVATE
COPY (a temporary directory owned by this process id): MIOpen appends what it learns to the user db, and a later process that
read those records chose other kernels -- the same `evaluate.main` command gave different last bits from run to run until the
copy was made private.  The files are keyed to ONE MIOpen build (their names carry its version): on any other stack MIOpen
ignores them silently and writes files of its own name -- `report()["miopen_foreign_files"]` shows exactly that, and
tests/test_kernel_choices_gpu.py fails on it.

PyTorch's TunableOp picks, per GEMM shape, the fastest of hipBLASLt's and rocBLAS's solutions; `tunableop/` ships the recorded
choices (look-up only: tuning stays off, a shape that is not in the file runs on the default solution).  The file's validator
lines make PyTorch ignore it on any other stack; `report()["tunableop_validators_match"]` compares them with the running one's.

Environment: an explicit MIOPEN_USER_DB_PATH / PYTORCH_TUNABLEOP_ENABLED set by the USER is respected (a value this module set in
a parent process is not: children make their own copy -- DIB_KERNEL_CHOICES_OWNER records, per variable family, which ones this
module set: a child overrides only those); DIB_NO_MIOPEN_DB=1 / DIB_NO_TUNABLEOP=1 opt out; DIB_MIOPEN_DB_INPLACE=1 works on the
shipped directory itself.

The data is keyed to one software stack, so what ships next to it is the TOOL that makes it for the running one:

    python -m detectinblur_amd.kernel_choices --fill [--shapes bench|coco-train|coco-eval|all] [--tune] [--install]

runs, in a fresh child process per shape set, the detector's train step / inference / estimator pass over the set's padded input
shapes with MIOpen's find results and TunableOp's tuning going into a private output directory (`--out`, default
./kernel_choices_out), and with `--install` copies the result over the package's `miopen_db/` and `tunableop/`.  `--tune` adds
MIOpen's solver search (MIOPEN_FIND_ENFORCE=SEARCH) for the b = 8 train shapes (slow: ~40 minutes).  `--report` prints `report()`
for the running stack behind one small convolution and GEMM and, when the shipped files are foreign to it, exactly that command.
"""
import atexit
import os
import shutil
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
_OWNER = "DIB_KERNEL_CHOICES_OWNER"          # "<pid>:<families>": the process whose private copies the variables name, and which families ("miopen", "tunableop") it set
_state = {}


def _cleanup(path, pid):
    if os.getpid() == pid:                   # a forked child runs the parent's atexit handlers: it must not delete the parent's copy
        shutil.rmtree(path, True)


def _ours():
    """The variable families ("miopen", "tunableop") that this module exported in ANOTHER process (a parent): only those may be
    overridden here; anything else in the environment is the user's and stays."""
    owner = os.environ.get(_OWNER)
    if owner is None:
        return ()
    pid, sep, fams = owner.partition(":")
    if pid == str(os.getpid()):
        return ()
    if not sep:                                   # a marker without families (an older build of this module in the parent): both
        return ("miopen", "tunableop")
    return tuple(f for f in fams.split(",") if f)


def use_shipped_kernel_choices():
    """Idempotent.  Must run before the first convolution / GEMM of the process; touches neither torch nor the GPU."""
    if _state.get("pid") == os.getpid():
        return
    inherited = _ours()
    _state.clear()
    _state["pid"] = os.getpid()
    mine = []
    # ---- MIOpen user find-db + perf-db
    db = os.path.join(_HERE, "miopen_db")
    if not os.environ.get("DIB_NO_MIOPEN_DB") and os.path.isdir(db) and ("miopen" in inherited or "MIOPEN_USER_DB_PATH" not in os.environ):
        if os.environ.get("DIB_MIOPEN_DB_INPLACE"):
            if os.access(db, os.W_OK):
                os.environ["MIOPEN_USER_DB_PATH"] = db
                _state["miopen_dir"] = db
                mine.append("miopen")
        else:
            tmp = tempfile.mkdtemp(prefix="dib_miopen_db_")
            shipped = {}
            for f in os.listdir(db):
                if os.path.isfile(os.path.join(db, f)):
                    shutil.copy(os.path.join(db, f), tmp)
                    shipped[f] = os.path.getsize(os.path.join(tmp, f))
            os.environ["MIOPEN_USER_DB_PATH"] = tmp
            _state["miopen_dir"], _state["miopen_shipped"] = tmp, shipped
            mine.append("miopen")
            atexit.register(_cleanup, tmp, os.getpid())
    # ---- TunableOp results
    csv = os.path.join(_HERE, "tunableop", "tunableop_results.csv")
    if not os.environ.get("DIB_NO_TUNABLEOP") and os.path.isfile(csv) and ("tunableop" in inherited or "PYTORCH_TUNABLEOP_ENABLED" not in os.environ):
        tdir = tempfile.mkdtemp(prefix="dib_tunableop_")
        for ordinal in range(16):                    # PyTorch inserts the device ordinal before the extension
            shutil.copy(csv, os.path.join(tdir, "tunableop_results%d.csv" % ordinal))
        os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
        if "tunableop" in inherited or "PYTORCH_TUNABLEOP_TUNING" not in os.environ:
            os.environ["PYTORCH_TUNABLEOP_TUNING"] = "0"
        os.environ["PYTORCH_TUNABLEOP_FILENAME"] = os.path.join(tdir, "tunableop_results.csv")
        _state["tunableop_csv"] = csv
        mine.append("tunableop")
        atexit.register(_cleanup, tdir, os.getpid())
    if mine:
        # families an ancestor exported and this process left alone (opted out here) stay marked as this module's: the variables
        # still name the ancestor's copies, and a grandchild must not take them for the user's
        os.environ[_OWNER] = "%d:%s" % (os.getpid(), ",".join(sorted(set(mine) | set(inherited))))


def _csv_validators(path):
    out = {}
    with open(path) as f:
        for line in f:
            parts = line.rstrip("\n").split(",")
            if parts[0] != "Validator":
                break
            out[parts[1]] = parts[2]
    return out


def report():
    """What the shipped data amounts to on the running stack (cheap; call it after the work whose choices are in question):
      miopen_user_db            the private directory MIOpen reads and appends to (None: not installed by this module)
      miopen_foreign_files      files in it that were NOT shipped: MIOpen wrote a db under another name, i.e. the shipped files
                                belong to another MIOpen build and were ignored
      miopen_db_growth_bytes    bytes MIOpen appended to the shipped files since the copy: find-db misses (0 = every convolution
                                shape met so far was in the shipped find-db)
      tunableop_validators_match  the shipped file's validator lines equal the running PyTorch's (None: TunableOp not installed
                                by this module, or torch.cuda.tunable unavailable)
      tunableop_entries_loaded  results PyTorch holds (after the first GEMM: the shipped entries + nothing, tuning being off)
      regenerate_with           only when the shipped files are foreign to the running stack: the command that makes this stack's
                                (also printed to stderr, once)"""
    r = {"installed": _state.get("pid") == os.getpid(), "miopen_user_db": _state.get("miopen_dir"), "miopen_foreign_files": None,
         "miopen_db_growth_bytes": None, "tunableop_validators_match": None, "tunableop_entries_loaded": None,
         "tunableop_shipped_entries": None}
    d, shipped = _state.get("miopen_dir"), _state.get("miopen_shipped")
    if d and shipped is not None and os.path.isdir(d):
        now = {f: os.path.getsize(os.path.join(d, f)) for f in os.listdir(d) if os.path.isfile(os.path.join(d, f))}
        # (MIOpen keeps companions next to a db under the db's own name: `<db>.time`, `<db>.lock`)
        r["miopen_foreign_files"] = sorted(f for f in now if not any(f == n or f.startswith(n + ".") for n in shipped))
        r["miopen_db_growth_bytes"] = sum(now.get(f, 0) - n for f, n in shipped.items())
    csv = _state.get("tunableop_csv")
    if csv:
        want = _csv_validators(csv)
        with open(csv) as f:
            r["tunableop_shipped_entries"] = sum(1 for line in f if not line.startswith("Validator"))
        try:
            import torch
            if not torch.cuda.is_initialized():
                raise RuntimeError("the GPU is not initialised yet (report() never initialises it)")
            have = dict(tuple(v) for v in torch.cuda.tunable.get_validators())
            r["tunableop_validators_match"] = all(have.get(k) == v for k, v in want.items())
            r["tunableop_validators_shipped"] = want
            if not r["tunableop_validators_match"]:
                r["tunableop_validators_running"] = have
            r["tunableop_entries_loaded"] = len(torch.cuda.tunable.get_results())
        except Exception as e:      # noqa: BLE001 -- no GPU, or a PyTorch without torch.cuda.tunable
            r["tunableop_error"] = "%s: %s" % (type(e).__name__, e)
    if r["miopen_foreign_files"] or r["tunableop_validators_match"] is False:
        # the shipped data belongs to another stack: say how to make this stack's, once per process on stderr and in the report
        r["regenerate_with"] = FILL_COMMAND
        if not _state.get("hinted"):
            _state["hinted"] = True
            import sys
            print(foreign_hint(r), file=sys.stderr)
    return r


# ---- the tool: regenerate the shipped data for the running stack ---------------------------------------------------------------
FILL_COMMAND = "python -m detectinblur_amd.kernel_choices --fill --shapes all --install"
_SIDES = list(range(800, 1345, 32))
SHAPE_SETS = {
    # (kind, shapes): "train" = padded b = 8 batch shapes (three train steps each), "eval" = batch-1 input sizes (detector + estimator)
    "bench": (("train", [(800, 1344)]), ("eval", [(800, 1333), (800, 1088), (480, 640), (640, 480), (427, 640)])),
    "coco-train": (("train", [(800, w) for w in _SIDES] + [(h, 800) for h in _SIDES[1:]] +
                    [(768, 1344), (1344, 768), (736, 1344), (1344, 736), (1088, 1088), (1024, 1024)]),),
    "coco-eval": (("eval", [(800, w) for w in _SIDES] + [(h, 800) for h in _SIDES[1:]] + [(h, 1344) for h in range(512, 800, 32)] +
                   [(1344, w) for w in range(512, 800, 32)]),),
}


def foreign_hint(rep=None):
    """The one line to print when the shipped find-db / TunableOp file does not belong to the running stack (None otherwise)."""
    if rep is None:
        rep = report()
    if rep.get("miopen_foreign_files") or rep.get("tunableop_validators_match") is False:
        return ("detectinblur_amd: the shipped kernel-choice data was made for another MIOpen / PyTorch build and is being ignored "
                "(first steps search for minutes, the train step is ~6 %% slower); regenerate it for this stack with:  %s" % FILL_COMMAND)
    return None


def _fill_worker(kind_shapes, tune):
    """Child process of --fill: MIOPEN_USER_DB_PATH / PYTORCH_TUNABLEOP_* already point into the output directory."""
    import time
    import numpy as np
    import torch
    from . import utils
    from .models.faster_rcnn import fasterrcnn_resnet50_fpn
    dev = torch.device("cuda", 0)
    t0 = time.time()
    for kind, shapes in kind_shapes:
        if kind == "train":
            torch.manual_seed(0)
            model = fasterrcnn_resnet50_fpn(num_classes=91, pretrained=False, pretrained_backbone=False).to(dev).train()
            opt = utils.make_sgd([p for p in model.parameters() if p.requires_grad], 0.0004, 0.9, 1e-4)
            for k, (H, W) in enumerate(shapes):
                hh, ww = min(H, 1333), min(W, 1333)          # the largest image of such a batch: the transform pads the batch to (H, W)
                g = torch.Generator().manual_seed(k)
                imgs = [torch.rand(3, hh, ww, generator=g).to(dev) for _ in range(8)]
                tg = [{"boxes": torch.tensor([[10.0, 20.0, 300.0, 400.0], [200.0, 100.0, 700.0, 600.0]], device=dev),
                       "labels": torch.tensor([3, 7], device=dev)} for _ in range(8)]
                ts = []
                for _ in range(1 if tune else 3):
                    torch.cuda.synchronize(); t1 = time.perf_counter()
                    loss = sum(model(list(imgs), [dict(t) for t in tg]).values())
                    opt.zero_grad(); loss.backward(); opt.step()
                    torch.cuda.synchronize(); ts.append(time.perf_counter() - t1)
                print("train shape %d x %d (%d of %d): first step %.1f s, last %.1f ms; %.0f s so far" % (H, W, k + 1, len(shapes), ts[0], ts[-1] * 1e3, time.time() - t0), flush=True)
            del model, opt
        else:
            from torch import nn
            from .models import net_transforms
            from .models.blur_estimator import resnet18
            m = fasterrcnn_resnet50_fpn(num_classes=91, pretrained=False, pretrained_backbone=False).to(dev).eval()
            est = resnet18(); est.fc = nn.Linear(512, 4); est = est.to(dev).eval()
            batcher = net_transforms.GeneralizedRCNNTransform(800, 1333, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], crop_images=True)
            mean, std = np.tile([0.485, 0.456, 0.406], (1, 1)), np.tile([0.229, 0.224, 0.225], (1, 1))
            with torch.no_grad():
                for k, (h, w) in enumerate(shapes):
                    img = torch.rand(3, min(h, 1333), min(w, 1333), device=dev)
                    m([img], newMeans=mean, newSTDs=std)
                    b, _ = batcher([img], None)
                    est(b.tensors)
                    torch.cuda.synchronize()
                    print("eval size %d x %d (%d of %d) at %.0f s" % (h, w, k + 1, len(shapes), time.time() - t0), flush=True)
            del m, est
    print("done", flush=True)


def _worker_argv(set_name, tune):
    import sys
    return [sys.executable, "-m", "detectinblur_amd.kernel_choices", "--worker", set_name] + (["--tune"] if tune else [])


def fill(shape_sets, out_dir, tune=False, install=False, package_dir=_HERE, worker_argv=_worker_argv, seed_from_package=True):
    """Runs every named shape set in a fresh child process whose MIOpen user db and TunableOp results file live in `out_dir`
    (seeded with the package's files, so that records which already belong to the running stack are kept and extended); returns
    {"miopen": [files], "tunableop": path or None, "returncodes": {...}}.  install: copy the result over the package's data."""
    import subprocess
    mdir, tdir = os.path.join(out_dir, "miopen_db"), os.path.join(out_dir, "tunableop")
    os.makedirs(mdir, exist_ok=True)
    os.makedirs(tdir, exist_ok=True)
    if seed_from_package:
        for sub, dst in (("miopen_db", mdir), ("tunableop", tdir)):
            src = os.path.join(package_dir, sub)
            if os.path.isdir(src):
                for f in os.listdir(src):
                    if os.path.isfile(os.path.join(src, f)) and not os.path.exists(os.path.join(dst, f)):
                        shutil.copy(os.path.join(src, f), dst)
    base = os.path.join(tdir, "tunableop_results.csv")
    if os.path.isfile(base):      # PyTorch reads and writes `<name><ordinal>.csv`
        shutil.copy(base, os.path.join(tdir, "tunableop_results0.csv"))
    rcs = {}
    for name in shape_sets:
        env = dict(os.environ, DIB_NO_MIOPEN_DB="1", DIB_NO_TUNABLEOP="1", MIOPEN_USER_DB_PATH=mdir, PYTORCH_TUNABLEOP_ENABLED="1",
                   PYTORCH_TUNABLEOP_TUNING="1", PYTORCH_TUNABLEOP_FILENAME=base, PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS="15",
                   PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS="3")
        env.pop(_OWNER, None)
        if tune:
            env["MIOPEN_FIND_ENFORCE"] = "3"          # SEARCH: every applicable solver's tuning grid, results into the user perf-db
        print("== %s%s: %s" % (name, " (solver search)" if tune else "", " ".join(worker_argv(name, tune))), flush=True)
        rcs[name] = subprocess.run(worker_argv(name, tune), env=env).returncode
    got = os.path.join(tdir, "tunableop_results0.csv")
    if os.path.isfile(got):
        shutil.copy(got, base)
    res = {"miopen": sorted(f for f in os.listdir(mdir) if f.endswith((".ufdb.txt", ".udb.txt"))),
           "tunableop": base if os.path.isfile(base) else None, "returncodes": rcs, "out": out_dir}
    if install and all(rc == 0 for rc in rcs.values()):
        for f in res["miopen"]:
            os.makedirs(os.path.join(package_dir, "miopen_db"), exist_ok=True)
            shutil.copy(os.path.join(mdir, f), os.path.join(package_dir, "miopen_db", f))
        if res["tunableop"]:
            os.makedirs(os.path.join(package_dir, "tunableop"), exist_ok=True)
            shutil.copy(base, os.path.join(package_dir, "tunableop", "tunableop_results.csv"))
        res["installed_into"] = package_dir
    return res


def main(argv=None):
    import argparse
    import json
    ap = argparse.ArgumentParser(prog="python -m detectinblur_amd.kernel_choices", description=__doc__.split("\n\n")[0])
    ap.add_argument("--fill", action="store_true", help="regenerate the MIOpen find-db / perf-db and the TunableOp file for the running stack")
    ap.add_argument("--shapes", default="bench", help="bench | coco-train | coco-eval | all (comma-separated)")
    ap.add_argument("--tune", action="store_true", help="also run MIOpen's solver search (MIOPEN_FIND_ENFORCE=SEARCH) over the set's train shapes")
    ap.add_argument("--install", action="store_true", help="copy the result over the package's miopen_db/ and tunableop/")
    ap.add_argument("--out", default="kernel_choices_out")
    ap.add_argument("--report", action="store_true", help="print report() for the running stack (needs a GPU)")
    ap.add_argument("--worker", help=argparse.SUPPRESS)
    a = ap.parse_args(argv)
    if a.worker:
        _fill_worker(SHAPE_SETS[a.worker], a.tune)
        return 0
    if a.report:
        use_shipped_kernel_choices()
        import torch
        x = torch.randn(2, 64, 200, 336, device="cuda").to(memory_format=torch.channels_last)
        torch.nn.Conv2d(64, 64, 3, padding=1).cuda().to(memory_format=torch.channels_last)(x)
        torch.nn.Linear(1024, 91).cuda()(torch.randn(1000, 1024, device="cuda"))
        torch.cuda.synchronize()
        rep = report()
        print(json.dumps(rep, indent=1))
        hint = foreign_hint(rep)
        if hint:
            print(hint)
        return 0
    if a.fill:
        names = list(SHAPE_SETS) if a.shapes == "all" else [n.strip() for n in a.shapes.split(",")]
        for n in names:
            if n not in SHAPE_SETS:
                ap.error("unknown shape set %r (bench, coco-train, coco-eval, all)" % n)
        res = fill(names, os.path.abspath(a.out), tune=a.tune, install=a.install)
        print(json.dumps(res, indent=1))
        return 0 if all(rc == 0 for rc in res["returncodes"].values()) else 1
    ap.print_help()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
