"""Shipped kernel-choice data for the detector's library kernels (MIOpen convolutions, hipBLASLt / rocBLAS GEMMs), opt-in.

`use_shipped_kernel_choices()` is what the drivers (train.py, evaluate.py, train_blur_estimator.py: the place of the
reference's `main`, train.py:89) and bench.py call before their first convolution.  Importing the package does NOT touch the
process environment any more: a drop-in library has no business exporting MIOPEN_* / PYTORCH_TUNABLEOP_* behind its caller's
back (round-4 review).  `report()` says what the running stack actually did with the data.

MIOpen picks each convolution's kernel by timing every applicable solver the first time it sees a shape (~90 s for this detector
at b = 8 x 800 x 1344 on a fresh machine) and remembers the result in a "user find-db"; next to it sits the user PERF-db
(`*.udb.txt`) of a tuning run (MIOPEN_FIND_ENFORCE=SEARCH over the train step's convolutions: 100.0 -> 94.3 ms per step).
`miopen_db/` ships both for every shape the bench, the drivers and the GPU tests meet on gfx950, for the ~50 padded batch-1 input
shapes real COCO images reach in evaluation (scratch/fill_dbs_grid.sh) and for the padded b = 8 batch shapes of COCO training
(800 x 800..1344 and 800..1344 x 800 in steps of 32, a few more: scratch/fill_dbs_train.sh) -- a shape without a record costs its
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
a parent process is not: children make their own copy); DIB_NO_MIOPEN_DB=1 / DIB_NO_TUNABLEOP=1 opt out; DIB_MIOPEN_DB_INPLACE=1
works on the shipped directory itself (scratch/tune_eval_db.py: to extend it).
"""
import atexit
import os
import shutil
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
_OWNER = "DIB_KERNEL_CHOICES_OWNER"          # pid of the process whose private copies the MIOPEN_* / PYTORCH_TUNABLEOP_* variables name
_state = {}


def _cleanup(path, pid):
    if os.getpid() == pid:                   # a forked child runs the parent's atexit handlers: it must not delete the parent's copy
        shutil.rmtree(path, True)


def _ours():
    """True when the variables in the environment were exported by this module in ANOTHER process (a parent)."""
    owner = os.environ.get(_OWNER)
    return owner is not None and owner != str(os.getpid())


def use_shipped_kernel_choices():
    """Idempotent.  Must run before the first convolution / GEMM of the process; touches neither torch nor the GPU."""
    if _state.get("pid") == os.getpid():
        return
    inherited = _ours()
    _state.clear()
    _state["pid"] = os.getpid()
    # ---- MIOpen user find-db + perf-db
    db = os.path.join(_HERE, "miopen_db")
    if not os.environ.get("DIB_NO_MIOPEN_DB") and os.path.isdir(db) and (inherited or "MIOPEN_USER_DB_PATH" not in os.environ):
        if os.environ.get("DIB_MIOPEN_DB_INPLACE"):
            if os.access(db, os.W_OK):
                os.environ["MIOPEN_USER_DB_PATH"] = db
                _state["miopen_dir"] = db
        else:
            tmp = tempfile.mkdtemp(prefix="dib_miopen_db_")
            shipped = {}
            for f in os.listdir(db):
                if os.path.isfile(os.path.join(db, f)):
                    shutil.copy(os.path.join(db, f), tmp)
                    shipped[f] = os.path.getsize(os.path.join(tmp, f))
            os.environ["MIOPEN_USER_DB_PATH"] = tmp
            _state["miopen_dir"], _state["miopen_shipped"] = tmp, shipped
            atexit.register(_cleanup, tmp, os.getpid())
    # ---- TunableOp results
    csv = os.path.join(_HERE, "tunableop", "tunableop_results.csv")
    if not os.environ.get("DIB_NO_TUNABLEOP") and os.path.isfile(csv) and (inherited or "PYTORCH_TUNABLEOP_ENABLED" not in os.environ):
        tdir = tempfile.mkdtemp(prefix="dib_tunableop_")
        for ordinal in range(16):                    # PyTorch inserts the device ordinal before the extension
            shutil.copy(csv, os.path.join(tdir, "tunableop_results%d.csv" % ordinal))
        os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
        if inherited or "PYTORCH_TUNABLEOP_TUNING" not in os.environ:
            os.environ["PYTORCH_TUNABLEOP_TUNING"] = "0"
        os.environ["PYTORCH_TUNABLEOP_FILENAME"] = os.path.join(tdir, "tunableop_results.csv")
        _state["tunableop_csv"] = csv
        atexit.register(_cleanup, tdir, os.getpid())
    os.environ[_OWNER] = str(os.getpid())


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
      tunableop_entries_loaded  results PyTorch holds (after the first GEMM: the shipped entries + nothing, tuning being off)"""
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
    return r
