"""detectinblur_amd: MI355X-native implementation of detectInBlur's `--gpu_blur` hot path.

Module names mirror the reference (mohammed-amr/detectInBlur) so call sites read the same:
    detectinblur_amd.models.blur_functions.blur_image_list / manual_blur
    detectinblur_amd.utils.expand_targets / fix_bounding_box_squeeze / get_norm_params
    detectinblur_amd.transforms.BlurImage, detectinblur_amd.motion_blur.*
All device work goes through the C ABI of include/dib.h (libdib_hip.so, hand-written HIP for
gfx950); there is no CPU fallback.
"""
__version__ = "0.1.0"

import os as _os

# (r4) Next to the find-db sits the user PERF-db (`*.udb.txt`) of a tuning run (`scratch/tune_miopen_train.sh`:
# MIOPEN_FIND_ENFORCE=SEARCH over the train step's convolutions, 7 minutes on one MI355X): the tile configuration of MIOpen's tunable
# solvers per shape, measured instead of taken from their heuristics -- train step 100.0 -> 94.3 ms on the same box
# (`scratch/t_miopen_ab.sh`).  It travels in the same private copy.
# MIOpen picks each convolution's kernel by timing every applicable solver the first time it sees a
# shape (~90 s for this detector at b=8 x 800x1344 on a fresh machine) and remembers the result in a
# "user find-db".  `miopen_db/` ships that database for every shape the bench, the drivers (800 x 1333 and
# native COCO sizes) and the GPU tests meet on gfx950 (scratch/fill_miopen_db.sh regenerates it), so a fresh
# process warms up in about a second and -- the choice being recorded, not re-measured -- every process
# picks the same kernels.  The process works on a PRIVATE COPY of it (a temporary directory):
# MIOpen appends what it learns about new shapes to the user db, and a later process that read those
# records chose other kernels than the process that wrote them -- the same `evaluate.main` command gave
# different last bits from one run to the next (tests/test_full_size_gpu.py), and a process that ran in
# MIOpen's deterministic mode left records that made every later process 50x slower
# (profiles/r4_nondeterminism.txt).  From a private copy every process starts from the same state.
# Respect an explicit MIOPEN_USER_DB_PATH; DIB_NO_MIOPEN_DB=1 opts out; DIB_MIOPEN_DB_INPLACE=1 works on
# the shipped directory itself (scratch/tune_eval_db.py: to extend the shipped db).  Must run before the
# first convolution.
if not _os.environ.get("DIB_NO_MIOPEN_DB") and "MIOPEN_USER_DB_PATH" not in _os.environ:
    _db = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "miopen_db")
    if _os.path.isdir(_db):
        if _os.environ.get("DIB_MIOPEN_DB_INPLACE"):
            if _os.access(_db, _os.W_OK):
                _os.environ["MIOPEN_USER_DB_PATH"] = _db
        else:
            import atexit as _atexit
            import shutil as _shutil
            import tempfile as _tempfile
            _tmp = _tempfile.mkdtemp(prefix="dib_miopen_db_")
            for _f in _os.listdir(_db):
                if _os.path.isfile(_os.path.join(_db, _f)):
                    _shutil.copy(_os.path.join(_db, _f), _tmp)
            _os.environ["MIOPEN_USER_DB_PATH"] = _tmp
            _atexit.register(_shutil.rmtree, _tmp, True)

# GEMMs (the 1x1 convolutions that run as GEMMs, the box head, the RPN predictor): PyTorch's TunableOp picks, per GEMM shape, the
# fastest of hipBLASLt's and rocBLAS's solutions instead of the libraries' heuristic default -- at batch 1 the trunk's small GEMMs
# (M = 1,050 .. 67,200 rows) run 1.3-2x faster that way (trunk replay 6.36 -> 5.71 ms, scratch/t_tunable.sh).  `tunableop/` ships the
# recorded choices for the shapes of the bench, the train step at b = 8 and batch-1 inference at every input size a COCO image reaches
# after the detector's transform (min side 800, max side 1333, padded to 32: a grid of 53 sizes) (scratch/fill_tunableop.sh and
# scratch/fill_dbs_grid.sh regenerate it; the find-db above covers the same grid); like the find-db above the process reads a PRIVATE COPY, with tuning off: a shape
# that is not in the file runs on the default solution, nothing is measured at run time, every process makes the same choice.
# The file's validator lines (PyTorch / ROCm / hipBLASLt / rocBLAS versions, gfx950) make PyTorch ignore it on any other stack.
# An explicit PYTORCH_TUNABLEOP_ENABLED (either value) is respected; DIB_NO_TUNABLEOP=1 opts out.  Must run before the first GEMM.
if not _os.environ.get("DIB_NO_TUNABLEOP") and "PYTORCH_TUNABLEOP_ENABLED" not in _os.environ:
    _csv = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "tunableop", "tunableop_results.csv")
    if _os.path.isfile(_csv):
        import atexit as _atexit
        import shutil as _shutil
        import tempfile as _tempfile
        _tdir = _tempfile.mkdtemp(prefix="dib_tunableop_")
        for _ordinal in range(16):                   # PyTorch inserts the device ordinal before the extension
            _shutil.copy(_csv, _os.path.join(_tdir, "tunableop_results%d.csv" % _ordinal))
        _os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
        _os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "0")
        _os.environ["PYTORCH_TUNABLEOP_FILENAME"] = _os.path.join(_tdir, "tunableop_results.csv")
        _atexit.register(_shutil.rmtree, _tdir, True)
