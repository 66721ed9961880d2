"""Blur time of the BASELINE image batch under heavier blur than BASELINE's (longer exposure fractions, the three `expl`
values of the evaluation sweep), per tile shape: how the smaller (13 x 25) segments of the default shape pay on PSFs that
need several windows.   python scratch/t_bigpsf.py"""
import sys, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
import bench
import os
from detectinblur_amd import _lib
if os.environ.get("DIB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["DIB_LIB"])     # e.g. an older build, for comparison
from detectinblur_amd import blur_ops
from detectinblur_amd.motion_blur.generate_PSF import PSF
from detectinblur_amd.motion_blur.generate_trajectory import Trajectory
dev = torch.device("cuda", 0)
images = [torch.rand(3, 800, 1333, generator=torch.Generator().manual_seed(1337 + i)).half().to(dev) for i in range(8)]
l = _lib.lib(); l.dib_debug_set_shape.argtypes = [ctypes.c_int]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timeit(fn, reps=100):
    for _ in range(100): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) * 1000 / reps
for expl in (0.005, 0.001, 0.00005):
    for frac in (1 / 5, 1 / 2, 1.0):
        np.random.seed(7)
        psfs = []
        for i in range(8):
            tr = Trajectory(canvas=256, max_len=96, expl=expl).fit().fit()
            p = PSF(canvas=256, trajectory=tr, fraction=[frac]); p.fit(); p.centerPSF()
            psfs.append(torch.HalfTensor(np.ascontiguousarray(p.PSFs[0][64:192, 64:192])).to(dev))
        tables = blur_ops.compact_psfs(psfs, normalize=True)
        taps = [tables.header(i)[0] for i in range(8)]
        nseg = [len(tables.segments(i)) for i in range(8)]
        idx = sorted(range(8), key=lambda k: -taps[k])
        ordered = [images[k] for k in idx]
        res = []
        for shape in (0, 1):
            l.dib_debug_set_shape(shape)
            res.append(timeit(lambda: blur_ops.sparse_blur(list(ordered), idx, tables, 0)))
        l.dib_debug_set_shape(0)
        fma = timeit(lambda: blur_ops.sparse_blur(list(ordered), idx, tables, 2))
        if os.environ.get("DIB_LIB"):
            print("expl %-8g fraction %.2f: taps %s segments %s | old build: shape 0 %.1f us  shape 1 %.1f us" % (expl, frac, taps, nseg, res[0], res[1]), flush=True)
        else:
            print("expl %-8g fraction %.2f: taps %s segments %s | quad %.1f us  256-wide %.1f us | quad fma16 %.1f us" % (
                expl, frac, taps, nseg, res[0], res[1], fma), flush=True)
