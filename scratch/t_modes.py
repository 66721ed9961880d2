"""Kernel time of the blur's three accumulation modes on the BASELINE batch (tables compacted ahead), and of the 256-wide shape's
DIB_ACC_FP32 (the second implementation)."""
import ctypes, json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from detectinblur_amd import _lib, blur_ops
dev = torch.device("cuda", 0)
host = bench.make_psfs_host(0)
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev, host)
l = _lib.lib()
l.dib_debug_set_shape.argtypes = [ctypes.c_int]; l.dib_debug_set_shape.restype = None
tables = blur_ops.compact_psfs(psfs, normalize=True)
idx = sorted(range(8), key=lambda k: -dicts[k]["psf_taps"]); ordered = [images[k] for k in idx]
def ev(fn, reps=200):
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(8): fn()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
out = {}
t_end = time.perf_counter() + 0.5
while time.perf_counter() < t_end: blur_ops.sparse_blur(list(ordered), idx, tables)
for name, mode in (("bitexact", _lib.DIB_ACC_BITEXACT), ("fma16", _lib.DIB_ACC_FMA16), ("fp32", _lib.DIB_ACC_FP32)):
    for _ in range(300): blur_ops.sparse_blur(list(ordered), idx, tables, mode)
    out[name + "_us"] = sorted(ev(lambda: blur_ops.sparse_blur(list(ordered), idx, tables, mode)) for _ in range(5))[2]
l.dib_debug_set_shape(1)
for _ in range(300): blur_ops.sparse_blur(list(ordered), idx, tables, _lib.DIB_ACC_FP32)
out["fp32_256wide_shape_us"] = sorted(ev(lambda: blur_ops.sparse_blur(list(ordered), idx, tables, _lib.DIB_ACC_FP32)) for _ in range(5))[2]
l.dib_debug_set_shape(0)
ex = blur_ops.sparse_blur(list(ordered), idx, tables)
f32 = blur_ops.sparse_blur(list(ordered), idx, tables, _lib.DIB_ACC_FP32)
out["fp32_max_abs_diff_vs_bitexact"] = max(float((a.float() - b.float()).abs().max()) for a, b in zip(ex, f32))
print(json.dumps(out))
