"""A/B of the blur step as one launch (fused compaction) against compaction + blur as two launches, on the BASELINE batch.
Prints the eager step (median of blocks of 100 steps, as bench.py times it) and the kernels' own durations."""
import ctypes, json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from detectinblur_amd import _lib, blur_ops
from detectinblur_amd.models import blur_functions as BF

dev = torch.device("cuda", 0)
host = bench.make_psfs_host(0)
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev, host)
l = _lib.lib()
l.dib_debug_set_step_fused.argtypes = [ctypes.c_int]; l.dib_debug_set_step_fused.restype = None

def step():
    batch = list(images)
    BF.blur_image_list(batch, dicts, psfs, psfs_complete=True)
    return batch

def blocks(n=30, k=100):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(k): step()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / k * 1e6)
    ts.sort(); return ts[len(ts) // 2], ts[0], ts[-1]

t_end = time.perf_counter() + 0.5
while time.perf_counter() < t_end: step()
out = {}
for rnd in range(3):
    for fused in (0, 1):
        l.dib_debug_set_step_fused(fused)
        for _ in range(300): step()
        out.setdefault("fused" if fused else "two_launch", []).append(blocks())
l.dib_debug_set_step_fused(1)
# kernel-only: events around 200 back-to-back steps
def ev(fn, reps=200):
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(8): fn()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
tables = blur_ops.compact_psfs(psfs, normalize=True)
idx = sorted(range(8), key=lambda k: -dicts[k]["psf_taps"]); ordered = [images[k] for k in idx]
out["blur_only_us"] = sorted(ev(lambda: blur_ops.sparse_blur(list(ordered), idx, tables)) for _ in range(5))[2]
for fused in (0, 1):
    l.dib_debug_set_step_fused(fused)
    out["step_events_us_fused%d" % fused] = sorted(ev(step) for _ in range(5))[2]
l.dib_debug_set_step_fused(1)
a, b = step(), None
l.dib_debug_set_step_fused(0); b = step(); l.dib_debug_set_step_fused(1)
out["bit_identical"] = all(torch.equal(x, y) for x, y in zip(a, b))
print(json.dumps(out))
