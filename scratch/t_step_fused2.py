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
l.dib_debug_compact_wg256.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
def step():
    batch = list(images); BF.blur_image_list(batch, dicts, psfs, psfs_complete=True); return batch
def ev(fn, reps=200):
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(8): fn()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
t_end = time.perf_counter() + 0.5
while time.perf_counter() < t_end: step()
out = {}
tabs = blur_ops.TapTables(128, 8, dev)
pa = _lib.ptr_array([p.data_ptr() for p in psfs]); st = blur_ops._stream(dev)
out["compact_1024_us"] = sorted(ev(lambda: blur_ops.compact_psfs(psfs, True)) for _ in range(5))[2]
out["compact_wg256_us"] = sorted(ev(lambda: l.dib_debug_compact_wg256(pa, 8, 1, tabs.buf.data_ptr(), st)) for _ in range(5))[2]
e = torch.empty(1, device=dev)
out["empty_kernel_us"] = sorted(ev(lambda: e.zero_()) for _ in range(5))[2]
for fused in (0, 1):
    l.dib_debug_set_step_fused(fused)
    out["step_events_us_fused%d" % fused] = sorted(ev(step) for _ in range(5))[2]
l.dib_debug_set_step_fused(1)
print(json.dumps(out))
