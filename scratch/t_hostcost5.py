"""Host cost of one eager step (blur_image_list -> blur_step -> dib_blur_step_packed) and of its parts, GPU kept busy elsewhere."""
import sys, os, time, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from detectinblur_amd import _lib, blur_ops
from detectinblur_amd.models import blur_functions as BF
dev = torch.device("cuda", 0)
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev, bench.make_psfs_host(0))
def step():
    batch = list(images); BF.blur_image_list(batch, dicts, psfs, psfs_complete=True); return batch
for _ in range(200): step()
torch.cuda.synchronize()
out = {}
def host(fn, n=2000):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        if _ % 16 == 15: torch.cuda.synchronize()
    ts.sort(); return ts[len(ts) // 2] * 1e6
out["step_host_us"] = host(step)
idx = list(range(8)); perm = sorted(range(8), key=lambda k: -dicts[k]["psf_taps"])
out["blur_step_host_us"] = host(lambda: blur_ops.blur_step([images[k] for k in perm], perm, psfs, True, 0, True, False))
# first-step latency: idle GPU, time from call to kernel completion
lat = []
for _ in range(200):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); lat.append(time.perf_counter() - t0)
lat.sort(); out["one_step_from_idle_us"] = lat[len(lat) // 2] * 1e6
print(json.dumps(out))
