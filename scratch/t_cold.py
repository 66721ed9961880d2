import sys, time
T0 = time.perf_counter()
sys.path.insert(0, '.')
import torch, bench
from detectinblur_amd import blur_ops
from detectinblur_amd.models import blur_functions as BF
torch.cuda.set_device(0)
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, torch.device("cuda", 0))
def step():
    blur_ops.invalidate_cache()
    batch = list(images)
    BF.blur_image_list(batch, dicts, psfs)
    return batch
print("setup done at %.2f s" % (time.perf_counter() - T0))
for rep in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("t=%.3f rep %d: host %.1f us/step, with sync %.1f us/step" % (t2 - T0, rep, (t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6), flush=True)
