import sys, time
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
for rep in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("rep %d: host %.1f us/step, with sync %.1f us/step" % (rep, (t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6), flush=True)
