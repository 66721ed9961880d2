import sys, time, cProfile, pstats
sys.path.insert(0, '.')
import torch, bench
from detectinblur_amd import blur_ops
from detectinblur_amd.models import blur_functions as BF
torch.cuda.set_device(0)
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, torch.device("cuda", 0))
def step():
    pass  # (round 3: the table cache is gone)
    batch = list(images)
    BF.blur_image_list(batch, dicts, psfs)
    return batch
for _ in range(500): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host %.1f us/step, with sync %.1f us/step" % ((t1 - t0) / 2000 * 1e6, (t2 - t0) / 2000 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
