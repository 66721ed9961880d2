"""Host cost of one eager step (blur_image_list = tap compaction + blur, two launches): the BASELINE batch vs the same call on
3 x 70 x 70 images, where the GPU needs ~10 us per step and the loop runs at the interpreter's pace."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, bench
from detectinblur_amd.models import blur_functions as BF
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev)
tiny = [torch.rand(3, 70, 70, device=dev).half() for _ in images]
for name, imgs in (("800x1333", images), ("70x70", tiny)):
    def step():
        batch = list(imgs)
        BF.blur_image_list(batch, dicts, psfs)
        return batch
    for _ in range(500): step()
    res = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2000): step()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        res.append(((t1 - t0) / 2000 * 1e6, (t2 - t0) / 2000 * 1e6))
    res.sort(key=lambda r: r[1])
    print("%-9s enqueue %.1f us/step, with final sync %.1f us/step (median of 5)" % (name, res[2][0], res[2][1]))
import os
if os.environ.get("DIB_HOSTPROF"):
    import cProfile, pstats

    def step():
        batch = list(tiny)
        BF.blur_image_list(batch, dicts, psfs)
        return batch
    pr = cProfile.Profile(); pr.enable()
    for _ in range(3000): step()
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
