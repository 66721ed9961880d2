"""Headline step variants: blur_image_list compacting on the current stream (the generic drop-in call) vs the engine's way
(tables compacted ahead on the side stream, handed over through tables=)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench as B
from detectinblur_amd import blur_ops
from detectinblur_amd.models import blur_functions as BF
dev = torch.device("cuda", 0)
host = B.make_psfs_host(0)
images, dicts, psfs, _, _ = B.make_workload(0, dev, host)


def generic():
    batch = list(images)
    BF.blur_image_list(batch, dicts, psfs)
    return batch


def ahead():
    batch = list(images)
    tables = blur_ops.compact_psfs_ahead(psfs, normalize=True, after_current=False)
    BF.blur_image_list(batch, dicts, psfs, tables=tables)
    return batch


a, b = generic(), ahead()
assert all(torch.equal(x, y) for x, y in zip(a, b))
for name, fn in (("generic", generic), ("ahead", ahead), ("generic", generic), ("ahead", ahead)):
    for _ in range(500):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        for _ in range(100):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 100 * 1e6)
    ts.sort()
    print("%-8s median %.1f us/step (min %.1f)" % (name, ts[10], ts[0]), flush=True)
