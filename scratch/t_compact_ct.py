"""Compaction workgroup size (DIB_COMPACT_THREADS=1024 vs the default 256 for the 128 canvas): kernel time alone, eager
step, and the bench's graph of 20 steps over two streams.  Run once per setting inside one gpurun call."""
import os, sys, time
sys.path.insert(0, '.')
import torch
import bench
from detectinblur_amd import blur_ops
from detectinblur_amd.models import blur_functions as BF
dev = torch.device("cuda", 0)
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev)
def step():
    pass  # (round 3: the table cache is gone)
    batch = list(images)
    BF.blur_image_list(batch, dicts, psfs)
    return batch
for _ in range(300): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): blur_ops.compact_psfs(psfs, normalize=True)
e1.record(); e1.synchronize(); tc = e0.elapsed_time(e1) * 5
def wall(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
te = wall(step, 1000)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
g = torch.cuda.CUDAGraph(); keep = []
with torch.cuda.graph(g, stream=streams[0], capture_error_mode="thread_local"):
    f = torch.cuda.Event(); f.record(streams[0]); streams[1].wait_event(f)
    for i in range(20):
        with torch.cuda.stream(streams[i % 2]): keep.append(step())
    j = torch.cuda.Event(); j.record(streams[1]); streams[0].wait_event(j)
for _ in range(20): g.replay()
res = sorted(wall(g.replay, 100) / 20 for _ in range(5))
print("DIB_COMPACT_THREADS=%s: compaction %.2f us back to back, eager step %.2f us, graph step %.2f us (median of 5; %.2f..%.2f)" % (
    os.environ.get("DIB_COMPACT_THREADS", "256 (default)"), tc, te, res[2], res[0], res[4]))
