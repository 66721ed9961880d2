"""Do two blur launches on two streams overlap (the drain of one under the ramp of the other)?  Per-launch time of the
BASELINE blur issued alternately on 1 / 2 / 3 streams, eager and as a HIP graph of 24 launches.   python scratch/t_overlap.py"""
import sys, time
sys.path.insert(0, '.')
import torch
import bench
from detectinblur_amd import blur_ops
dev = torch.device("cuda", 0)
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev)
tables = blur_ops.compact_psfs(psfs, normalize=True)
idx = sorted(range(8), key=lambda k: -dicts[k]["psf_taps"])
ordered = [images[k] for k in idx]
def blur(): return blur_ops.sparse_blur(list(ordered), idx, tables, 0)
for _ in range(300): blur()
torch.cuda.synchronize()
def wall(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for ns in (1, 2, 3):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    keep = [None] * 6
    def round_():
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                keep[i] = blur()
    t = wall(round_, 600) / ns
    g = torch.cuda.CUDAGraph(); kk = []
    with torch.cuda.graph(g, stream=streams[0], capture_error_mode="thread_local"):
        f = torch.cuda.Event(); f.record(streams[0])
        for s in streams[1:]: s.wait_event(f)
        for i in range(24):
            with torch.cuda.stream(streams[i % ns]): kk.append(blur())
        for s in streams[1:]:
            j = torch.cuda.Event(); j.record(s); streams[0].wait_event(j)
    for _ in range(10): g.replay()
    tg = wall(g.replay, 200) / 24
    print("%d stream(s): eager %.2f us per blur, graph %.2f us per blur" % (ns, t, tg), flush=True)
    del g, kk
