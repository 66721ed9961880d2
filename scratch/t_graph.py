"""Does a HIP graph of G bench steps (compaction + blur each), alternating between two captured streams, overlap the
tail of one blur with the compaction / ramp of the next?   python scratch/t_graph.py [G]"""
import sys, time
sys.path.insert(0, '.')
import torch
import bench
from detectinblur_amd import blur_ops
from detectinblur_amd.models import blur_functions as BF
dev = torch.device("cuda", 0)
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 20
def step():
    pass  # (round 3: the table cache is gone)
    batch = list(images)
    BF.blur_image_list(batch, dicts, psfs)
    return batch
for _ in range(300): step()
torch.cuda.synchronize()
def timeit(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("eager step: %.2f us" % timeit(step, 2000))
ref = [b.clone() for b in step()]
for nstreams in (1, 2, 3):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    g = torch.cuda.CUDAGraph()
    keep = []
    with torch.cuda.graph(g, stream=streams[0]):
        fork = torch.cuda.Event(); fork.record(streams[0])
        for s in streams[1:]: s.wait_event(fork)
        for i in range(G):
            with torch.cuda.stream(streams[i % nstreams]):
                keep.append(step())
        for s in streams[1:]:
            j = torch.cuda.Event(); j.record(s); streams[0].wait_event(j)
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    ok = all(torch.equal(a, b) for out in keep for a, b in zip(out, ref))
    t = timeit(g.replay, 200) / G
    print("graph of %d steps on %d stream(s): %.2f us per step, outputs identical: %s" % (G, nstreams, t, ok))
    del g, keep

# Variant: every step's tap compaction on its own captured stream C (they only depend on the resident PSFs), the blurs
# alternating between streams A and B and waiting for their tables' event -- the way engine.py runs the two (compaction on
# a side stream behind the PSF upload).
for nblur in (1, 2, 3):
    sc = torch.cuda.Stream(); sb = [torch.cuda.Stream() for _ in range(nblur)]
    g = torch.cuda.CUDAGraph(); keep = []
    with torch.cuda.graph(g, stream=sb[0], capture_error_mode="thread_local"):
        fork = torch.cuda.Event(); fork.record(sb[0])
        for s in [sc] + sb[1:]: s.wait_event(fork)
        for i in range(G):
            with torch.cuda.stream(sc):
                tabs = blur_ops.compact_psfs(psfs, normalize=True)
                tabs.ready = torch.cuda.Event(); tabs.ready.record(sc)
            with torch.cuda.stream(sb[i % nblur]):
                batch = list(images)
                BF.blur_image_list(batch, dicts, psfs, tables=tabs)
                keep.append((batch, tabs))
        for s in [sc] + sb[1:]:
            j = torch.cuda.Event(); j.record(s); sb[0].wait_event(j)
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    ok = all(torch.equal(a, b) for out, _ in keep for a, b in zip(out, ref))
    t = timeit(g.replay, 200) / G
    print("graph of %d steps, compaction stream + %d blur stream(s): %.2f us per step, outputs identical: %s" % (G, nblur, t, ok))
    del g, keep
