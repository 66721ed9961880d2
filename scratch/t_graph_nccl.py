"""bench.py's graph capture with an RCCL process group alive in the process (world size 1: the only multi-rank
ingredient a 1-GPU box can provide): does the capture survive the collective backend's watchdog thread?
    MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 python scratch/t_graph_nccl.py"""
import os, sys, time
sys.path.insert(0, '.')
import torch, torch.distributed as dist
import bench
from detectinblur_amd import blur_ops
from detectinblur_amd.models import blur_functions as BF
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="env://", device_id=dev)
t = torch.ones(4, device=dev); dist.all_reduce(t); dist.barrier()
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev)
def step():
    pass  # (round 3: the table cache is gone)
    batch = list(images)
    BF.blur_image_list(batch, dicts, psfs)
    return batch
for _ in range(100): step()
ref = [b.clone() for b in step()]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
g = torch.cuda.CUDAGraph(); keep = []
with torch.cuda.graph(g, stream=streams[0], capture_error_mode="thread_local"):
    f = torch.cuda.Event(); f.record(streams[0]); streams[1].wait_event(f)
    for i in range(20):
        with torch.cuda.stream(streams[i % 2]): keep.append(step())
    j = torch.cuda.Event(); j.record(streams[1]); streams[0].wait_event(j)
for _ in range(10): g.replay()
dist.barrier(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): g.replay()
dist.barrier(); torch.cuda.synchronize()
print("graph under an RCCL process group: %.2f us per step, identical %s" % ((time.perf_counter() - t0) / 2000 * 1e6,
      all(torch.equal(a, b) for out in keep for a, b in zip(out, ref))))
el = torch.tensor([1.0], dtype=torch.float64, device=dev); dist.all_reduce(el, op=dist.ReduceOp.MAX)
dist.destroy_process_group()
