"""Which Python lines issue the large aten::copy_ calls of a train step (torch profiler with shapes and stacks)."""
import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench as B
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
host = B.make_psfs_host(0)
images, dicts, psfs, _, _ = B.make_workload(0, dev, host)
# reuse the bench's step function by running it under the profiler: train_step_bench builds model + step internally, so
# profile the whole call with few steps and skip what happens before the last step by only looking at shapes
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    B.train_step_bench(images, dicts, psfs, dev, None, 1, 0, 1, 1, account=False)
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_",) and e.input_shapes and e.input_shapes[0]:
        n = 1
        for d in e.input_shapes[0]:
            n *= d
        if n >= 4_000_000:
            chain, p = [], e.cpu_parent
            while p is not None and len(chain) < 6:
                chain.append(p.name)
                p = p.cpu_parent
            st = [f for f in (e.stack or []) if "detectinblur_amd" in f][:1]
            cnt[(tuple(e.input_shapes[0]), " < ".join(chain), " ".join(s.split("/")[-1] for s in st))] += 1
for (shape, chain, where), c in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print("%3d x %-22s %s   %s" % (c, shape, chain, where))
