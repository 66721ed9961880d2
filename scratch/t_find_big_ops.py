"""Every ATen / custom op of a train step that touches a tensor of >= 30 M elements (the 550 MB class), with its parent chain."""
import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench as B
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
host = B.make_psfs_host(0)
images, dicts, psfs, _, _ = B.make_workload(0, dev, host)
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    B.train_step_bench(images, dicts, psfs, dev, None, 1, 0, 1, 1, account=False)
cnt = collections.Counter()
for e in prof.events():
    if not e.input_shapes:
        continue
    big = 0
    for sh in e.input_shapes:
        n = 1
        for d in (sh or []):
            n *= d
        if sh:
            big = max(big, n)
    if big >= 30_000_000 and e.name.startswith("aten::") and not any(c.name.startswith("aten::") for c in e.cpu_children):
        chain, p = [], e.cpu_parent
        while p is not None and len(chain) < 3:
            chain.append(p.name.replace("autograd::engine::evaluate_function: ", "eval:"))
            p = p.cpu_parent
        cnt[(e.name, str([tuple(s) for s in e.input_shapes if s][:2]), " < ".join(chain))] += 1
for (name, shapes, chain), c in sorted(cnt.items(), key=lambda kv: (-kv[1], kv[0])):
    print("%3d x %-28s %-46s %s" % (c, name, shapes, chain))
