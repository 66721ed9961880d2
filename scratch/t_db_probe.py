import os, sys
sys.path.insert(0, '.')
import torch
import bench
dev = torch.device("cuda", 0)
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev)
p = os.environ.get("MIOPEN_USER_DB_PATH")
print("MIOPEN_USER_DB_PATH", p, os.listdir(p) if p and os.path.isdir(p) else None, "TMPDIR", os.environ.get("TMPDIR"))
r = bench.train_step_bench(images, dicts, psfs, dev, None, 1, 0, 4, 3, account=False)[0]
print("ms_per_step", r["ms_per_step"])
print("db dir after", os.listdir(p))
