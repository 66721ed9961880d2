"""N detector train steps of bench.py's train_step and nothing else (for rocprofv3): python scratch/train_only.py [steps]"""
import sys
sys.path.insert(0, '.')
import torch
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
print(bench.train_step_bench(images, dicts, psfs, dev, None, 1, 0, n, 4, account=False)[0])      # rank 1: no flop accounting / torch profiler under rocprofv3
