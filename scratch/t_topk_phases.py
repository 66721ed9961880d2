"""Where dib_topk_levels spends its time: row length and k varied separately (one row)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from detectinblur_amd.models import detector_ops as ops
def timeit(name, fn, reps=50):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%-50s %.1f us per call" % (name, (t2 - t0) / reps * 1e6), flush=True)
g = torch.Generator().manual_seed(0)
for cnt in (201600, 50400, 12600, 3000):
    v = torch.randn(1, cnt, generator=g).cuda()
    for k in (1, 64, 1000, 2000):
        if k <= cnt:
            timeit("cnt=%d k=%d" % (cnt, k), lambda: ops.topk_levels_hip(v, [cnt], [k], k))
v = torch.randn(1, 201600, generator=g).cuda()
timeit("cnt=201600 k=cnt-free take_all (k=2048 of 2048)", lambda: ops.topk_levels_hip(v[:, :2048].contiguous(), [2048], [2048], 2048))
timeit("empty launch floor (cnt=1,k=1)", lambda: ops.topk_levels_hip(v[:, :1].contiguous(), [1], [1], 1))
