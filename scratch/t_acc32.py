import sys
sys.path.insert(0, '.')
import torch, bench
from detectinblur_amd import blur_ops, _lib
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, torch.device("cuda", 0))
tables = blur_ops.compact_psfs(psfs, normalize=True)
idx = list(range(8))
for mode, name in ((0, "bitexact"), (1, "fp32 accumulate")):
    for _ in range(5): blur_ops.sparse_blur(list(images), idx, tables, mode)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): blur_ops.sparse_blur(list(images), idx, tables, mode)
    e1.record(); e1.synchronize()
    print("%-16s %.1f us per launch" % (name, e0.elapsed_time(e1) / 50 * 1e3))
