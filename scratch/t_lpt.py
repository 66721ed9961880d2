import sys
sys.path.insert(0, '.')
import torch, bench
from detectinblur_amd import blur_ops
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, torch.device("cuda", 0))
taps = [43, 53, 28, 26, 56, 51, 19, 26]
def run(order, name):
    tables = blur_ops.compact_psfs([psfs[i] for i in order], normalize=True)
    imgs = [images[i] for i in order]
    idx = list(range(8))
    for _ in range(5): blur_ops.sparse_blur(list(imgs), idx, tables)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): blur_ops.sparse_blur(list(imgs), idx, tables)
    e1.record(); e1.synchronize()
    print("%-12s %s  %.2f us" % (name, [taps[i] for i in order], e0.elapsed_time(e1) / 100 * 1e3), flush=True)
for rep in range(2):
    run(list(range(8)), "as given")
    run(sorted(range(8), key=lambda i: -taps[i]), "heavy first")
    run(sorted(range(8), key=lambda i: taps[i]), "light first")
