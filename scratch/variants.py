import sys, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from detectinblur_amd import blur_ops, _lib
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, torch.device("cuda", 0))
tables = blur_ops.compact_psfs(psfs, normalize=True)
idx = list(range(8))
l = _lib.lib(); l.dib_debug_set_variant.argtypes = [ctypes.c_int, ctypes.c_int]; l.dib_debug_set_variant.restype = None
ref = None
for nw, tpw in ((8,1),(8,2),(4,1),(4,2)):
    l.dib_debug_set_variant(nw, tpw)
    for _ in range(5): outs = blur_ops.sparse_blur(list(images), idx, tables)
    torch.cuda.synchronize()
    if ref is None: ref = [o.clone() for o in outs]
    same = all(torch.equal(a, b) for a, b in zip(ref, outs))
    # back-to-back launches, one event pair: average launch duration incl. inter-kernel gaps
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n): blur_ops.sparse_blur(list(images), idx, tables)
    e1.record(); e1.synchronize()
    print("NW=%d TPW=%d identical=%s  avg %.2f us per launch" % (nw, tpw, same, e0.elapsed_time(e1) / n * 1e3))
