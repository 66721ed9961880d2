import sys, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from detectinblur_amd import blur_ops, _lib
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, torch.device("cuda", 0))
tables = blur_ops.compact_psfs(psfs, normalize=True)
idx = list(range(8))
l = _lib.lib(); l.dib_debug_set_tile_order.argtypes = [ctypes.c_int]; l.dib_debug_set_tile_order.restype = None
order = int(sys.argv[1]) if len(sys.argv) > 1 else -1
ref = None
for rep in range(3 if order < 0 else 1):
  for o in ((0, 1) if order < 0 else (order,)):
    l.dib_debug_set_tile_order(o)
    for _ in range(5): outs = blur_ops.sparse_blur(list(images), idx, tables)
    torch.cuda.synchronize()
    if ref is None: ref = [x.clone() for x in outs]
    same = all(torch.equal(a, b) for a, b in zip(ref, outs))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n): blur_ops.sparse_blur(list(images), idx, tables)
    e1.record(); e1.synchronize()
    print("tile order %d identical=%s  avg %.2f us per launch" % (o, same, e0.elapsed_time(e1) / n * 1e3), flush=True)
