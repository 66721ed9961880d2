import sys, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from detectinblur_amd import blur_ops, _lib
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, torch.device("cuda", 0))
tables = blur_ops.compact_psfs(psfs, normalize=True)
idx = list(range(8))
for _ in range(5): blur_ops.sparse_blur(list(images), idx, tables)
torch.cuda.synchronize()
nblk = 512
dbg = torch.zeros(nblk * 8, dtype=torch.int64, device="cuda")
l = _lib.lib(); l.dib_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]; l.dib_debug_set_stamp_buffer.restype = None
l.dib_debug_set_stamp_buffer(dbg.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); blur_ops.sparse_blur(list(images), idx, tables); e1.record()
torch.cuda.synchronize()
print("event ms", e0.elapsed_time(e1))
l.dib_debug_set_stamp_buffer(None)
d = dbg.cpu().numpy().reshape(nblk, 8).astype(np.int64)
pro = d[:, 1] - d[:, 0]; total = d[:, 2] - d[:, 0]; nit = d[:, 3]
wall = d[:, 7] - d[:, 7].min()
print("wall end span (100MHz ticks):", wall.max(), " end-time deciles:", [int(np.percentile(wall, q)) for q in range(0, 101, 10)])
for name, a in (("prologue fill", pro), ("total", total), ("items", nit), ("cycles/item", total / np.maximum(nit, 1))):
    print("%-16s mean %9.0f  p10 %9.0f  p50 %9.0f  p90 %9.0f  max %9.0f" % (name, a.mean(), np.percentile(a, 10), np.percentile(a, 50), np.percentile(a, 90), a.max()))
iss = d[:,4] >> 32; accu = d[:,4] & 0xffffffff; sto = d[:,5] >> 32; com = d[:,5] & 0xffffffff; bar = d[:,6] & 0xffffffff; walk = d[:,6] >> 32
for name, a in (("walk", walk), ("issue_fill", iss), ("accumulate", accu), ("store", sto), ("commit", com), ("barrier", bar)):
    print("%-12s per-WG total mean %9.0f  per item %8.0f" % (name, a.mean(), (a / np.maximum(nit,1)).mean()))
