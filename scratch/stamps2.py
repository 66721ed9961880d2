import sys, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from detectinblur_amd import blur_ops, _lib
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, torch.device("cuda", 0))
tables = blur_ops.compact_psfs(psfs, normalize=True)
idx = list(range(8))
l = _lib.lib(); l.dib_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]; l.dib_debug_set_stamp_buffer.restype = None
l.dib_debug_set_sched.argtypes=[ctypes.c_int]; l.dib_debug_set_sched.restype=None; l.dib_debug_set_sched(int(sys.argv[1]) if len(sys.argv)>1 else 0)
for _ in range(5): blur_ops.sparse_blur(list(images), idx, tables)
torch.cuda.synchronize()
nblk = 1024
dbg = torch.zeros(nblk * 16, dtype=torch.int64, device="cuda")
l.dib_debug_set_stamp_buffer(dbg.data_ptr())
blur_ops.sparse_blur(list(images), idx, tables); torch.cuda.synchronize()
l.dib_debug_set_stamp_buffer(None)
d = dbg.cpu().numpy().reshape(2, nblk, 8).astype(np.int64)[1]
ws, we = d[:,0], d[:,1]; t0 = ws.min()
print("WG start spread (10ns ticks): p0 %d p50 %d p100 %d ; end: p0 %d p50 %d p100 %d" % (0, np.percentile(ws-t0,50), (ws-t0).max(), (we-t0).min(), np.percentile(we-t0,50), (we-t0).max()))
nt = d[:,5]
print("tiles per WG: min %d mean %.2f max %d, total %d" % (nt.min(), nt.mean(), nt.max(), nt.sum()))
for name, a in (("decode", d[:,2]), ("tile", d[:,3]), ("barrier+ticket", d[:,4])):
    print("%-16s per WG mean %9.0f   per tile %8.0f" % (name, a.mean(), a.sum() / nt.sum()))
print("cycles per WG (sum of phases) mean %.0f ; wall per WG mean %.1f us" % ((d[:,2]+d[:,3]+d[:,4]).mean(), (we-ws).mean()/100))
