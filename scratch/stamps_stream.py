"""Per-workgroup phase breakdown of the persistent blur kernel (dib_debug_set_stamp_buffer): wave 0's shader
cycles in fill (wait for window loads + LDS write + 2 barriers), deferred store, look-ahead issue, tap loops,
ticket resolve; tiles per workgroup; XCD / list histogram."""
import sys, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from detectinblur_amd import blur_ops, _lib
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, torch.device("cuda", 0))
tables = blur_ops.compact_psfs(psfs, normalize=True)
idx = sorted(range(8), key=lambda k: -dicts[k]["psf_taps"])
ordered = [images[k] for k in idx]
l = _lib.lib(); l.dib_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]; l.dib_debug_set_stamp_buffer.restype = None
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if flags:
    l.dib_debug_set_variant.argtypes = [ctypes.c_int, ctypes.c_int]; l.dib_debug_set_variant(0, flags)
print('=== flags', flags)
for _ in range(20): outs = blur_ops.sparse_blur(list(ordered), idx, tables, mode)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(300): blur_ops.sparse_blur(list(ordered), idx, tables, mode)
torch.cuda.synchronize()
e0.record()
for _ in range(100): blur_ops.sparse_blur(list(ordered), idx, tables, mode)
e1.record(); e1.synchronize()
print("kernel us (100 back-to-back launches): %.2f" % (e0.elapsed_time(e1) * 10))
nblk = 2048
dbg = torch.zeros(nblk * 8, dtype=torch.int64, device="cuda")
l.dib_debug_set_stamp_buffer(dbg.data_ptr())
blur_ops.sparse_blur(list(ordered), idx, tables, mode); torch.cuda.synchronize()
l.dib_debug_set_stamp_buffer(None)
d = dbg.cpu().numpy().reshape(nblk, 8).astype(np.int64)
d = d[d[:, 5] != 0]
print("workgroups stamped:", len(d), " tiles total:", d[:, 6].sum(), " tiles per WG min/mean/max:", d[:, 6].min(), d[:, 6].mean(), d[:, 6].max())
tot = d[:, 5].astype(np.float64)
for name, k in (("fill(wait+write+barriers)", 0), ("store(deferred)", 1), ("issue(look-ahead)", 2), ("taps", 3)):
    a = d[:, k]
    print("  %-26s mean %9.0f cyc  %5.1f %% of WG life   per tile %7.0f" % (name, a.mean(), 100 * a.sum() / tot.sum(), a.sum() / d[:, 6].sum()))
print("  WG life cycles mean %.0f p10 %.0f p90 %.0f max %.0f" % (tot.mean(), np.percentile(tot, 10), np.percentile(tot, 90), tot.max()))
print("  lists:", np.unique(d[:, 7], return_counts=True))
