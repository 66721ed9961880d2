import sys, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from detectinblur_amd import blur_ops, _lib
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, torch.device("cuda", 0))
tables = blur_ops.compact_psfs(psfs, normalize=True)
idx = sorted(range(8), key=lambda k: -dicts[k]["psf_taps"])
ordered = [images[k] for k in idx]
l = _lib.lib(); l.dib_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]; l.dib_debug_set_waves.argtypes = [ctypes.c_int]
l.dib_debug_set_waves(2)
for _ in range(50): blur_ops.sparse_blur(list(ordered), idx, tables)
nblk = 8192
dbg = torch.zeros(nblk * 8, dtype=torch.int64, device="cuda")
l.dib_debug_set_stamp_buffer(dbg.data_ptr())
blur_ops.sparse_blur(list(ordered), idx, tables); torch.cuda.synchronize()
l.dib_debug_set_stamp_buffer(None)
d = dbg.cpu().numpy().reshape(nblk, 8).astype(np.int64)
d = d[d[:, 5] != 0]
print("workgroups:", len(d))
names = ["prologue (start -> before loads)", "issue loads", "wait + LDS write + barrier", "taps", "stores + drain"]
tot = (d[:, 5] - d[:, 0]).astype(np.float64)
for k, n in enumerate(names):
    a = d[:, k + 1] - d[:, k]
    print("  %-34s mean %8.0f cyc  p10 %8.0f p50 %8.0f p90 %8.0f  (%4.1f %%)" % (n, a.mean(), np.percentile(a, 10), np.percentile(a, 50), np.percentile(a, 90), 100 * a.sum() / tot.sum()))
print("  WG life mean %.0f cycles; span of the launch (first start -> last end): %.0f cycles" % (tot.mean(), d[:, 5].max() - d[:, 0].min()))
w = d[:, 6]; print("  wall span (100 MHz ticks):", w.max() - w.min())
