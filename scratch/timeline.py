"""Residency timeline of the 128-wide blur kernel: per workgroup the 100 MHz wall-clock time it started and ended, the
XCD / shader engine / CU it ran on.  Needs a library built with -DDIB_TIMELINE (see scratch/README.md):
    DIB_LIB=detectinblur_amd/libdib_hip_tl.so python scratch/timeline.py
Prints: slots busy over time (per microsecond), the idle share before / between / after workgroups per CU."""
import os, sys, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
from detectinblur_amd import _lib
if os.environ.get("DIB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["DIB_LIB"])
import bench
from detectinblur_amd import blur_ops
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, torch.device("cuda", 0))
tables = blur_ops.compact_psfs(psfs, normalize=True)
idx = sorted(range(8), key=lambda k: -dicts[k]["psf_taps"])
ordered = [images[k] for k in idx]
l = _lib.lib(); l.dib_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
for _ in range(50): blur_ops.sparse_blur(list(ordered), idx, tables)
n = 16384
for rep in range(3):
    dbg = torch.zeros(n * 8, dtype=torch.int64, device="cuda")
    l.dib_debug_set_stamp_buffer(dbg.data_ptr())
    for _ in range(2): blur_ops.sparse_blur(list(ordered), idx, tables)
    torch.cuda.synchronize()
    l.dib_debug_set_stamp_buffer(None)
    d = dbg.cpu().numpy().reshape(n, 8)
    d = d[d[:, 1] != 0]
    t0 = d[:, 0].min()
    b, e = (d[:, 0] - t0) / 100.0, (d[:, 1] - t0) / 100.0          # microseconds
    hw, xcc = d[:, 2], d[:, 3] & 0xf
    cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)
    span = e.max()
    print("launch %d: %d workgroups on %d CUs, span %.2f us, mean life %.2f us, busy slot-time / (2048 slots x span) = %.3f" % (
        rep, len(d), len(np.unique(cu)), span, (e - b).mean(), (e - b).sum() / (2048 * span)))
    grid = np.arange(0, span, 1.0)
    occ = [(int(((b <= t) & (e > t)).sum())) for t in grid]
    print("  resident workgroups at t = 0,1,2.. us:", " ".join(str(o) for o in occ))
    last_start = b.max()
    print("  last workgroup starts at %.2f us; from then on the chip only drains (%.2f us)" % (last_start, span - last_start))
    # per CU: how many workgroups, first start, last end
    per = {}
    for c, bb, ee in zip(cu, b, e): per.setdefault(int(c), []).append((bb, ee))
    cnt = np.array([len(v) for v in per.values()]); ends = np.array([max(x[1] for x in v) for v in per.values()])
    print("  workgroups per CU: min %d mean %.1f max %d; CU finish time: p10 %.1f p50 %.1f p90 %.1f max %.1f us" % (
        cnt.min(), cnt.mean(), cnt.max(), *np.percentile(ends, [10, 50, 90]), ends.max()))
    for t in (5.0, 20.0, 35.0):
        live = (b <= t) & (e > t)
        c, k = np.unique(np.unique(cu[live], return_counts=True)[1], return_counts=True)
        x, kx = np.unique(xcc[live], return_counts=True)
        print("  t = %4.1f us: resident per CU histogram %s; per XCD %s" % (t, dict(zip(c.tolist(), k.tolist())), kx.tolist()))
    # turnaround: per CU, gap between a workgroup's end and the next start after it
    gaps = []
    for v in per.values():
        starts = np.sort([x[0] for x in v]); ends_ = np.sort([x[1] for x in v])
        for ee in ends_:
            j = np.searchsorted(starts, ee)
            if j < len(starts): gaps.append(starts[j] - ee)
    gaps = np.array(gaps)
    print("  end -> next start on the same CU: p10 %.2f p50 %.2f p90 %.2f us" % tuple(np.percentile(gaps, [10, 50, 90])))
    hwf = {"wave": hw & 0xf, "simd": (hw >> 4) & 3, "pipe": (hw >> 6) & 3, "cu": (hw >> 8) & 0xf, "sh": (hw >> 12) & 1, "se": (hw >> 13) & 7}
    print("  HW_ID field ranges:", {k: (int(v.min()), int(v.max())) for k, v in hwf.items()})
