import sys, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from detectinblur_amd import blur_ops, _lib
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, torch.device("cuda", 0))
tables = blur_ops.compact_psfs(psfs, normalize=True)
idx = list(range(8))
l = _lib.lib(); l.dib_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]; l.dib_debug_set_stamp_buffer.restype = None
ref = None
for nw in (8,):
    for _ in range(5): outs = blur_ops.sparse_blur(list(images), idx, tables)
    torch.cuda.synchronize()
    if ref is None: ref = [o.clone() for o in outs]
    else: print("variants identical:", all(torch.equal(a, b) for a, b in zip(ref, outs)))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(30):
        e0.record(); blur_ops.sparse_blur(list(images), idx, tables); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); print("NW=%d  kernel ms median %.4f min %.4f" % (nw, ts[len(ts)//2], ts[0]))
    nblk = 4096
    dbg = torch.zeros(nblk * 8, dtype=torch.int64, device="cuda")
    l.dib_debug_set_stamp_buffer(dbg.data_ptr())
    blur_ops.sparse_blur(list(images), idx, tables); torch.cuda.synchronize()
    l.dib_debug_set_stamp_buffer(None)
    d = dbg.cpu().numpy().reshape(nblk, 8).astype(np.int64)
    d = d[d[:, 4] != 0]
    nblk = len(d); print("   workgroups stamped:", nblk)
    print("   segments per PSF:", [len(tables.segments(i)) for i in range(8)], "taps:", [tables.header(i)[0] for i in range(8)])
    for name, a in (("fill", d[:,1]-d[:,0]), ("accumulate", d[:,2]-d[:,1]), ("rest+store", d[:,3]-d[:,2]), ("total", d[:,3]-d[:,0])):
        print("   %-12s mean %8.0f p10 %8.0f p50 %8.0f p90 %8.0f" % (name, a.mean(), np.percentile(a,10), np.percentile(a,50), np.percentile(a,90)))
    ws, we = d[:,4], d[:,5]
    t0 = ws.min(); ws = ws - t0; we = we - t0
    print("   wall span (10ns ticks):", we.max(), "WG duration ticks mean", (we-ws).mean())
    # concurrency over time
    T = int(we.max()) + 1
    ev = np.zeros(T + 2); np.add.at(ev, ws, 1); np.add.at(ev, we + 1, -1)
    conc = np.cumsum(ev)[:T]
    print("   resident WGs: mean %.0f max %.0f ; per-decile-of-time:" % (conc.mean(), conc.max()), [int(conc[int(T*q/10)]) for q in range(10)])
    hw = d[:,6] & 0xffffffff; xcc = d[:,6] >> 32
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
    cuid = xcc * 1000 + se * 100 + sh * 16 + cu
    u, cnt = np.unique(cuid, return_counts=True)
    print("   distinct CUs seen:", len(u), " WGs per CU min/mean/max:", cnt.min(), cnt.mean(), cnt.max())
    # max concurrent on one CU
    best = 0
    for c in u[:40]:
        m = cuid == c
        e2 = np.zeros(T + 2); np.add.at(e2, ws[m], 1); np.add.at(e2, we[m] + 1, -1)
        best = max(best, np.cumsum(e2).max())
    print("   max concurrent WGs on a CU (first 40 CUs):", best)
