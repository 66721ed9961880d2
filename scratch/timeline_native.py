"""Residency timeline of the default blur kernel on the NATIVE-size ragged batch (bench.COCO_NATIVE_SIZES) and, for comparison,
on the BASELINE batch: per workgroup start / end (100 MHz wall clock) from a -DDIB_TIMELINE build:
    DIB_HIP_LIB=scratch/libdib_hip_tl.so python scratch/timeline_native.py
Prints the launch's span, workgroup lifetimes, start / end distributions and resident workgroups per microsecond."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from detectinblur_amd import _lib, blur_ops

dev = torch.device("cuda", 0)
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev)
tables = blur_ops.compact_psfs(psfs, normalize=True, vruns=True)
idx = sorted(range(8), key=lambda k: -dicts[k]["psf_taps"])
native = [torch.rand(3, h, w, generator=torch.Generator().manual_seed(31 + i)).half().to(dev) for i, (h, w) in enumerate(bench.COCO_NATIVE_SIZES)]
l = _lib.lib(); l.dib_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
MODE = {"bitexact": _lib.DIB_ACC_BITEXACT, "fma16": _lib.DIB_ACC_FMA16, "fast16": 3}[os.environ.get("DIB_TL_MODE", "bitexact")]   # DIB_TL_MODE=fma16: the tolerance mode's timeline


def timeline(name, imgs):
    ordered = [imgs[k] for k in idx]
    for _ in range(50): blur_ops.sparse_blur(list(ordered), idx, tables, MODE)
    n = 16384
    for rep in range(3):
        dbg = torch.zeros(n * 8, dtype=torch.int64, device="cuda")
        l.dib_debug_set_stamp_buffer(dbg.data_ptr())
        for _ in range(2): blur_ops.sparse_blur(list(ordered), idx, tables, MODE)
        torch.cuda.synchronize()
        l.dib_debug_set_stamp_buffer(None)
        d = dbg.cpu().numpy().reshape(n, 8)
        rec = np.nonzero(d[:, 1] != 0)[0]
        d = d[rec]
        t0 = d[:, 0].min()
        b, e = (d[:, 0] - t0) / 100.0, (d[:, 1] - t0) / 100.0
        hw, xcc = d[:, 2], d[:, 3] & 0xf
        img = (d[:, 3] >> 32).astype(np.int64)          # image in descriptor order (heaviest first)
        cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)
        span = e.max()
        print("%s launch %d: %d workgroups on %d CUs, span %.2f us, life mean %.2f p10 %.2f p90 %.2f us" % (
            name, rep, len(d), len(np.unique(cu)), span, (e - b).mean(), *np.percentile(e - b, [10, 90])))
        print("  start p0/50/90/100: %s   end p0/10/50/90/100: %s" % (np.round(np.percentile(b, [0, 50, 90, 100]), 2).tolist(),
                                                                   np.round(np.percentile(e, [0, 10, 50, 90, 100]), 2).tolist()))
        f, tp = (d[:, 4] - t0) / 100.0, (d[:, 5] - t0) / 100.0
        taps = np.array([dicts[idx[i]]["psf_taps"] for i in range(8)])
        print("  wave 0 of a workgroup: first window ready p10/50/90/100 %s; taps done p10/50/90/100 %s" % (np.round(np.percentile(f, [10, 50, 90, 100]), 2).tolist(), np.round(np.percentile(tp, [10, 50, 90, 100]), 2).tolist()))
        print("  phases (mean us): start -> window ready %.2f, -> taps done %.2f, -> end %.2f" % ((f - b).mean(), (tp - f).mean(), (e - tp).mean()))
        pr, li = (d[:, 6] - t0) / 100.0, (d[:, 7] - t0) / 100.0
        print("  per image p50 of: start | prologue done | loads issued | window ready | taps done | end")
        for i in range(8):
            m = img == i
            print("    image %d (%d taps): %.2f | %.2f | %.2f | %.2f | %.2f | %.2f" % (i, taps[i], np.median(b[m]), np.median(pr[m]), np.median(li[m]), np.median(f[m]), np.median(tp[m]), np.median(e[m])))
        print("  per image: window ready p50, taps p50:", " ".join("(%.1f, %.1f)" % (np.median(f[img == i]), np.median((tp - f)[img == i])) for i in range(8)))
        grid = np.arange(0, span, 1.0)
        print("  resident at t = 0,1,2.. us:", " ".join(str(int(((b <= t) & (e > t)).sum())) for t in grid))
        cnt = np.unique(cu, return_counts=True)[1]
        print("  workgroups per CU: min %d mean %.1f max %d" % (cnt.min(), cnt.mean(), cnt.max()))
        taps = np.array([dicts[idx[i]]["psf_taps"] for i in range(8)])
        print("  per image (taps, workgroups, end p50 / max):", " ".join("(%d, %d, %.1f / %.1f)" % (taps[i], (img == i).sum(), np.median(e[img == i]), e[img == i].max()) for i in range(8)))
        ucu = np.unique(cu)
        load = np.array([taps[img[cu == c]].sum() for c in ucu]); n_wg = np.array([(cu == c).sum() for c in ucu]); fin = np.array([e[cu == c].max() for c in ucu])
        print("  per CU: taps-load min %d mean %.0f max %d; corr(load, finish) %.2f; corr(n_wg, finish) %.2f" % (load.min(), load.mean(), load.max(), np.corrcoef(load, fin)[0, 1], np.corrcoef(n_wg, fin)[0, 1]))
        for k in sorted(set(n_wg.tolist())):
            print("    CUs with %d workgroups: %d, finish mean %.1f us, taps-load mean %.0f" % (k, (n_wg == k).sum(), fin[n_wg == k].mean(), load[n_wg == k].mean()))
        if name == "native" and rep == 0:      # where the dispatcher puts the 1-D grid's workgroups: list x = b & 7, entry t = b >> 3
            for x in (0, 3):
                sel = np.nonzero((rec & 7) == x)[0]
                order = np.argsort(rec[sel] >> 3)
                seq = cu[sel][order]
                print("  list %d (XCD id %s): CU (low 8 bits) of entries 0..: %s" % (x, sorted(set(xcc[sel].tolist())), " ".join("%02x" % (c & 0xff) for c in seq[:80])))
                same = [(seq[t] == seq[t + 32]) for t in range(len(seq) - 32)]
                first = {}
                for t, c in enumerate(seq): first.setdefault(int(c), []).append(t)
                print("    entries t and t + 32 on one CU: %d of %d; entries per CU: %s" % (sum(same), len(same), sorted(len(v) for v in first.values())))
                print("    entry lists of the first CUs:", [first[int(c)] for c in seq[:6]])
        o = np.argsort(fin)[-5:]
        for c in o: print("    late CU %x: finish %.1f, images %s" % (ucu[c], fin[c], sorted(img[cu == ucu[c]].tolist())))


timeline("native", native)
timeline("baseline", images)
# event-timed loop for the same launches (this build, for scale)
for name, imgs in (("native", native), ("baseline", images)):
    ordered = [imgs[k] for k in idx]
    print(name, "kernel_ms (event loop of 200)", sorted(bench.kernel_time_ms(lambda k: blur_ops.sparse_blur(list(ordered), idx, tables, MODE), 200) for _ in range(5))[2])
