"""DIB_ACC_FAST16 against the oracle's restatement of its accumulation order (oracle.tap_order_vruns), bit for bit, on random PSFs of
every kind the compaction's grouping has a branch for (scatter of 1 .. 400 taps over spreads 1 .. 60, thick slanted bands, columns
longer than a group / a segment, corner taps, rasterised trajectories at all exposures) x random image shapes (all pad branches, half
tiles, single rows).  Also: the group records read back from the device's tables hold as many weights as the PSF has taps.
    python scratch/t_fast16_soak.py [cases]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, torch
import dib_oracle as O
from detectinblur_amd import _lib, blur_ops
from detectinblur_amd.motion_blur.generate_PSF import PSF
from detectinblur_amd.motion_blur.generate_trajectory import Trajectory

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rs = np.random.RandomState(2026)
np.random.seed(99)
dev = torch.device("cuda", 0)


def scatter():
    n, sp = int(rs.randint(1, 401)), int(rs.choice([1, 2, 4, 8, 12, 20, 40, 60]))
    a = np.zeros((128, 128), np.float64)
    a[np.clip(rs.randint(-sp, sp + 1, n) + 63, 0, 127), np.clip(rs.randint(-sp, sp + 1, n) + 63, 0, 127)] = rs.random_sample(n) + 0.02
    return "scatter n=%d spread=%d" % (n, sp), a


def band():
    length, thick = int(rs.randint(3, 50)), int(rs.randint(1, 7))
    a = np.zeros((128, 128), np.float64)
    x, y = 63.0 - length / 3, 63.0 - length / 2
    for k in range(length * 3):
        x += rs.uniform(-0.2, 0.5); y += rs.uniform(0.05, 0.45)
        for dy in range(thick):
            a[int(np.clip(y + dy, 0, 127)), int(np.clip(x, 0, 127))] += rs.random_sample() + 0.1
    return "band length=%d thick=%d" % (length, thick), a


def column():
    a = np.zeros((128, 128), np.float64)
    for _ in range(int(rs.randint(1, 4))):
        r0, n, c = int(rs.randint(0, 100)), int(rs.randint(2, 60)), int(rs.randint(0, 128))
        a[r0:min(128, r0 + n), c] += rs.random_sample(min(128, r0 + n) - r0) + 0.1
    return "columns", a


def corners():
    a = np.zeros((128, 128), np.float64)
    for _ in range(int(rs.randint(1, 9))):
        a[int(rs.choice([0, 1, 2, 125, 126, 127])), int(rs.choice([0, 1, 126, 127]))] += rs.random_sample() + 0.1
    return "corners", a


def trajectory():
    while True:
        tr = Trajectory(canvas=256, max_len=int(rs.choice([60, 96, 140])), expl=float(rs.choice([0.1, 0.01, 0.005, 0.001]))).fit()
        frac = float(rs.choice([1 / 100, 1 / 10, 1 / 5, 1 / 2, 1]))
        try:
            p = PSF(canvas=256, trajectory=tr, fraction=[frac]); p.fit(); p.centerPSF()
        except IndexError:           # a long trajectory that leaves the canvas (the reference raises the same): draw another
            continue
        return "trajectory fraction=%.2f" % frac, np.asarray(p.PSFs[0][64:192, 64:192], np.float64)


kinds = [scatter, scatter, band, column, corners, trajectory, trajectory]
t0 = time.time(); bad = 0; refused = 0; by = {}
for it in range(n_cases):
    name, a = kinds[it % len(kinds)]()
    if a.sum() <= 0:
        continue
    psf = O.to_half_like_torch(a / a.sum())
    C = int(rs.choice([1, 3])); H = int(rs.choice([1, 7, 33, 64, 97, 150])); W = int(rs.choice([5, 40, 129, 181, 260]))
    img = rs.random_sample((C, H, W)).astype(np.float16)
    if it % 5 == 0:
        img[:, rs.randint(0, H):, :] = 0                       # zero rows: 0 * w
    tabs = blur_ops.compact_psfs([torch.from_numpy(psf).to(dev)], normalize=bool(it % 2), vruns=True)
    try:
        got = blur_ops.sparse_blur([torch.from_numpy(img).to(dev)], [0], tabs, _lib.DIB_ACC_FAST16)[0].cpu().numpy()
    except _lib.DibError as e:                                  # the reference's own refusal (reflect padding of a 64-pixel side)
        assert "Padding size" in str(e), e
        refused += 1
        continue
    pn = O.normalize_psf(psf) if it % 2 else psf
    rows, cols, _ = O.taps_of(pn)
    want = O.manual_blur(img, pn, fma16=True, tap_order=O.tap_order_vruns(rows, cols))
    ok = np.array_equal(got.view(np.uint16).reshape(want.shape), want.view(np.uint16))
    groups = tabs.vgroups(0) if len(rows) <= 4096 else None
    if groups is not None and sum(len(w) for seg in groups for _off, w in seg) != len(rows):     # every tap in exactly one group
        ok = False
    k = name.split(" ")[0]; by[k] = by.get(k, 0) + 1
    if not ok:
        bad += 1
        print("MISMATCH case %d: %s, image %s, normalize %d, %d taps" % (it, name, (C, H, W), it % 2, len(rows)), flush=True)
print("%d cases (%d refused like the reference: padding >= side) in %.0f s (%s): %d mismatches -- DIB_ACC_FAST16 == oracle.manual_blur(fma16, tap_order_vruns) bit for bit" % (
    n_cases, refused, time.time() - t0, ", ".join("%s %d" % kv for kv in sorted(by.items())), bad))
sys.exit(1 if bad else 0)
