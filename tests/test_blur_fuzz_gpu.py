"""Randomised parity sweep of the HIP blur against the oracle: random image shapes (all three padding
regimes), channel counts, PSF canvases, tap counts / extents / weight magnitudes, batch compositions.
Seeded; DIB_FUZZ_CASES=<n> widens it (the default keeps the suite fast)."""
import os

import numpy as np
import pytest
import torch

import dib_oracle as O

pytestmark = pytest.mark.gpu


def _case(rs):
    K = 256 if rs.random_sample() < 0.15 else 128
    regime = rs.randint(0, 3)
    if K == 128 and regime == 0:      # zero padding: one side below 64
        H, W = (rs.randint(1, 64), rs.randint(1, 300)) if rs.random_sample() < 0.5 else (rs.randint(1, 200), rs.randint(1, 64))
    else:                             # reflect (> 64 both) or replicate (any size, 256 canvas)
        lo = 65 if K == 128 else 1
        H, W = rs.randint(lo, 260), rs.randint(lo, 700)
    C = rs.randint(1, 4)
    img = (rs.random_sample((C, H, W)) * rs.choice([1.0, 1.0, 255.0, 1e-3])).astype(np.float16)
    a = np.zeros((K, K), np.float64)
    n = rs.randint(1, 60)
    spread = rs.choice([1, 3, 8, 20, 40, K // 2 - 1])
    rr = np.clip(rs.randint(-spread, spread + 1, n) + K // 2 - 1, 0, K - 1)
    cc = np.clip(rs.randint(-spread, spread + 1, n) + K // 2 - 1, 0, K - 1)
    a[rr, cc] = rs.random_sample(n) + 0.01
    if rs.random_sample() < 0.2:
        a[rs.randint(0, K), rs.randint(0, K)] = 0.5      # a stray far tap: extra segments
    return img, O.to_half_like_torch(a * rs.choice([1.0, 0.37, 3.0]))


@pytest.mark.parametrize("seed", range(int(os.environ.get("DIB_FUZZ_BATCHES", "6"))))
def test_random_batches_bit_exact(seed):
    from detectinblur_amd.models import blur_functions as BF
    rs = np.random.RandomState(1000 + seed)
    per_batch = int(os.environ.get("DIB_FUZZ_CASES", "10"))
    cases = [_case(rs) for _ in range(per_batch)]
    for K in (128, 256):
        sub = [(i, p) for i, p in cases if p.shape[0] == K]
        if not sub:
            continue
        imgs = [i for i, _ in sub]
        psfs = [p for _, p in sub]
        dicts = [{"blurring": rs.random_sample() < 0.85} for _ in sub]
        want = [a.copy() for a in imgs]
        O.blur_image_list(want, dicts, psfs)
        got = [torch.from_numpy(a).cuda() for a in imgs]
        BF.blur_image_list(got, dicts, [torch.from_numpy(p).cuda() for p in psfs])
        for k, (g, w) in enumerate(zip(got, want)):
            # blurred entries come back squeezed (reference :69), skipped ones are the caller's own tensors
            assert g.shape == w.shape, (seed, K, k, imgs[k].shape, dicts[k])
            assert np.array_equal(g.cpu().numpy().view(np.uint16), w.view(np.uint16)), (seed, K, k, imgs[k].shape)


@pytest.mark.parametrize("seed", range(2))
@pytest.mark.parametrize("mode", ["fp32", "fma16"])
def test_random_images_other_accumulation_modes_bit_exact(seed, mode):
    """The same random population through DIB_ACC_FP32 / DIB_ACC_FMA16, against the oracle's restatement
    of each mode."""
    from detectinblur_amd import _lib
    from detectinblur_amd.models import blur_functions as BF
    rs = np.random.RandomState(5000 + seed)
    acc = _lib.DIB_ACC_FP32 if mode == "fp32" else _lib.DIB_ACC_FMA16
    for _ in range(int(os.environ.get("DIB_FUZZ_CASES", "10"))):
        img, psf = _case(rs)
        pn = O.normalize_psf(psf)
        want = O.manual_blur(img, pn, fp32_accumulate=(mode == "fp32"), fma16=(mode == "fma16"))
        got = BF.manual_blur(torch.from_numpy(img).cuda(), torch.from_numpy(pn).cuda(), acc_mode=acc).cpu().numpy()
        assert got.shape == want.shape and np.array_equal(got.view(np.uint16), want.view(np.uint16)), (seed, mode, img.shape)


@pytest.mark.parametrize("seed", range(2))
def test_random_boxes_expand_and_clamp(seed):
    """expand_targets on random boxes / PSF extents / image sizes against the oracle."""
    from detectinblur_amd import utils
    rs = np.random.RandomState(7000 + seed)
    for _ in range(20):
        img, psf = _case(rs)
        if psf.shape[0] != 128:
            continue
        C, H, W = img.shape
        n = rs.randint(0, 12)
        b = rs.uniform(-20, max(H, W) + 20, (n, 4)).astype(np.float32)
        want = O.expand_boxes(b.copy(), psf, H, W) if n else b
        t = [{"boxes": torch.from_numpy(b.copy()).cuda()}]
        utils.expand_targets(t, [{"blurring": True}], [torch.from_numpy(psf).cuda()], [torch.from_numpy(img).cuda()])
        assert np.array_equal(t[0]["boxes"].cpu().numpy(), want)


def test_random_trajectories_rasterise_bit_exact_on_device():
    """HIP rasteriser + centring + crop + Half conversion against the native host classes (themselves
    pinned to the reference's goldens) for random blur types and exposure fractions."""
    from detectinblur_amd import blur_ops
    from detectinblur_amd.motion_blur.generate_PSF import PSF
    from detectinblur_amd.motion_blur.generate_trajectory import Trajectory
    rs = np.random.RandomState(31)
    np.random.seed(4321)
    trajs, fracs, want64, want16 = [], [], [], []
    for _ in range(24):
        expl = float(rs.choice([0.005, 0.001, 0.00005, 0.01]))
        frac = float(rs.choice([1 / 25, 1 / 18, 1 / 10, 1 / 5, 1 / 2, 1, rs.uniform(0.02, 1.0)]))
        tr = Trajectory(canvas=256, max_len=96, expl=expl).fit()
        p = PSF(canvas=256, trajectory=tr, fraction=[frac])
        p.fit()
        p.centerPSF()
        crop = np.ascontiguousarray(p.PSFs[0][64:192, 64:192])
        trajs.append(np.asarray(tr.x, dtype=np.complex128)); fracs.append(frac)
        want64.append(crop); want16.append(O.to_half_like_torch(crop))
    p64, p16 = blur_ops.rasterize_psfs(torch.from_numpy(np.stack(trajs)), fracs, canvas=256, center=True)
    assert np.array_equal(p64.cpu().numpy().view(np.uint64), np.stack(want64).view(np.uint64))
    assert np.array_equal(p16.cpu().numpy().view(np.uint16), np.stack(want16).view(np.uint16))


def test_compaction_general_path_dense_and_many_taps():
    """PSFs that leave the fast path of the compaction kernel: more non-zeros than its LDS stage holds
    (> 4096), a fully dense PSF, and weights that underflow in the division -- all against the oracle."""
    from detectinblur_amd import blur_ops
    rs = np.random.RandomState(99)
    cases = []
    a = np.zeros((128, 128), np.float16); idx = rs.choice(128 * 128, 5000, replace=False)
    a.flat[idx] = (rs.random_sample(5000) + 0.05).astype(np.float16); cases.append(a)            # > STAGE_TAPS
    cases.append((rs.random_sample((128, 128)) + 0.01).astype(np.float16))                       # dense
    b = np.zeros((128, 128), np.float16); b[10, 10] = 60000.0; b[rs.randint(0, 128, 300), rs.randint(0, 128, 300)] = 1e-7
    b[10, 10] = 60000.0; cases.append(b)                                                           # underflow: taps vanish
    c = np.zeros((256, 256), np.float16); idx = rs.choice(256 * 256, 4097, replace=False)
    c.flat[idx] = (rs.random_sample(4097) + 0.05).astype(np.float16); cases.append(c)            # 256 canvas, just over the stage
    for psf in cases:
        tabs = blur_ops.compact_psfs([torch.from_numpy(psf).cuda()], normalize=True)
        rr, cc, ww = O.taps_of(O.normalize_psf(psf))
        r, c_, w = tabs.taps(0)
        assert tabs.header(0)[0] == len(rr)
        assert np.array_equal(r.numpy(), rr) and np.array_equal(c_.numpy(), cc)
        assert np.array_equal((w.numpy() & 0xffff).astype(np.uint16), ww.view(np.uint16))
        if len(rr):
            assert tuple(tabs.header(0)[1:5]) == (rr.min(), rr.max(), cc.min(), cc.max())
        segs = tabs.segments(0)
        assert segs[0][0] == 0 and segs[-1][1] == len(rr) and all(s[1] == t[0] for s, t in zip(segs, segs[1:]))
