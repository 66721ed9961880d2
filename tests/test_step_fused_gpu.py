"""The blur step as ONE launch (csrc/dib_blur.hip: blur_step_f16_kernel -- the grid's first workgroups compact the batch's
PSFs, the blur workgroups behind them wait for a counter) against the two-launch path, the stand-alone compaction kernel and
the oracle, bit for bit (reference models/blur_functions.py:92-100, `blur_image_list`).

What can go wrong in such a hand-off is a blur workgroup reading a table line that is not the compaction's final one: every
case below therefore changes the PSFs from step to step (a stale table of an earlier step gives other pixels), runs many steps
back to back without a host synchronisation, and compares every output element."""
import ctypes

import numpy as np
import pytest
import torch

import dib_oracle as O

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _lib_hooks():
    from detectinblur_amd import _lib
    l = _lib.lib()
    l.dib_debug_set_step_fused.argtypes = [ctypes.c_int]
    l.dib_debug_set_step_fused.restype = None
    l.dib_debug_compact_wg256.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    l.dib_debug_compact_wg256.restype = ctypes.c_int
    return l


def _psf(rs, n, spread, K=128):
    a = np.zeros((K, K), np.float64)
    c = K // 2 - 1
    a[np.clip(rs.randint(-spread, spread + 1, n) + c, 0, K - 1), np.clip(rs.randint(-spread, spread + 1, n) + c, 0, K - 1)] = rs.random_sample(n) + 0.05
    return O.to_half_like_torch(a)


def _table_parts(buf, i, words, K=128):
    """Every word of table i that the blur or the box growth reads: header, row pointers, taps, segments, both offset
    arrays with their 8 zero words (what lies behind them is scratch of whichever kernel wrote the table)."""
    t = buf[i * words:(i + 1) * words].cpu().numpy()
    ntaps, nsegs = int(t[0]), int(t[7])
    taps_off = (8 + K + 1 + 3) & ~3
    segs_off = taps_off + 2 * K * K
    lt_off = segs_off + 4 * K * K
    ltq_off = lt_off + K * K + 8
    return (t[:8].copy(), t[8:8 + K + 1].copy(), t[taps_off:taps_off + 2 * ntaps].copy(), t[segs_off:segs_off + 4 * nsegs].copy(),
            t[lt_off:lt_off + ntaps + 8].copy(), t[ltq_off:ltq_off + ntaps + 8].copy())


def _special_psfs():
    rs = np.random.RandomState(5)
    out = []
    for n, spread in ((1, 0), (3, 1), (17, 6), (60, 20), (200, 45), (300, 60)):
        out.append(_psf(rs, n, spread))
    z = np.zeros((128, 128), np.float16)
    out.append(z.copy())                                                   # sum 0: every element a NaN tap (general path)
    a = z.copy(); a[60, 60] = 60000.0; a[61, 61] = 6e-8; a[70, 3] = 1.0     # 6e-8 / 60000 underflows to 0: that tap vanishes
    out.append(a)
    b = z.copy(); b[10, 10] = 60000.0; b[5, 5] = 6e-8; b[5, 6] = 6e-8; b[127, 127] = 6e-8; b[0, 0] = 3.0
    out.append(b)
    d = (rs.random_sample((128, 128)) * 0.5 + 0.01).astype(np.float16)     # dense: 16,384 taps (> the LDS stage)
    out.append(d)
    e = z.copy(); e[::2, ::3] = 0.25                                        # 2,752 taps: more than the stage, many pieces
    out.append(e)
    f = z.copy(); f[:, 64] = 0.5                                            # one column: a piece hit in every (wave, i) pair
    out.append(f)
    g = z.copy(); g[64, :] = 0.125                                          # one row
    out.append(g)
    h = z.copy(); h[0, 0] = 1.0; h[127, 127] = 1.0                          # the corners
    out.append(h)
    k = z.copy(); k[40:48, 30:100] = 0.01                                   # 560 taps in 70 pieces per row band
    out.append(k)
    return out


@pytest.mark.parametrize("normalize", [1, 0])
def test_wg256_compaction_writes_the_standalone_kernels_tables(normalize):
    from detectinblur_amd import _lib, blur_ops
    l = _lib_hooks()
    psfs = _special_psfs()
    t_psfs = [_dev(p) for p in psfs]
    want = blur_ops.compact_psfs(t_psfs, normalize=bool(normalize))
    got = blur_ops.TapTables(128, len(psfs), t_psfs[0].device)
    got.buf.fill_(-7)
    _lib.check(l.dib_debug_compact_wg256(_lib.ptr_array([p.data_ptr() for p in t_psfs]), len(psfs), normalize, got.buf.data_ptr(),
                                         blur_ops._stream(t_psfs[0].device)))
    torch.cuda.synchronize()
    for i in range(len(psfs)):
        for a, b, name in zip(_table_parts(got.buf, i, got.words), _table_parts(want.buf, i, want.words),
                              ("header", "rowptr", "taps", "segments", "ltaps", "ltaps_q")):
            assert np.array_equal(a, b), (i, name)


def _case(rs, count, big=False):
    imgs, psfs = [], []
    for i in range(count):
        C = (3, 1, 2)[i % 3]
        H, W = (66 + (i * 11) % 70, 70 + (i * 29) % 200) if not big else (300 + 37 * (i % 5), 500 + 61 * (i % 4))
        imgs.append(rs.random_sample((C, H, W)).astype(np.float16))
        psfs.append(_psf(rs, 3 + (i * 7) % 50, 1 + (i * 5) % 45))
    return imgs, psfs


def test_fused_step_equals_two_launch_step_and_oracle():
    from detectinblur_amd import blur_ops
    l = _lib_hooks()
    rs = np.random.RandomState(41)
    try:
        for count, big in ((1, False), (9, False), (8, True), (32, False), (5, True)):
            imgs, psfs = _case(rs, count, big)
            t_imgs, t_psfs = [_dev(a) for a in imgs], [_dev(p) for p in psfs]
            l.dib_debug_set_step_fused(1)
            one = blur_ops.blur_step(t_imgs, list(range(count)), t_psfs)
            l.dib_debug_set_step_fused(0)
            two = blur_ops.blur_step(t_imgs, list(range(count)), t_psfs)
            want = [a.copy() for a in imgs]
            O.blur_image_list(want, [{"blurring": True}] * count, psfs)
            for a, b, w in zip(one, two, want):
                assert torch.equal(a, b)
                assert np.array_equal(a.cpu().numpy().view(np.uint16).reshape(w.shape), w.view(np.uint16))
    finally:
        l.dib_debug_set_step_fused(1)


def test_fused_step_with_special_psfs_skipped_images_and_shared_tables():
    """PSFs of the compaction's rare paths (sum 0, vanishing taps, more taps than the LDS stage), images that share a PSF,
    skipped images (table_index < 0), FMA16 accumulation."""
    from detectinblur_amd import _lib, blur_ops
    l = _lib_hooks()
    rs = np.random.RandomState(42)
    psfs = _special_psfs()[:9] + [_special_psfs()[10]]
    t_psfs = [_dev(p) for p in psfs]
    imgs = [rs.random_sample(((3, 1)[i % 2], 70 + 9 * i, 90 + 13 * i)).astype(np.float16) for i in range(14)]
    t_imgs = [_dev(a) for a in imgs]
    index = [i % len(psfs) if i % 5 != 3 else -1 for i in range(14)]
    try:
        for acc in (_lib.DIB_ACC_BITEXACT, _lib.DIB_ACC_FMA16):
            l.dib_debug_set_step_fused(1)
            one = blur_ops.blur_step(t_imgs, index, t_psfs, acc_mode=acc)
            l.dib_debug_set_step_fused(0)
            two = blur_ops.blur_step(t_imgs, index, t_psfs, acc_mode=acc)
            for i, (a, b) in enumerate(zip(one, two)):
                if index[i] < 0:
                    assert a is t_imgs[i]
                else:
                    assert torch.equal(a.view(torch.int16), b.view(torch.int16)), i     # NaNs compare as bits
    finally:
        l.dib_debug_set_step_fused(1)


def test_many_fused_steps_back_to_back_with_new_psfs_every_step():
    """300 steps queued without a host synchronisation, on two alternating table buffers, every step with other PSFs and a
    different number of them, a long-running kernel in front now and then (the launch then starts on a busy chip): a blur
    workgroup that read a table too early, or a table line left over from two steps ago, gives other pixels."""
    from detectinblur_amd import blur_ops
    l = _lib_hooks()
    rs = np.random.RandomState(43)
    pool_imgs = [_dev(rs.random_sample((3, 128 + 32 * (i % 4), 200 + 64 * (i % 3))).astype(np.float16)) for i in range(12)]
    pool_psfs = [_dev(_psf(rs, 2 + (7 * i) % 90, 1 + (5 * i) % 50)) for i in range(40)]
    big = torch.rand(4096, 4096, device="cuda")
    plan = []
    for s in range(300):
        n = 1 + (s * 7) % 12
        plan.append(([pool_imgs[(s + k) % 12] for k in range(n)], [pool_psfs[(3 * s + 5 * k) % 40] for k in range(n)]))
    try:
        l.dib_debug_set_step_fused(0)
        want = [blur_ops.blur_step(im, list(range(len(im))), ps) for im, ps in plan]
        torch.cuda.synchronize()
        l.dib_debug_set_step_fused(1)
        got = []
        for s, (im, ps) in enumerate(plan):
            if s % 37 == 5:
                big = (big @ big) * 1e-3
            got.append(blur_ops.blur_step(im, list(range(len(im))), ps, psfs_complete=True))
        torch.cuda.synchronize()
        for s, (g, w) in enumerate(zip(got, want)):
            for a, b in zip(g, w):
                assert torch.equal(a, b), s
    finally:
        l.dib_debug_set_step_fused(1)


def test_fused_step_full_size_batch_repeated():
    """The BASELINE launch (8 x 3 x 800 x 1333: 6,600 blur workgroups behind 8 compacting ones), 40 steps in a row with the
    PSFs rotated between the images: every output against the two-launch path."""
    from detectinblur_amd import blur_ops
    l = _lib_hooks()
    rs = np.random.RandomState(44)
    imgs = [torch.rand(3, 800, 1333, device="cuda").half() for _ in range(8)]
    psfs = [_dev(_psf(rs, 15 + 11 * i, 3 + 2 * i)) for i in range(8)]
    try:
        l.dib_debug_set_step_fused(0)
        want = [blur_ops.blur_step(imgs, list(range(8)), psfs[r:] + psfs[:r]) for r in range(8)]
        torch.cuda.synchronize()
        l.dib_debug_set_step_fused(1)
        bad = []
        for s in range(40):
            one = blur_ops.blur_step(imgs, list(range(8)), psfs[s % 8:] + psfs[:s % 8], psfs_complete=True)
            bad.append(torch.stack([(a != b).any() for a, b in zip(one, want[s % 8])]).any())     # queued: no host sync in the loop
            del one
        assert not bool(torch.stack(bad).any())
    finally:
        l.dib_debug_set_step_fused(1)
