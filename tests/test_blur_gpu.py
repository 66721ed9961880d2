"""GPU parity tests: the HIP path (through the C ABI, include/dib.h) against the CPU oracle and
against the golden vectors captured from the real reference.  Integer / fp16 / fp64 results are
compared BIT FOR BIT; nothing here reads /root/reference."""
import hashlib

import numpy as np
import pytest
import torch

import dib_oracle as O
import golden_inputs as GI

pytestmark = pytest.mark.gpu


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view({2: np.uint16, 4: np.uint32, 8: np.uint64}[a.dtype.itemsize])


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# ------------------------------------------------------------------ tap compaction

@pytest.mark.parametrize("param", GI.PARAMS)
@pytest.mark.parametrize("fi", range(len(GI.FRACTIONS)))
def test_compaction_matches_reference_nonzero(golden, param, fi):
    from detectinblur_amd import blur_ops
    half = GI.golden_psf(param, fi, "half")
    tabs = blur_ops.compact_psfs([_dev(half)], normalize=True)
    key = "psf_p%g_f%d" % (param, fi)
    r, c, w = tabs.taps(0)
    want_rc = golden.psf[key + "_norm_rc"].astype(np.int64)
    assert np.array_equal(np.stack([r.numpy(), c.numpy()], 1), want_rc)       # indices bit-exact, row-major
    assert np.array_equal((w.numpy() & 0xffff).astype(np.uint16), golden.psf[key + "_norm_w"])
    ntaps, rmin, rmax, cmin, cmax = tabs.header(0)
    assert ntaps == len(want_rc)
    assert (rmin, rmax, cmin, cmax) == (want_rc[:, 0].min(), want_rc[:, 0].max(), want_rc[:, 1].min(), want_rc[:, 1].max())
    assert np.uint16(tabs.buf[6].item() & 0xffff) == golden.psf[key + "_sum"][0]


def test_compaction_batch_and_unnormalised():
    from detectinblur_amd import blur_ops
    rs = np.random.RandomState(5)
    psfs = []
    for k in range(5):
        a = np.zeros((128, 128), np.float16)
        n = 1 + 40 * k
        a[rs.randint(0, 128, n), rs.randint(0, 128, n)] = rs.random_sample(n).astype(np.float16)
        psfs.append(a)
    tabs = blur_ops.compact_psfs(_dev(np.stack(psfs)), normalize=False)
    for k, a in enumerate(psfs):
        rr, cc = np.nonzero(a)
        r, c, w = tabs.taps(k)
        assert np.array_equal(r.numpy(), rr) and np.array_equal(c.numpy(), cc)
        assert np.array_equal((w.numpy() & 0xffff).astype(np.uint16), a[rr, cc].view(np.uint16))
    # 256 canvas, fp32
    a = np.zeros((256, 256), np.float32)
    a[rs.randint(0, 256, 300), rs.randint(0, 256, 300)] = rs.random_sample(300).astype(np.float32)
    a[255, 255] = 0.5
    tabs = blur_ops.compact_psfs([_dev(a)], normalize=False)
    rr, cc = np.nonzero(a)
    r, c, w = tabs.taps(0)
    assert np.array_equal(r.numpy(), rr) and np.array_equal(c.numpy(), cc)
    assert np.array_equal(w.numpy().astype(np.uint32), a[rr, cc].view(np.uint32))


def test_half_sum_is_exactly_rounded():
    from detectinblur_amd import blur_ops
    rs = np.random.RandomState(9)
    for scale in (1e-5, 1e-2, 1.0, 30.0):
        a = (rs.random_sample((128, 128)) * scale * (rs.random_sample((128, 128)) < 0.05)).astype(np.float16)
        tabs = blur_ops.compact_psfs([_dev(a)], normalize=True)
        assert np.uint16(tabs.buf[6].item() & 0xffff) == O.half_sum_exact(a).view(np.uint16)
        r, c, w = tabs.taps(0)
        rr, cc, ww = O.taps_of(O.normalize_psf(a))
        assert np.array_equal(r.numpy(), rr) and np.array_equal(c.numpy(), cc)
        assert np.array_equal((w.numpy() & 0xffff).astype(np.uint16), ww.view(np.uint16))


def test_all_zero_psf_normalises_to_nan_taps_like_the_reference():
    """psf / psf.sum() of an all-zero PSF is NaN everywhere and NaN counts as non-zero (torch.nonzero):
    16,384 taps.  Also a PSF whose tiny entries underflow to zero in the division loses those taps."""
    from detectinblur_amd import blur_ops
    tabs = blur_ops.compact_psfs([_dev(np.zeros((128, 128), np.float16))], normalize=True)
    assert tabs.header(0)[0] == 128 * 128
    r, c, w = tabs.taps(0)
    assert np.array_equal(r.numpy(), np.repeat(np.arange(128), 128)) and np.array_equal(c.numpy(), np.tile(np.arange(128), 128))
    assert np.isnan((w.numpy() & 0xffff).astype(np.uint16).view(np.float16)).all()
    a = np.zeros((128, 128), np.float16)
    a[60, 60] = 60000.0; a[61, 61] = 6e-8; a[70, 3] = 1.0          # 6e-8 / 60000 underflows to 0 in fp16
    tabs = blur_ops.compact_psfs([_dev(a)], normalize=True)
    rr, cc, ww = O.taps_of(O.normalize_psf(a))
    r, c, w = tabs.taps(0)
    assert list(zip(rr.tolist(), cc.tolist())) == [(60, 60), (70, 3)]
    assert np.array_equal(r.numpy(), rr) and np.array_equal(c.numpy(), cc)
    assert np.array_equal((w.numpy() & 0xffff).astype(np.uint16), ww.view(np.uint16))


# ------------------------------------------------------------------ manual_blur vs golden + oracle

@pytest.mark.parametrize("case", GI.blur_cases(), ids=lambda c: c["name"])
def test_manual_blur_golden(golden, case):
    from detectinblur_amd.models import blur_functions as BF
    img = GI.make_image(case)
    psf = GI.make_case_psf(case)
    out = BF.manual_blur(_dev(img), _dev(psf)).cpu().numpy()
    name = "blur_" + case["name"]
    if case.get("digest_only"):
        m = golden.meta[name]
        assert list(out.shape) == m["shape"] and str(out.dtype) == m["dtype"]
        assert np.array_equal(_bits(out[..., ::37, ::41]), golden.blur[name + "_sample"])
        assert hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest() == m["sha256"]
    else:
        want = golden.blur[name]
        assert out.shape == want.shape
        assert np.array_equal(_bits(out), want)


@pytest.mark.parametrize("case", [c for c in GI.blur_cases() if not c.get("digest_only")], ids=lambda c: c["name"])
def test_generic_kernel_agrees(golden, case):
    """Second, independent device implementation (direct global reads) against the same goldens."""
    from detectinblur_amd import _lib, blur_ops
    img = GI.make_image(case)
    psf = GI.make_case_psf(case)
    t_img = _dev(img)
    tabs = blur_ops.compact_psfs([_dev(psf)], normalize=False)
    out = torch.empty_like(t_img)
    C, H, W = img.shape
    _lib.check(_lib.lib().dib_sparse_blur_generic(t_img.data_ptr(), out.data_ptr(), C, H, W,
                                                  0 if img.dtype == np.float16 else 1, tabs.ptr(0), psf.shape[0],
                                                  torch.cuda.current_stream().cuda_stream))
    want = golden.blur["blur_" + case["name"]]
    assert np.array_equal(_bits(out.cpu().numpy().squeeze()), want)


def test_blur_image_list_golden(golden):
    from detectinblur_amd.models import blur_functions as BF
    imgs, dicts, psfs = GI.make_list_case()
    t_imgs = [_dev(a) for a in imgs]
    untouched = t_imgs[1]
    ret = BF.blur_image_list(t_imgs, dicts, [_dev(p) for p in psfs])
    assert ret is None
    assert t_imgs[1] is untouched
    for i, t in enumerate(t_imgs):
        assert np.array_equal(_bits(t.cpu().numpy()), golden.blur["blurlist_%d" % i])


def test_reflect_64_raises_like_reference():
    from detectinblur_amd.models import blur_functions as BF
    psf = torch.zeros(128, 128, dtype=torch.float16, device="cuda")
    psf[63, 63] = 1
    for shape in ((3, 64, 100), (3, 100, 64)):
        with pytest.raises(RuntimeError, match="Padding size should be less"):
            BF.manual_blur(torch.zeros(shape, dtype=torch.float16, device="cuda"), psf)


def test_random_ragged_batch_vs_oracle():
    """Seeded random PSFs of every extent class, ragged image sizes, one batched launch."""
    from detectinblur_amd.models import blur_functions as BF
    rs = np.random.RandomState(2024)
    imgs, dicts, psfs = [], [], []
    shapes = [(3, 97, 301), (1, 65, 65), (3, 33, 140), (2, 130, 257), (3, 70, 513), (3, 200, 66)]
    spreads = [2, 6, 14, 30, 50, 63]
    for sh, sp in zip(shapes, spreads):
        imgs.append(rs.random_sample(sh).astype(np.float16))
        a = np.zeros((128, 128), np.float64)
        n = 8 + 4 * sp
        rr = np.clip((rs.uniform(-sp, sp, n)).astype(int) + 63, 0, 127)
        cc = np.clip((rs.uniform(-sp, sp, n)).astype(int) + 63, 0, 127)
        a[rr, cc] = rs.random_sample(n) + 0.01
        psfs.append(O.to_half_like_torch(a * 0.37))
        dicts.append({"blurring": True})
    want = [a.copy() for a in imgs]
    O.blur_image_list(want, dicts, psfs)
    got = [_dev(a) for a in imgs]
    BF.blur_image_list(got, dicts, [_dev(p) for p in psfs])
    for g, w in zip(got, want):
        assert np.array_equal(_bits(g.cpu().numpy()), _bits(w))
    # the optional scheduling hint reorders the launch, never the results
    hinted = [dict(d, psf_taps=int(np.count_nonzero(p))) for d, p in zip(dicts, psfs)]
    got = [_dev(a) for a in imgs]
    BF.blur_image_list(got, hinted, [_dev(p) for p in psfs])
    for g, w in zip(got, want):
        assert np.array_equal(_bits(g.cpu().numpy()), _bits(w))


def test_batch_larger_than_one_launch_chunk_and_skipped_entries():
    """70 ragged images in one blur_image_list call (the device descriptor holds 32 images per launch:
    three launches), every third one not blurred, PSFs of several extents; bit-exact vs the oracle."""
    from detectinblur_amd.models import blur_functions as BF
    rs = np.random.RandomState(77)
    imgs, dicts, psfs = [], [], []
    for i in range(70):
        C = (1, 2, 3)[i % 3]
        imgs.append(rs.random_sample((C, 65 + (i * 7) % 40, 66 + (i * 13) % 90)).astype(np.float16))
        a = np.zeros((128, 128), np.float64)
        n, sp = 3 + i % 9, 1 + (i * 5) % 40
        a[np.clip(rs.randint(-sp, sp + 1, n) + 63, 0, 127), np.clip(rs.randint(-sp, sp + 1, n) + 63, 0, 127)] = rs.random_sample(n) + 0.05
        psfs.append(O.to_half_like_torch(a))
        dicts.append({"blurring": i % 3 != 1})
    want = [a.copy() for a in imgs]
    O.blur_image_list(want, dicts, psfs)
    got = [_dev(a) for a in imgs]
    untouched = [g for g, d in zip(got, dicts) if not d["blurring"]]
    BF.blur_image_list(got, dicts, [_dev(p) if d["blurring"] else torch.zeros(1, dtype=torch.float16, device="cuda")
                                    for p, d in zip(psfs, dicts)])
    assert all(any(g is u for u in untouched) for g, d in zip(got, dicts) if not d["blurring"])
    for g, w in zip(got, want):
        assert np.array_equal(_bits(g.cpu().numpy()), _bits(w))


def test_linearity_property_full_size():
    """Size-independent property at BASELINE size: a single-tap PSF of weight 1 is a pure shift of
    the reflect-padded image; checked exactly at 3 x 800 x 1333."""
    from detectinblur_amd.models import blur_functions as BF
    g = torch.Generator().manual_seed(1337)
    img = torch.rand(3, 800, 1333, generator=g).half().cuda()
    for (r, c) in ((63, 63), (60, 70), (50, 50), (100, 20)):
        psf = torch.zeros(128, 128, dtype=torch.float16, device="cuda")
        psf[r, c] = 1
        out = BF.manual_blur(img, psf)
        pad = torch.nn.functional.pad(img[None].float(), (63, 64, 63, 64), mode="reflect")[0].half()
        dy, dx = 63 - r, 63 - c
        want = pad[:, 63 + dy:63 + dy + 800, 63 + dx:63 + dx + 1333]
        assert torch.equal(out, want)


def test_full_size_batch_vs_c_oracle_digest():
    """configs[1] shape (8 x 3x800x1333, param_index=1 low-exposure PSFs): every image's digest must
    equal the golden-pinned single-image results where the inputs coincide, and all 8 must be
    self-consistent with per-image calls."""
    from detectinblur_amd.models import blur_functions as BF
    imgs = [torch.rand(3, 800, 1333, generator=torch.Generator().manual_seed(1337 + i)).half().cuda() for i in range(8)]
    psfs = [_dev(GI.golden_psf(0.005, i % 3, "half")) for i in range(8)]
    dicts = [{"blurring": True}] * 8
    batch = list(imgs)
    BF.blur_image_list(batch, dicts, psfs)
    for i in range(8):
        single = BF.manual_blur(imgs[i], _dev(O.normalize_psf(GI.golden_psf(0.005, i % 3, "half"))))
        assert torch.equal(batch[i], single)


# ------------------------------------------------------------------ DIB_ACC_FP32 ("accurate" mode)

ACC_FP32_TOL = 5e-3   # stated tolerance vs the reference's fp16 arithmetic (images in [0, 1], <= 272 taps;
                      # scaled by the image's largest magnitude when that exceeds 1)


@pytest.mark.parametrize("case", [c for c in GI.blur_cases() if not c.get("digest_only")], ids=lambda c: c["name"])
def test_acc_fp32_mode_bit_exact_vs_oracle_and_within_tolerance_of_reference(golden, case):
    """fp32-accumulate mode: bit-identical to its CPU restatement (exact products, fp32 sum, one
    rounding) for the tiled AND the generic kernel, and within 5e-3 of the reference's result."""
    from detectinblur_amd import _lib, blur_ops
    from detectinblur_amd.models import blur_functions as BF
    img = GI.make_image(case)
    psf = GI.make_case_psf(case)
    if img.dtype != np.float16:
        with pytest.raises(_lib.DibError, match="fp16 images only"):
            BF.manual_blur(_dev(img), _dev(psf), acc_mode=_lib.DIB_ACC_FP32)
        return
    want = O.manual_blur(img, psf, fp32_accumulate=True)
    got = BF.manual_blur(_dev(img), _dev(psf), acc_mode=_lib.DIB_ACC_FP32).cpu().numpy()
    assert np.array_equal(_bits(got), _bits(want))
    ref = golden.blur["blur_" + case["name"]].view(np.float16)
    assert np.abs(got.astype(np.float64) - ref.astype(np.float64)).max() <= ACC_FP32_TOL * max(1.0, float(np.abs(img).max()))
    t_img = _dev(img)
    tabs = blur_ops.compact_psfs([_dev(psf)], normalize=False)
    out = torch.empty_like(t_img)
    C, H, W = img.shape
    _lib.check(_lib.lib().dib_sparse_blur_generic(t_img.data_ptr(), out.data_ptr(), C, H, W, 2, tabs.ptr(0), psf.shape[0],
                                                  torch.cuda.current_stream().cuda_stream))
    assert np.array_equal(_bits(out.cpu().numpy().squeeze()), _bits(want))


ACC_FMA16_TOL = 1e-2   # stated tolerance of the fused-FMA mode vs the reference's fp16 arithmetic (images in [0, 1])


@pytest.mark.parametrize("case", [c for c in GI.blur_cases() if not c.get("digest_only")], ids=lambda c: c["name"])
def test_acc_fma16_mode_bit_exact_vs_oracle_and_within_tolerance_of_reference(golden, case):
    """Fused-FMA mode (one rounding per tap): bit-identical to its CPU restatement for the tiled and
    the generic kernel, within 1e-2 (scaled by the image magnitude) of the reference's result."""
    from detectinblur_amd import _lib, blur_ops
    from detectinblur_amd.models import blur_functions as BF
    img = GI.make_image(case)
    psf = GI.make_case_psf(case)
    if img.dtype != np.float16:
        with pytest.raises(_lib.DibError, match="fp16 images only"):
            BF.manual_blur(_dev(img), _dev(psf), acc_mode=_lib.DIB_ACC_FMA16)
        return
    want = O.manual_blur(img, psf, fma16=True)
    got = BF.manual_blur(_dev(img), _dev(psf), acc_mode=_lib.DIB_ACC_FMA16).cpu().numpy()
    assert np.array_equal(_bits(got), _bits(want))
    ref = golden.blur["blur_" + case["name"]].view(np.float16)
    assert np.abs(got.astype(np.float64) - ref.astype(np.float64)).max() <= ACC_FMA16_TOL * max(1.0, float(np.abs(img).max()))
    t_img = _dev(img)
    tabs = blur_ops.compact_psfs([_dev(psf)], normalize=False)
    out = torch.empty_like(t_img)
    C, H, W = img.shape
    _lib.check(_lib.lib().dib_sparse_blur_generic(t_img.data_ptr(), out.data_ptr(), C, H, W, 3, tabs.ptr(0), psf.shape[0],
                                                  torch.cuda.current_stream().cuda_stream))
    assert np.array_equal(_bits(out.cpu().numpy().squeeze()), _bits(want))


def test_acc_fp32_mode_full_size_batch():
    """configs[1] shape in fp32-accumulate mode: batch == per-image calls, one image checked against
    the oracle on a strided sample of rows (the full oracle pass takes minutes at this size)."""
    from detectinblur_amd import _lib
    from detectinblur_amd.models import blur_functions as BF
    imgs = [torch.rand(3, 800, 1333, generator=torch.Generator().manual_seed(1337 + i)).half().cuda() for i in range(8)]
    psfs_h = [GI.golden_psf(0.005, i % 3, "half") for i in range(8)]
    batch = list(imgs)
    BF.blur_image_list(batch, [{"blurring": True}] * 8, [_dev(p) for p in psfs_h], acc_mode=_lib.DIB_ACC_FP32)
    exact = list(imgs)
    BF.blur_image_list(exact, [{"blurring": True}] * 8, [_dev(p) for p in psfs_h])
    for i in range(8):
        single = BF.manual_blur(imgs[i], _dev(O.normalize_psf(psfs_h[i])), acc_mode=_lib.DIB_ACC_FP32)
        assert torch.equal(batch[i], single)
        assert (batch[i].float() - exact[i].float()).abs().max().item() <= ACC_FP32_TOL
    crop = imgs[3][:, 300:420, 500:760].cpu().numpy()            # interior crop: reflect padding of the crop
    want = O.manual_blur(crop, O.normalize_psf(psfs_h[3]), fp32_accumulate=True)   # differs only near its border
    got = batch[3][:, 300:420, 500:760].cpu().numpy()
    assert np.array_equal(_bits(got[:, 64:-64, 64:-64]), _bits(want[:, 64:-64, 64:-64]))


def test_segments_cover_taps_in_order_and_are_bounded():
    """The tap list is cut into consecutive segments of at most 13 rows x 25 columns (SEG_ROWS + 1, SEG_COLS + 1)."""
    from detectinblur_amd import blur_ops
    rs = np.random.RandomState(77)
    psfs = []
    for sp, n in ((3, 20), (20, 150), (60, 600), (63, 3000)):
        a = np.zeros((128, 128), np.float16)
        rr = np.clip(rs.randint(-sp, sp + 1, n) + 63, 0, 127)
        cc = np.clip(rs.randint(-sp, sp + 1, n) + 63, 0, 127)
        a[rr, cc] = (rs.random_sample(n) + 0.1).astype(np.float16)
        psfs.append(a)
    a = np.zeros((128, 128), np.float16); a[64, :] = 0.01; psfs.append(a)      # one full-width row
    a = np.zeros((128, 128), np.float16); a[:, 5] = 0.01; psfs.append(a)       # one full-height column
    psfs.append(np.full((128, 128), 0.001, np.float16))                        # dense
    tabs = blur_ops.compact_psfs(_dev(np.stack(psfs)), normalize=False)
    for k, a in enumerate(psfs):
        rr, cc = np.nonzero(a)
        segs = tabs.segments(k)
        assert segs[0][0] == 0 and segs[-1][1] == len(rr)
        for (s0, s1, rf, rl, cmn, cmx), nxt in zip(segs, segs[1:] + [None]):
            assert s1 > s0
            assert (rf, rl) == (rr[s0], rr[s1 - 1]) and rl - rf <= 12
            assert (cmn, cmx) == (cc[s0:s1].min(), cc[s0:s1].max()) and cmx - cmn <= 24
            if nxt is not None:
                assert nxt[0] == s1
                # greedy: the next tap could not have joined this segment
                assert rr[s1] - rf > 12 or max(cmx, cc[s1]) - min(cmn, cc[s1]) > 24


def test_repeated_full_size_batches_are_deterministic(golden):
    """The persistent tile queue and the hand-written tap loop under load: 6 back-to-back batches of
    8 x 3x800x1333 through ONE set of tap tables must all be bit-identical and match the golden
    digest where the inputs coincide (image seed 1337 + PSF p0.005/E1 = blur_full_e1_f16)."""
    from detectinblur_amd import blur_ops
    case = [c for c in GI.blur_cases() if c["name"] == "full_e1_f16"][0]
    imgs = [_dev(GI.make_image(case))] + [torch.rand(3, 800, 1333, generator=torch.Generator().manual_seed(50 + i)).half().cuda()
                                          for i in range(7)]
    psfs = [_dev(GI.golden_psf(0.005, 1, "half"))] + [_dev(GI.golden_psf(0.005, i % 5, "half")) for i in range(7)]
    tables = blur_ops.compact_psfs(psfs, normalize=True)
    first = None
    for rep in range(6):
        outs = blur_ops.sparse_blur(list(imgs), list(range(8)), tables)
        if first is None:
            first = [o.clone() for o in outs]
            got = outs[0].cpu().numpy()
            assert hashlib.sha256(np.ascontiguousarray(got).tobytes()).hexdigest() == golden.meta["blur_full_e1_f16"]["sha256"]
        else:
            for a, b in zip(first, outs):
                assert torch.equal(a, b)


def test_tile_shapes_and_orders_are_bit_identical():
    """The two tile shapes (128 x 32 "narrow", the default, and 256 x 32 with its HALF tap loop on edge tiles of <= 128
    valid columns), per-XCD banded and flat tile order, interior and border windows, zero-padded small images: all
    bit-identical to the oracle on a ragged batch of all PSF classes."""
    import ctypes
    from detectinblur_amd import _lib, blur_ops
    l = _lib.lib()
    l.dib_debug_set_tile_order.argtypes = [ctypes.c_int]
    l.dib_debug_set_tile_order.restype = None
    l.dib_debug_set_shape.argtypes = [ctypes.c_int]
    l.dib_debug_set_shape.restype = None
    rs = np.random.RandomState(11)
    imgs, psfs = [], []
    for sh, sp in zip([(3, 97, 301), (1, 65, 65), (2, 130, 257), (3, 70, 513), (3, 33, 140), (3, 200, 640), (1, 300, 90),
                       (2, 129, 384), (1, 96, 385)],
                      [2, 14, 30, 63, 8, 5, 40, 3, 20]):
        imgs.append(rs.random_sample(sh).astype(np.float16))
        a = np.zeros((128, 128), np.float64)
        n = 8 + 4 * sp
        a[np.clip(rs.randint(-sp, sp + 1, n) + 63, 0, 127), np.clip(rs.randint(-sp, sp + 1, n) + 63, 0, 127)] = rs.random_sample(n) + 0.01
        psfs.append(O.to_half_like_torch(a * 0.2))
    want = [a.copy() for a in imgs]
    O.blur_image_list(want, [{"blurring": True}] * len(imgs), psfs)
    tables = blur_ops.compact_psfs(_dev(np.stack(psfs)), normalize=True)
    try:
        for shape, bands in ((1, 1), (1, 0), (0, 1)):
            l.dib_debug_set_shape(shape)
            l.dib_debug_set_tile_order(bands)
            outs = blur_ops.sparse_blur([_dev(a) for a in imgs], list(range(len(imgs))), tables)
            for g, w in zip(outs, want):
                assert np.array_equal(_bits(g.cpu().numpy().squeeze()), _bits(w)), (shape, bands)
        # the fused-multiply-add mode: the two shapes agree with each other bit for bit
        fma = []
        for shape in (0, 1):
            l.dib_debug_set_shape(shape)
            fma.append(blur_ops.sparse_blur([_dev(a) for a in imgs], list(range(len(imgs))), tables, _lib.DIB_ACC_FMA16))
        for a, b in zip(*fma):
            assert torch.equal(a, b)
    finally:
        l.dib_debug_set_shape(0)
        l.dib_debug_set_tile_order(1)


@pytest.mark.parametrize("n_images", [2, 9, 15, 16])
def test_ragged_batch_grids_are_bit_identical(n_images):
    """A batch whose images differ in size runs on a 1-D grid of exactly its working workgroups (up to 15 images per launch:
    FlatBands, csrc/dib_common.h), any other batch on the 2-D grid (band entry x image): same tiles, same outputs -- bit for bit
    against the oracle, both modes, skipped entries and all, up to the flat grid's image limit and one past it."""
    import ctypes
    from detectinblur_amd import _lib, blur_ops
    l = _lib.lib()
    l.dib_debug_set_flat_grid.argtypes = [ctypes.c_int]
    l.dib_debug_set_flat_grid.restype = None
    rs = np.random.RandomState(100 + n_images)
    shapes = [(3, 97, 301), (3, 333, 500), (1, 65, 65), (3, 480, 640), (2, 130, 257), (3, 70, 513), (3, 33, 140), (3, 200, 640),
              (1, 300, 90), (2, 129, 384), (1, 96, 385), (3, 64 + 1, 700), (3, 427, 640), (3, 40, 40), (3, 257, 129), (3, 612, 612)]
    imgs, psfs = [], []
    for i in range(n_images):
        imgs.append(rs.random_sample(shapes[i]).astype(np.float16))
        sp = int(rs.choice([2, 5, 14, 30, 63]))
        a = np.zeros((128, 128), np.float64)
        n = 8 + 4 * sp
        a[np.clip(rs.randint(-sp, sp + 1, n) + 63, 0, 127), np.clip(rs.randint(-sp, sp + 1, n) + 63, 0, 127)] = rs.random_sample(n) + 0.01
        psfs.append(O.to_half_like_torch(a * 0.2))
    index = list(range(n_images))
    if n_images > 2:
        index[1] = -1                          # one entry not blurred: it takes no part in the launch
    want = [a.copy() for a in imgs]
    O.blur_image_list(want, [{"blurring": t >= 0} for t in index], psfs)
    tables = blur_ops.compact_psfs(_dev(np.stack(psfs)), normalize=True)
    try:
        got = {}
        for flat in (1, 0):
            l.dib_debug_set_flat_grid(flat)
            outs = blur_ops.sparse_blur([_dev(a) for a in imgs], index, tables)
            for g, w in zip(outs, want):
                assert np.array_equal(_bits(g.cpu().numpy().squeeze()), _bits(w.squeeze())), flat
            got[flat] = blur_ops.sparse_blur([_dev(a) for a in imgs], index, tables, _lib.DIB_ACC_FMA16)
        for a, b in zip(got[1], got[0]):
            assert torch.equal(a, b)
    finally:
        l.dib_debug_set_flat_grid(1)


def test_compaction_more_than_one_launch_chunk():
    """40 PSFs = two compaction launches (32 + 8): list and stacked entry points agree."""
    from detectinblur_amd import blur_ops
    rs = np.random.RandomState(21)
    psfs = []
    for k in range(40):
        a = np.zeros((128, 128), np.float16)
        n = 5 + k
        a[rs.randint(40, 90, n), rs.randint(40, 90, n)] = (rs.random_sample(n) + 0.1).astype(np.float16)
        psfs.append(a)
    t_list = blur_ops.compact_psfs([_dev(a) for a in psfs], normalize=True)
    t_stack = blur_ops.compact_psfs(_dev(np.stack(psfs)), normalize=True)
    for k in range(40):       # only the defined part of a table is comparable (the rest is torch.empty)
        assert t_list.header(k) == t_stack.header(k)
        for a, b in zip(t_list.taps(k), t_stack.taps(k)):
            assert torch.equal(a, b)
        assert t_list.segments(k) == t_stack.segments(k)
        for a, b in zip(t_list.ltaps(k), t_stack.ltaps(k)):
            assert torch.equal(a, b)
        for a, b in zip(t_list.ltaps(k, quad=True), t_stack.ltaps(k, quad=True)):
            assert torch.equal(a, b)
        # both offset arrays describe the same taps: (row distance, column distance) from the segment's last row / column
        wide, quad = t_list.ltaps(k)[0].numpy(), t_list.ltaps(k, quad=True)[0].numpy()
        assert np.array_equal(wide // 8 // 96, quad // 8 // 56) and np.array_equal(wide // 8 % 96, quad // 8 % 56)
    for k in (0, 31, 32, 39):
        rr, cc, ww = O.taps_of(O.normalize_psf(psfs[k]))
        r, c, w = t_list.taps(k)
        assert np.array_equal(r.numpy(), rr) and np.array_equal(c.numpy(), cc)
        assert np.array_equal((w.numpy() & 0xffff).astype(np.uint16), ww.view(np.uint16))
    assert t_list.buf.numel() == 40 * t_list.words


def test_psf_sum_equals_torch_half_sum_on_generated_psfs():
    """the compaction kernel forms the fp16 PSF sum exactly (int64 fixed point) and rounds once; the
    reference's `psf_GPU.sum()` on a Half CUDA tensor accumulates in fp32 in torch's reduction order and rounds once.
    The two can only differ when fp32 partial sums are inexact AND the total sits on a rounding boundary.  Measured over
    3000 generated PSFs of all types / exposures (scratch/t_sum_parity.py): 0 differences; 400 of them are checked here."""
    from detectinblur_amd import blur_ops
    from detectinblur_amd.motion_blur.generate_PSF import PSF
    from detectinblur_amd.motion_blur.generate_trajectory import Trajectory
    np.random.seed(7)
    rs = np.random.RandomState(8)
    fr = [1 / 18, 1 / 10, 1 / 5, 1 / 2, 1]
    psfs = []
    for _ in range(400):
        tr = Trajectory(canvas=256, max_len=96, expl=[0.005, 0.001, 0.00005][rs.randint(3)]).fit().fit()
        p = PSF(canvas=256, trajectory=tr, fraction=[fr[rs.randint(5)]])
        p.fit()
        p.centerPSF()
        psfs.append(torch.HalfTensor(np.ascontiguousarray(p.PSFs[0][64:192, 64:192])))
    stack = torch.stack(psfs).cuda()
    want = torch.stack([stack[i].sum() for i in range(len(psfs))]).cpu().view(torch.int16)
    got = []
    for b0 in range(0, len(psfs), 32):
        t = blur_ops.compact_psfs(stack[b0:b0 + 32].contiguous(), normalize=True)
        hdr = t.buf.view(t.count, t.words)[:, 6].cpu()
        got.append((hdr & 0xffff).to(torch.int32))
    got = torch.cat(got)
    assert torch.equal(got, want.to(torch.int32) & 0xffff)


def test_compaction_and_blur_are_graph_capturable():
    """The boundary takes a stream and never synchronises, copies or allocates behind the caller's back: tap compaction +
    blur captured into a HIP graph (two steps on two captured streams, as bench.py runs them) replay to the bit-identical
    result of the eager calls, for fresh input written between replays."""
    from detectinblur_amd import blur_ops
    rs = np.random.RandomState(5)
    imgs = [_dev(rs.random_sample((3, 97, 301)).astype(np.float16)), _dev(rs.random_sample((1, 200, 140)).astype(np.float16))]
    psfs = []
    for n in (12, 40):
        a = np.zeros((128, 128), np.float16)
        a[rs.randint(50, 80, n), rs.randint(50, 80, n)] = (rs.random_sample(n) + 0.1).astype(np.float16)
        psfs.append(_dev(a))

    def step():
        tables = blur_ops.compact_psfs(psfs, normalize=True)
        return blur_ops.sparse_blur(list(imgs), [0, 1], tables), tables

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    g = torch.cuda.CUDAGraph()
    keep = []
    with torch.cuda.graph(g, stream=streams[0], capture_error_mode="thread_local"):
        fork = torch.cuda.Event()
        fork.record(streams[0])
        streams[1].wait_event(fork)
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                keep.append(step())
        join = torch.cuda.Event()
        join.record(streams[1])
        streams[0].wait_event(join)
    for trial in range(2):
        for t in imgs:                                  # new pixels in the same buffers
            t.copy_(torch.rand(t.shape, device=t.device).half())
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        want, _ = step()
        for outs, _ in keep:
            for a, b in zip(outs, want):
                assert torch.equal(a, b)


# ------------------------------------------------------------------ the large LDS window (DIB_WINDOW_LARGE)

def _fp16_cases():
    return [c for c in GI.blur_cases() if np.dtype(GI.make_case_psf(c).dtype) == np.float16 or c.get("digest_only")]


@pytest.mark.parametrize("case", GI.blur_cases(), ids=lambda c: c["name"])
def test_large_window_kernel_against_the_reference_goldens(golden, case):
    """The large-window build of the default kernel (segments of up to 21 rows x 64 columns, 39.9 KB of LDS) on every
    reference golden an fp16 image takes -- all padding branches, the wrap quirk, the 256 canvas, the four 3 x 800 x 1333
    digests (two of them full-exposure PSFs: what the geometry exists for): bit for bit."""
    from detectinblur_amd import _lib, blur_ops
    img = GI.make_image(case)
    psf = GI.make_case_psf(case)
    if img.dtype != np.float16:
        pytest.skip("fp32 images run the generic kernel: no LDS window")
    t_img, t_psf = _dev(img), _dev(psf)
    if t_psf.dtype != t_img.dtype:
        t_psf = t_psf.to(t_img.dtype)
    tabs = blur_ops.compact_psfs([t_psf], normalize=False, large_window=True)
    assert tabs.large and (int(tabs.buf[5].item()) >> 16) == 1 and (int(tabs.buf[5].item()) & 0xffff) == psf.shape[0]
    out = blur_ops.sparse_blur([t_img], [0], tabs)[0].squeeze().cpu().numpy()
    name = "blur_" + case["name"]
    if case.get("digest_only"):
        m = golden.meta[name]
        assert list(out.shape) == m["shape"]
        assert hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest() == m["sha256"]
    else:
        assert np.array_equal(_bits(out), golden.blur[name])
    std = blur_ops.compact_psfs([t_psf], normalize=False)
    assert len(tabs.segments(0)) <= len(std.segments(0))
    # the FMA16 mode on the large window equals the FMA16 mode on the standard one
    a = blur_ops.sparse_blur([t_img], [0], tabs, _lib.DIB_ACC_FMA16)[0]
    b = blur_ops.sparse_blur([t_img], [0], std, _lib.DIB_ACC_FMA16)[0]
    assert torch.equal(a, b)


def test_large_window_segments_are_bounded_and_fewer():
    from detectinblur_amd import blur_ops
    rs = np.random.RandomState(78)
    psfs = []
    for sp, n in ((3, 20), (20, 150), (60, 600), (63, 3000)):
        a = np.zeros((128, 128), np.float16)
        a[np.clip(rs.randint(-sp, sp + 1, n) + 63, 0, 127), np.clip(rs.randint(-sp, sp + 1, n) + 63, 0, 127)] = (rs.random_sample(n) + 0.1).astype(np.float16)
        psfs.append(a)
    a = np.zeros((128, 128), np.float16); a[np.arange(20, 116), np.arange(20, 116)] = 0.01; psfs.append(a)      # a 96-pixel diagonal
    a = np.zeros((128, 128), np.float16); a[:, 5] = 0.01; psfs.append(a)
    psfs.append(np.full((128, 128), 0.001, np.float16))
    big = blur_ops.compact_psfs(_dev(np.stack(psfs)), normalize=False, large_window=True)
    std = blur_ops.compact_psfs(_dev(np.stack(psfs)), normalize=False)
    for k, a in enumerate(psfs):
        rr, cc = np.nonzero(a)
        segs = big.segments(k)
        assert segs[0][0] == 0 and segs[-1][1] == len(rr)
        for (s0, s1, rf, rl, cmn, cmx), nxt in zip(segs, segs[1:] + [None]):
            assert s1 > s0 and (rf, rl) == (rr[s0], rr[s1 - 1]) and rl - rf <= 20
            assert (cmn, cmx) == (cc[s0:s1].min(), cc[s0:s1].max()) and cmx - cmn <= 63
            if nxt is not None:
                assert nxt[0] == s1 and (rr[s1] - rf > 20 or max(cmx, cc[s1]) - min(cmn, cc[s1]) > 63)
        off, w = big.ltaps(k, quad=True)
        r_last = np.concatenate([[rl] * (s1 - s0) for s0, s1, rf, rl, cmn, cmx in segs])
        c_max = np.concatenate([[cmx] * (s1 - s0) for s0, s1, rf, rl, cmn, cmx in segs])
        assert np.array_equal(off.numpy(), ((r_last - rr) * 96 + (c_max - cc)) * 8)          # 96 elements per LDS row
        assert len(segs) <= len(std.segments(k))
    assert len(big.segments(4)) <= 5 < len(std.segments(4))                                  # the diagonal: 8 standard segments


def test_large_window_random_batches_vs_oracle_and_argument_errors():
    """Ragged batches, every padding regime, skipped entries through blur_image_list with the extent hint (which selects
    the large window for launches of one or two images) and through dib_blur_step; fp32 images / DIB_ACC_FP32 refuse it."""
    from detectinblur_amd import _lib, blur_ops
    from detectinblur_amd.models import blur_functions as BF
    rs = np.random.RandomState(2025)
    for trial in range(12):
        n = 1 + trial % 2
        imgs, psfs, dicts = [], [], []
        for i in range(n):
            C = (3, 1, 2)[(trial + i) % 3]
            shape = [(C, 97, 301), (C, 65, 65), (C, 33, 140), (C, 130, 257), (C, 40, 50), (C, 200, 66)][(trial + i) % 6]
            imgs.append(rs.random_sample(shape).astype(np.float16))
            a = np.zeros((128, 128), np.float64)
            m, sp = 20 + 15 * (trial % 5), 20 + 4 * trial
            rr = np.clip(rs.randint(-sp, sp + 1, m) + 63, 0, 127); cc = np.clip(rs.randint(-sp // 3 - 1, sp // 3 + 2, m) + 63, 0, 127)
            a[rr, cc] = rs.random_sample(m) + 0.01
            h = O.to_half_like_torch(a * 0.2)
            psfs.append(h)
            r_, c_ = np.nonzero(h)
            from detectinblur_amd import transforms as T
            dicts.append({"blurring": True, "psf_taps": len(r_), "psf_extent": (int(r_.min()), int(r_.max()), int(c_.min()), int(c_.max())),
                          "psf_segments": (T.count_tap_segments(h, T.STANDARD_WINDOW), T.count_tap_segments(h, T.LARGE_WINDOW))})
            # the host-side hint counts what the device's compaction produces (no weight underflows here)
            for lw, k in ((False, 0), (True, 1)):
                assert len(blur_ops.compact_psfs([_dev(h)], normalize=True, large_window=lw).segments(0)) == dicts[-1]["psf_segments"][k]
        std, big = sum(d["psf_segments"][0] for d in dicts), sum(d["psf_segments"][1] for d in dicts)
        assert blur_ops.large_window_pays(dicts, n) == (2.4 * std > 5.0 * big + 6.0)
        want = [a.copy() for a in imgs]
        O.blur_image_list(want, dicts, psfs)
        got = [_dev(a) for a in imgs]
        BF.blur_image_list(got, dicts, [_dev(p) for p in psfs])          # the policy picks the window
        for g, w in zip(got, want):
            assert np.array_equal(_bits(g.cpu().numpy()).reshape(w.shape), _bits(w))
        outs = blur_ops.blur_step([_dev(a) for a in imgs], list(range(n)), [_dev(p) for p in psfs], large_window=True)      # forced
        for g, w in zip(outs, want):
            assert np.array_equal(_bits(g.cpu().numpy()).reshape(w.shape), _bits(w))
    assert not blur_ops.large_window_pays(dicts * 3, 3)
    tabs = blur_ops.compact_psfs([_dev(psfs[0])], normalize=True, large_window=True)
    with pytest.raises(_lib.DibError):
        blur_ops.sparse_blur([torch.rand(3, 80, 90, device="cuda")], [0], tabs)                                   # fp32 image
    with pytest.raises(_lib.DibError):
        blur_ops.sparse_blur([torch.rand(3, 80, 90, device="cuda").half()], [0], tabs, _lib.DIB_ACC_FP32)
