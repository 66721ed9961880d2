"""Pins the CPU oracle (oracle/dib_oracle.py) against golden vectors produced by the REAL
reference (oracle/gen_goldens.py, run in the build container).  Bit-exact unless stated."""
import hashlib

import numpy as np
import pytest

import dib_oracle as O
import golden_inputs as GI


def _cx(re, im):
    return re + 1j * im


@pytest.mark.parametrize("param", GI.PARAMS)
@pytest.mark.parametrize("seed", GI.TRAJ_SEEDS)
def test_trajectory_bit_exact(golden, param, seed):
    np.random.seed(seed)
    r1 = O.trajectory(256, 2000, 96, param)
    r2 = O.trajectory(256, 2000, 96, param)
    key = "traj_p%g_s%d" % (param, seed)
    assert np.array_equal(_cx(r1[0], r1[1]), golden.traj[key + "_fit1"])
    assert np.array_equal(_cx(r2[0], r2[1]), golden.traj[key + "_fit2"])
    assert [r2[2], r2[3]] == list(golden.traj[key + "_len"])
    # RNG stream position (uniform + cached gaussian) must match too
    assert [np.random.uniform(), np.random.randn()] == list(golden.traj[key + "_next"])


def test_trajectory_expl_none_and_big_shakes(golden):
    np.random.seed(7)
    expl = 0.1 * np.random.uniform(0, 1)
    r = O.trajectory(64, 500, 60, expl)
    assert np.array_equal(_cx(r[0], r[1]), golden.traj["traj_none_s7"])
    assert [expl, r[2], r[3]] == list(golden.traj["traj_none_s7_expl"])
    np.random.seed(11)
    r = O.trajectory(256, 2000, 96, 0.9)
    assert r[3] >= 5          # the cexp branch really ran
    assert np.array_equal(_cx(r[0], r[1]), golden.traj["traj_big_s11"])
    assert [r[2], r[3]] == list(golden.traj["traj_big_s11_len"])


@pytest.mark.parametrize("param", GI.PARAMS)
@pytest.mark.parametrize("fi", range(len(GI.FRACTIONS)))
def test_psf_chain_bit_exact(golden, param, fi):
    np.random.seed(GI.psf_seed(param, fi))
    O.trajectory(256, 2000, 96, param)
    x_re, x_im, _, _ = O.trajectory(256, 2000, 96, param)
    raw = O.psf_rasterize(x_re, x_im, [GI.FRACTIONS[fi]], 256)[0]
    assert np.array_equal(raw, GI.golden_psf(param, fi, "raw"))
    cen = O.psf_center(raw)
    assert np.array_equal(cen, GI.golden_psf(param, fi, "cen"))
    crop = O.psf_crop128(cen)
    assert np.array_equal(crop, GI.golden_psf(param, fi, "crop"))
    half = O.to_half_like_torch(crop)
    assert np.array_equal(half.view(np.uint16), GI.golden_psf(param, fi, "half").view(np.uint16))
    key = "psf_p%g_f%d" % (param, fi)
    assert O.half_sum_exact(half).view(np.uint16) == golden.psf[key + "_sum"][0]
    norm = O.normalize_psf(half)
    r, c, w = O.taps_of(norm)
    assert np.array_equal(np.stack([r, c], 1), golden.psf[key + "_norm_rc"].astype(np.int64))
    assert np.array_equal(w.view(np.uint16), golden.psf[key + "_norm_w"])


def test_psf_multi_fraction(golden):
    x = golden.psf["psf_multi_traj"]
    out = O.psf_rasterize(x.real, x.imag, [1 / 100, 1 / 10, 1 / 2, 1], 128)
    for i, a in enumerate(out):
        assert np.array_equal(a, golden.psf["psf_multi_%d" % i])


def _bits(a):
    return a.view(np.uint16 if a.dtype == np.float16 else np.uint32)


@pytest.mark.parametrize("case", [c for c in GI.blur_cases() if not c.get("digest_only")],
                         ids=lambda c: c["name"])
def test_manual_blur_bit_exact(golden, case):
    out = O.manual_blur(GI.make_image(case), GI.make_case_psf(case))
    want = golden.blur["blur_" + case["name"]]
    assert out.shape == want.shape
    assert np.array_equal(_bits(out), want)


# DIB_ACC_FP32 ("accurate" mode): fp32 running sum, one rounding.  Stated
# tolerance against the reference's fp16 arithmetic: 5e-3 absolute for images in [0, 1] and <= 272 taps
# (scaled by the image's largest magnitude when that exceeds 1).
ACC_FP32_TOL = 5e-3


@pytest.mark.parametrize("case", [c for c in GI.blur_cases() if not c.get("digest_only") and c.get("dtype", "float16") == "float16"],
                         ids=lambda c: c["name"])
def test_fp32_accumulate_mode_within_stated_tolerance(golden, case):
    img = GI.make_image(case)
    if img.dtype != np.float16:
        pytest.skip("fp16 images only")
    out = O.manual_blur(img, GI.make_case_psf(case), fp32_accumulate=True)
    want = golden.blur["blur_" + case["name"]].view(np.float16)
    assert out.dtype == np.float16 and out.shape == want.shape
    assert np.abs(out.astype(np.float64) - want.astype(np.float64)).max() <= ACC_FP32_TOL * max(1.0, float(np.abs(img).max()))


ACC_FMA16_TOL = 1e-2


@pytest.mark.parametrize("case", [c for c in GI.blur_cases() if not c.get("digest_only") and c.get("dtype", "float16") == "float16"],
                         ids=lambda c: c["name"])
def test_fma16_mode_within_stated_tolerance(golden, case):
    img = GI.make_image(case)
    if img.dtype != np.float16:
        pytest.skip("fp16 images only")
    out = O.manual_blur(img, GI.make_case_psf(case), fma16=True)
    want = golden.blur["blur_" + case["name"]].view(np.float16)
    assert out.dtype == np.float16 and out.shape == want.shape
    assert np.abs(out.astype(np.float64) - want.astype(np.float64)).max() <= ACC_FMA16_TOL * max(1.0, float(np.abs(img).max()))


def test_manual_blur_coco_size_digest(golden):
    case = [c for c in GI.blur_cases() if c["name"] == "coco_e2_f16"][0]
    out = O.manual_blur(GI.make_image(case), GI.make_case_psf(case))
    m = golden.meta["blur_coco_e2_f16"]
    assert list(out.shape) == m["shape"]
    assert hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest() == m["sha256"]


def test_blur_image_list(golden):
    imgs, dicts, psfs = GI.make_list_case()
    before = imgs[1].copy()
    O.blur_image_list(imgs, dicts, psfs)
    for i, a in enumerate(imgs):
        assert np.array_equal(_bits(a), golden.blur["blurlist_%d" % i])
    assert np.array_equal(imgs[1], before)


def test_reflect_needs_more_than_64():
    img = np.zeros((3, 64, 100), dtype=np.float16)
    psf = np.zeros((128, 128), dtype=np.float16)
    psf[63, 63] = 1
    with pytest.raises(RuntimeError):
        O.manual_blur(img, psf)


@pytest.mark.parametrize("case", GI.box_cases(), ids=lambda c: c["name"])
def test_expand_boxes(golden, case):
    boxes, psf, shape = GI.make_box_case(case)
    out = O.expand_boxes(boxes, psf, shape[1], shape[2])
    assert np.array_equal(out.view(np.uint32), golden.boxes["boxes_" + case["name"]])


def test_squeeze_boxes(golden):
    out = O.clamp_boxes(GI.make_squeeze_boxes(), 100, 150)
    assert np.array_equal(out.view(np.uint32), golden.boxes["boxes_squeeze"])


def test_expand_refuses_256():
    with pytest.raises(Exception, match="not 128 wide"):
        O.psf_extents(np.ones((256, 256), dtype=np.float16))


def test_norm_params(golden):
    d = GI.norm_dicts()
    for flag in (False, True):
        m, s = O.norm_params(d, flag)
        assert np.array_equal(m, golden.norm["norm_means_%d" % flag])
        assert np.array_equal(s, golden.norm["norm_stds_%d" % flag])
    m, s = O.norm_params(None, True)
    assert np.array_equal(m, golden.norm["norm_none_means"])
    assert np.array_equal(s, golden.norm["norm_none_stds"])


def test_axis_stats_against_blurdicts(golden):
    recs = golden.meta["blurimage"]
    n = 0
    for mode, blob in recs.items():
        if mode == "preblurred":
            continue
        for call, rec in enumerate(blob["records"]):
            if not rec["blurring"]:
                continue
            k = "bd_%s_%d" % (mode, call)
            psf = np.zeros(rec["psf_shape"], dtype=rec["psf_dtype"])
            psf[golden.blurdict[k + "_r"].astype(int), golden.blurdict[k + "_c"].astype(int)] = golden.blurdict[k + "_v"]
            th, s1, s2 = O.psf_axis_stats(psf)
            assert th.hex() == rec["theta_rad"]
            assert s1.hex() == rec["scale_factor_lambda1"]
            assert s2.hex() == rec["scale_factor_lambda2"]
            n += 1
    assert n > 20


def test_cpu_fft_blur_close(golden):
    """A18 is pinned only approximately: cv2 is absent, the golden came from the reference run
    with a numpy min-max shim for cv2.normalize.  Tolerance: 1 grey level."""
    out = O.cpu_fft_blur(GI.make_fft_image(), GI.make_fft_psf())
    want = golden.fft["fft_out"]
    assert out.shape == want.shape
    assert np.abs(out.astype(int) - want.astype(int)).max() <= 1


def test_tap_segments_and_vertical_run_order_restatements():
    """oracle.tap_segments is the greedy cut the library's compaction makes (the product's host-side hint, transforms.count_tap_segments,
    restates the same rule independently: the counts must agree), and tap_order_vruns -- the order DIB_ACC_FAST16 accumulates in
    (tests/test_fast16_gpu.py pins the device against it) -- is a permutation that walks every PSF column of a segment downwards."""
    import dib_oracle as O
    from detectinblur_amd import transforms as TR
    rs = np.random.RandomState(3)
    psfs = []
    for n, spread in ((1, 0), (7, 2), (40, 6), (120, 20), (300, 50)):
        a = np.zeros((128, 128), np.float16)
        a[np.clip(rs.randint(-spread, spread + 1, n) + 63, 0, 127), np.clip(rs.randint(-spread, spread + 1, n) + 63, 0, 127)] = 0.5
        psfs.append(a)
    band = np.zeros((128, 128), np.float16)
    for k in range(30):
        band[50 + k, 60 + k // 3:63 + k // 3] = 0.25          # a thick slanted band: 30 rows, i.e. three segments
    psfs.append(band)
    for psf in psfs:
        rows, cols, _ = O.taps_of(psf)
        segs = O.tap_segments(rows, cols)
        assert len(segs) == TR.count_tap_segments(psf, (12, 24))
        assert segs[0][0] == 0 and segs[-1][1] == len(rows) and all(a[1] == b[0] for a, b in zip(segs, segs[1:]))
        for a, b in segs:
            assert rows[b - 1] - rows[a] <= 12 and cols[a:b].max() - cols[a:b].min() <= 24
        order = O.tap_order_vruns(rows, cols)
        assert sorted(order) == list(range(len(rows)))
        seg_of = np.zeros(len(rows), int)
        for k, (a, b) in enumerate(segs):
            seg_of[a:b] = k
        assert all(seg_of[order[i]] <= seg_of[order[i + 1]] for i in range(len(order) - 1))     # segment by segment
        # cut the order back into its groups: a group goes down one column, at most four taps
        groups = [[order[0]]]
        for p, q in zip(order, order[1:]):
            if seg_of[p] == seg_of[q] and cols[p] == cols[q] and rows[q] == rows[p] + 1 and len(groups[-1]) < 4:
                groups[-1].append(q)
            else:
                groups.append([q])
        for k, (a, b) in enumerate(segs):
            mine = [g for g in groups if seg_of[g[0]] == k]
            assert [len(g) for g in mine] == sorted((len(g) for g in mine), reverse=True)        # by size, fours first
            taps = {(rows[j], cols[j]) for j in range(a, b)}
            for g in mine:
                if len(g) < 4:          # a short group ends its run: the segment has no tap right below it
                    assert (rows[g[-1]] + 1, cols[g[-1]]) not in taps
    # the band: three segments, every tap in a run of two or more
    rows, cols, _ = O.taps_of(band)
    assert len(O.tap_segments(rows, cols)) == 3
