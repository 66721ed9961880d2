"""DIB_ACC_FAST16 -- the tolerance mode built for speed (include/dib.h): DIB_ACC_FMA16's arithmetic, one fused fp16 multiply-add per
pixel and tap, with every window's taps regrouped into vertical runs so that they share their LDS reads.  north_star asks for the
reference's result "within a stated fp32 tolerance": the tolerance is ACC_FAST16_TOL below, against the reference's own goldens
(models/blur_functions.py:11-69).  A tolerance alone would let a wrong tap with a small weight through, so the mode is ALSO pinned
bit for bit: against the oracle's restatement of its accumulation order (oracle/dib_oracle.py: tap_order_vruns) and against that
order read back from the device's own tables."""
import numpy as np
import pytest
import torch

import dib_oracle as O

pytestmark = pytest.mark.gpu

ACC_FAST16_TOL = 1e-2   # absolute, images in [0, 1]: the stated tolerance of the mode against the reference's fp16 arithmetic


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _bits(t):
    return t.cpu().numpy().view(np.uint16)


def _psf(rs, n, spread, K=128):
    a = np.zeros((K, K), np.float64)
    c = K // 2 - 1
    a[np.clip(rs.randint(-spread, spread + 1, n) + c, 0, K - 1), np.clip(rs.randint(-spread, spread + 1, n) + c, 0, K - 1)] = rs.random_sample(n) + 0.05
    return O.to_half_like_torch(a / a.sum())


def _band_psf(rs, length, thick):
    """A thick, slanted band like a rasterised trajectory: every tap has vertical neighbours."""
    a = np.zeros((128, 128), np.float64)
    x, y = 63.0 - length / 3, 63.0 - length / 2
    for k in range(length * 3):
        x += rs.uniform(0.0, 0.5); y += rs.uniform(0.1, 0.45)
        for dy in range(thick):
            for dx in range(2):
                a[int(np.clip(y + dy, 0, 127)), int(np.clip(x + dx, 0, 127))] += rs.random_sample() + 0.1
    return O.to_half_like_torch(a / a.sum())


def _generated_psfs():
    import bench
    return [O.normalize_psf(O.to_half_like_torch(p)) for p in bench.make_psfs_host(0)[0]]


def _cases():
    rs = np.random.RandomState(77)
    out = []
    for p in _generated_psfs()[:4]:
        out.append(("generated", p))
    out.append(("band", _band_psf(rs, 9, 3)))
    out.append(("tall band, several segments", _band_psf(rs, 40, 5)))
    out.append(("scattered: single taps", _psf(rs, 40, 9)))
    out.append(("wide scatter: many segments", _psf(rs, 160, 40)))
    one = np.zeros((128, 128), np.float16); one[63, 63] = 1.0
    out.append(("one tap", one))
    col = np.zeros((128, 128), np.float64); col[50:80, 64] = 1.0 / 30
    out.append(("one column: runs longer than a group and than a segment", O.to_half_like_torch(col)))
    corner = np.zeros((128, 128), np.float16); corner[0, 0] = 0.25; corner[1, 0] = 0.25; corner[127, 127] = 0.25; corner[126, 127] = 0.25
    out.append(("corners (wrap rows / columns)", corner))
    return out


@pytest.mark.parametrize("name,psf", _cases(), ids=[c[0] for c in _cases()])
def test_fast16_equals_its_restated_order_bit_for_bit(name, psf):
    from detectinblur_amd import _lib, blur_ops
    rs = np.random.RandomState(len(name))
    for shape in ((3, 97, 150), (1, 40, 200), (2, 200, 333), (3, 64 + 3, 129)):
        img = rs.random_sample(shape).astype(np.float16)
        tabs = blur_ops.compact_psfs([_dev(psf)], normalize=False, vruns=True)
        got = blur_ops.sparse_blur([_dev(img)], [0], tabs, _lib.DIB_ACC_FAST16)[0]
        rows, cols, _ = O.taps_of(psf)
        want = O.manual_blur(img, psf, fma16=True, tap_order=O.tap_order_vruns(rows, cols))
        assert np.array_equal(_bits(got).reshape(want.shape), want.view(np.uint16)), (name, shape)
        # the other tolerance mode and the bit-exact one are untouched by tables that carry the groups
        assert torch.equal(blur_ops.sparse_blur([_dev(img)], [0], tabs, _lib.DIB_ACC_FMA16)[0],
                           blur_ops.sparse_blur([_dev(img)], [0], blur_ops.compact_psfs([_dev(psf)], normalize=False), _lib.DIB_ACC_FMA16)[0])
        exact = blur_ops.sparse_blur([_dev(img)], [0], tabs)[0]
        assert np.array_equal(_bits(exact).reshape(want.shape), O.manual_blur(img, psf).view(np.uint16))
        assert float((exact.float() - got.float()).abs().max()) <= ACC_FAST16_TOL


def test_the_groups_in_the_table_are_the_taps_regrouped():
    """Every tap exactly once, every group a vertical run of at most four, a segment's groups sorted by size, the offsets those of the
    taps' own window positions."""
    from detectinblur_amd import blur_ops
    for name, psf in _cases():
        tabs = blur_ops.compact_psfs([_dev(psf)], normalize=False, vruns=True)
        rows, cols, wbits = (t.numpy() for t in tabs.taps(0))
        groups = tabs.vgroups(0)
        assert groups is not None, name
        order = O.tap_order_vruns(rows, cols)
        k = 0
        for (t0, t1, rf, rl, cmn, cmx), seg in zip(tabs.segments(0), groups):
            assert sum(len(w) for _, w in seg) == t1 - t0, name
            assert [len(w) for _, w in seg] == sorted((len(w) for _, w in seg), reverse=True), name      # by size, fours first
            for off, wts in seg:
                assert 1 <= len(wts) <= 4
                # tap j of the group reads window rows j .. j + 3 from `off`: w[0] is the group's HIGHEST PSF row; the loop runs
                # j = n - 1 .. 0, i.e. lowest row first -- the restated order
                for j in reversed(range(len(wts))):
                    t = order[k]; k += 1
                    assert (int(wbits[t]) & 0xffff) == wts[j], (name, k)
                    assert off + j * 56 * 8 == ((rl - int(rows[t])) * 56 + (cmx - int(cols[t]))) * 8, (name, k)
        assert k == len(rows)
    # tables compacted without the flag carry no groups, and the mode refuses them on the host
    from detectinblur_amd import _lib
    plain = blur_ops.compact_psfs([_dev(_cases()[0][1])], normalize=False)
    assert plain.vgroups(0) is None
    with pytest.raises(ValueError, match="vruns=True"):
        blur_ops.sparse_blur([_dev(np.zeros((3, 80, 80), np.float16))], [0], plain, _lib.DIB_ACC_FAST16)


def _golden_cases():
    import golden_inputs as GI
    return [c for c in GI.blur_cases() if not c.get("digest_only")]


@pytest.mark.parametrize("case", _golden_cases(), ids=lambda c: c["name"])
def test_fast16_within_the_stated_tolerance_of_the_reference_goldens(golden, case):
    """Against the outputs of the reference's own manual_blur (tests/golden/blur.npz, oracle/gen_goldens.py): every fp16 golden on
    the 128 canvas -- all padding branches, the wrap rows, single taps, dense and subnormal PSFs."""
    import golden_inputs as GI
    from detectinblur_amd import _lib
    from detectinblur_amd.models import blur_functions as BF
    img, psf = GI.make_image(case), GI.make_case_psf(case)
    if img.dtype != np.float16 or psf.shape[0] != 128:
        with pytest.raises((_lib.DibError, ValueError)):
            BF.manual_blur(_dev(img), _dev(psf), acc_mode=_lib.DIB_ACC_FAST16)
        return
    got = BF.manual_blur(_dev(img), _dev(psf), acc_mode=_lib.DIB_ACC_FAST16).cpu().numpy()
    ref = golden.blur["blur_" + case["name"]].view(np.float16)
    assert np.abs(got.astype(np.float64) - ref.astype(np.float64).reshape(got.shape)).max() <= ACC_FAST16_TOL * max(1.0, float(np.abs(img).max()))
    rows, cols, _ = O.taps_of(psf)
    # (a PSF of more than 4,096 taps -- the dense golden -- is beyond the compaction's LDS stage: its table carries no groups and
    # the mode runs the plain fused loop in row-major order)
    want = O.manual_blur(img, psf, fma16=True, tap_order=O.tap_order_vruns(rows, cols) if len(rows) <= 4096 else None)
    assert np.array_equal(_bits(torch.from_numpy(got)).reshape(want.shape), want.view(np.uint16))


def test_fast16_on_the_baseline_batch_and_a_ragged_one():
    """Whole batches through both grids (2-D for equal sizes, the 1-D grid of a ragged batch), blur_step and sparse_blur, at
    BASELINE's size: bit-identical to the restated order on a strided sample, within tolerance of the bit-exact result everywhere."""
    import bench
    from detectinblur_amd import _lib, blur_ops
    psfs = _generated_psfs()
    t_psfs = [_dev(O.to_half_like_torch(p)) for p in bench.make_psfs_host(0)[0]]       # un-normalised, as the drop-in receives them
    rs = np.random.RandomState(9)
    for sizes in ([(800, 1333)] * 8, bench.COCO_NATIVE_SIZES):
        imgs = [rs.random_sample((3, h, w)).astype(np.float16) for h, w in sizes]
        t_imgs = [_dev(a) for a in imgs]
        tabs = blur_ops.compact_psfs(t_psfs, normalize=True, vruns=True)
        fast = blur_ops.sparse_blur(t_imgs, list(range(8)), tabs, _lib.DIB_ACC_FAST16)
        step = blur_ops.blur_step(t_imgs, list(range(8)), t_psfs, acc_mode=_lib.DIB_ACC_FAST16)
        exact = blur_ops.sparse_blur(t_imgs, list(range(8)), tabs)
        for i in range(8):
            assert torch.equal(fast[i], step[i]), i
            assert float((fast[i].float() - exact[i].float()).abs().max()) <= ACC_FAST16_TOL, i
        for i in (0, 7):              # the oracle (seconds per full-size image)
            rows, cols, _ = O.taps_of(psfs[i])
            order = O.tap_order_vruns(rows, cols)
            want = O.manual_blur(imgs[i], psfs[i], fma16=True, tap_order=order)
            assert np.array_equal(_bits(fast[i]), want.view(np.uint16)), i
