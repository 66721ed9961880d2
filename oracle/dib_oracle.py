"""CPU oracle: a restatement of the reference's hot path in numpy.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import this package; the product package
`detectinblur_amd` never does (it raises if its HIP library is missing).

Every function restates, from the arithmetic spec in SURVEY.md section 8 / appendix A, one
function of mohammed-amr/detectInBlur and cites the reference file:line it follows.  It is
written in explicit real arithmetic (no complex objects, no torch.roll) so that it doubles as
the spec for the HIP kernels.

PARITY PINNING: `oracle/gen_goldens.py` imports the real reference (in the build container,
where /root/reference exists) and writes `tests/golden/*.npz`; `tests/test_oracle_golden.py`
checks this file against those vectors bit for bit (fp64 trajectories / PSFs, fp16 blur
results, fp32 boxes).  Pinned numeric environment of those runs: numpy 2.2.6, glibc 2.35,
torch 2.10 (CPU half arithmetic = fp32 compute + one rounding, which equals native fp16
arithmetic by the p_wide >= 2p+2 double-rounding theorem).

The detector (A12-A15) is NOT covered here: its arithmetic lives in an un-vendored, unpinned
torchvision -- parity unpinned for that part.
"""
import ctypes
import ctypes.util
import math

import numpy as np

# --------------------------------------------------------------------------------------
# A1  Trajectory.fit            (reference motion_blur/generate_trajectory.py:38-98)
# --------------------------------------------------------------------------------------

_libm = None


def _cexp(re, im):
    """np.exp(complex) == glibc cexp() (probe-verified; NOT equal to (cos, sin) from
    separate libm calls in ~0.16 % of arguments).  generate_trajectory.py:70."""
    z = np.exp(complex(re, im))
    return float(z.real), float(z.imag)


def _m():
    global _libm
    if _libm is None:
        _libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
        _libm.fma.restype = ctypes.c_double
        _libm.fma.argtypes = [ctypes.c_double] * 3
        _libm.hypot.restype = ctypes.c_double
        _libm.hypot.argtypes = [ctypes.c_double] * 2
    return _libm


def _fma(a, b, c):
    return _m().fma(a, b, c)


def cabs_numpy(re, im):
    """|re + i*im| exactly as numpy >= 1.25 computes np.abs(complex128) on an FMA machine:
    max * sqrt(fma(min/max, min/max, 1)) (probe-verified 200k/200k).  generate_trajectory.py:80."""
    a, b = abs(re), abs(im)
    mx, mn = (a, b) if a >= b else (b, a)
    if mx == 0.0:
        return 0.0
    d = mn / mx
    return mx * math.sqrt(_fma(d, d, 1.0))


def trajectory(canvas=64, iters=2000, max_len=60, expl=None, rng=np.random):
    """Returns (x_re, x_im, tot_length, big_expl_count); x = x_re + i*x_im is `Trajectory.x`.

    Follows generate_trajectory.py:8-36 (constructor draw for expl=None) and :38-98 (fit).
    RNG draw order on the legacy global stream: 4x uniform, then per step uniform,
    [uniform if big shake], randn (real), randn (imag).  One call = ONE fit; the reference's
    caller runs fit twice (transforms.py:316-317)."""
    if expl is None:
        expl = 0.1 * rng.uniform(0, 1)                                    # :29
    centripetal = 0.7 * rng.uniform(0, 1)                                 # :48
    prob_big_shake = 0.2 * rng.uniform(0, 1)                              # :50
    gaussian_shake = 10 * rng.uniform(0, 1)                               # :52
    init_angle = 360 * rng.uniform(0, 1)                                  # :53
    rad = float(init_angle) * (math.pi / 180.0)                          # np.deg2rad
    v_im0 = float(np.sin(rad))                                            # :55
    v_re0 = float(np.cos(rad))                                            # :56
    step = max_len / (iters - 1)
    # :59  v = v0 * max_len / (iters-1)  (python complex: scale then true division)
    v_re = (v_re0 * max_len) / (iters - 1)
    v_im = (v_im0 * max_len) / (iters - 1)
    if expl > 0:                                                          # :61-62
        v_re = v_re0 * expl
        v_im = v_im0 * expl
    x_re = np.zeros(iters, dtype=np.float64)
    x_im = np.zeros(iters, dtype=np.float64)
    tot_length = 0.0
    big = 0
    centripetal = float(centripetal)
    gaussian_shake = float(gaussian_shake)
    threshold = float(prob_big_shake) * expl                              # :69
    for t in range(iters - 1):
        nd_re = nd_im = 0.0
        if rng.uniform() < threshold:                                     # :69
            e_re, e_im = _cexp(0.0, math.pi + (rng.uniform() - 0.5))     # :70
            a_re, a_im = 2 * v_re, 2 * v_im
            nd_re = a_re * e_re - a_im * e_im
            nd_im = a_re * e_im + a_im * e_re
            big += 1
        g_re = rng.randn()                                                # :76 (real first)
        g_im = rng.randn()
        in_re = gaussian_shake * g_re - centripetal * x_re[t]             # :75-76
        in_im = gaussian_shake * g_im - centripetal * x_im[t]
        dv_re = nd_re + (expl * in_re) * step                             # :75-77
        dv_im = nd_im + (expl * in_im) * step
        v_re = v_re + dv_re                                               # :79
        v_im = v_im + dv_im
        # :80  numpy complex128 / real == multiply by the reciprocal (Smith's form with b.imag=0)
        scl = 1.0 / cabs_numpy(v_re, v_im)
        v_re = (v_re * scl) * step
        v_im = (v_im * scl) * step
        x_re[t + 1] = x_re[t] + v_re                                      # :81
        x_im[t + 1] = x_im[t] + v_im
        # :82  builtin abs() of a complex128 scalar = glibc hypot (probe-verified), unlike np.abs above
        tot_length = tot_length + _m().hypot(x_re[t + 1] - x_re[t], x_im[t + 1] - x_im[t])
    x_re = x_re + canvas / 2                                              # :92
    x_im = x_im + canvas / 2
    return x_re, x_im, tot_length, big


# --------------------------------------------------------------------------------------
# A2  PSF.fit                    (reference motion_blur/generate_PSF.py:31-83)
# --------------------------------------------------------------------------------------

def sample_weight(t, frac, prev, iters):
    """t_proportion of sample t for exposure window (prev, frac].  generate_PSF.py:47-56."""
    fn = frac * iters
    pn = prev * iters
    if fn >= t and pn < t - 1:
        return 1
    if fn >= t - 1 and pn < t - 1:
        return fn - (t - 1)
    if fn >= t and pn < t:
        return t - pn
    if fn >= t - 1 and pn < t:
        return (frac - prev) * iters
    return 0


def psf_rasterize(x_re, x_im, fractions, canvas=256):
    """Returns the list of PSFs (cumulative over `fractions`, as the reference's shared
    accumulator makes them), each canvas x canvas float64.  generate_PSF.py:31-83.
    Row index = imaginary part, column index = real part."""
    iters = len(x_re)
    acc = np.zeros((canvas, canvas), dtype=np.float64)
    out = []
    for j, frac in enumerate(fractions):
        prev = 0 if j == 0 else fractions[j - 1]
        for t in range(iters):
            w = sample_weight(t, frac, prev, iters)
            re, im = float(x_re[t]), float(x_im[t])
            m2 = int(min(canvas - 1, max(1, math.floor(re))))             # :59
            m1 = int(min(canvas - 1, max(1, math.floor(im))))             # :61
            for (row, col) in ((m1, m2), (m1, m2 + 1), (m1 + 1, m2), (m1 + 1, m2 + 1)):  # :64-75
                tri = max(0.0, 1.0 - abs(re - col)) * max(0.0, 1.0 - abs(im - row))
                acc[row, col] += w * tri
        out.append(acc / iters)                                           # :77
    return out


# --------------------------------------------------------------------------------------
# A3  PSF.centerPSF              (reference motion_blur/generate_PSF.py:106-123)
# --------------------------------------------------------------------------------------

def _pairwise(a):
    """numpy's DOUBLE pairwise_sum: blocks of <=128 with 8 interleaved accumulators combined as
    ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)); larger inputs split at n/2 rounded down to x8."""
    n = a.size
    if n < 8:
        s = 0.0
        for v in a:
            s += float(v)
        return s
    if n <= 128:
        r = [float(a[i]) for i in range(8)]
        i = 8
        while i < n - (n % 8):
            for k in range(8):
                r[k] += float(a[i + k])
            i += 8
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        while i < n:
            res += float(a[i])
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return _pairwise(a[:n2]) + _pairwise(a[n2:])


def numpy_sum(a):
    """np.sum of a contiguous float64 array restated (probe-verified bit for bit on 1-D and
    2-D inputs, numpy 2.2.6): the flattened data is consumed in iterator chunks of 8192
    elements; each chunk is pairwise-summed and the chunk sums are accumulated left to right."""
    a = np.ascontiguousarray(a, dtype=np.float64).ravel()
    s = 0.0
    for i in range(0, a.size, 8192):
        s += _pairwise(a[i:i + 8192])
    return s


def psf_center_offsets(psf):
    """(offsetX, offsetY) of generate_PSF.py:106-120: weighted centroid of psf>0 cells,
    accumulated in np.nonzero (row-major) order, minus canvas/2, truncated toward zero."""
    canvas = psf.shape[0]
    total = numpy_sum(psf)                                             # :108  np.sum
    rows, cols = np.nonzero(psf > 0)                                      # :110
    ax = 0.0
    ay = 0.0
    for r, c in zip(rows, cols):                                          # :113-117
        w = float(psf[r, c]) / total
        ax += float(c) * w
        ay += float(r) * w
    return int(ax - canvas / 2), int(ay - canvas / 2)                     # :119-120


def psf_center(psf):
    """generate_PSF.py:106-123: roll the centroid to the canvas centre (circular)."""
    ox, oy = psf_center_offsets(psf)
    n = psf.shape[0]
    ri = (np.arange(n) + oy) % n
    ci = (np.arange(n) + ox) % n
    return psf[np.ix_(ri, ci)].copy()


def psf_crop128(psf):
    """transforms.py:334-335 / :308-309: centre 128x128 window of a 256 canvas."""
    if psf.shape[0] > 128:
        return psf[64:192, 64:192]
    return psf


def make_psf(param, fraction, canvas=256, max_len=96, rng=np.random, center=True):
    """On-the-fly PSF exactly as BlurImage builds it (transforms.py:316-335): two trajectory
    fits (the first only advances the RNG), rasterise, centre, crop."""
    trajectory(canvas, 2000, max_len, param, rng)
    x_re, x_im, _, _ = trajectory(canvas, 2000, max_len, param, rng)
    psf = psf_rasterize(x_re, x_im, [fraction], canvas)[0]
    if not center:
        return psf
    return np.ascontiguousarray(psf_crop128(psf_center(psf)))


# --------------------------------------------------------------------------------------
# A6  host -> device PSF conversion         (reference engine.py:84)
# --------------------------------------------------------------------------------------

def to_half_like_torch(a):
    """torch.HalfTensor(ndarray): float64 -> float32 -> float16 (two roundings; probe-verified
    on 2e6 values, differs from a direct float64->float16 cast in ~5e-5 of them)."""
    a = np.asarray(a)
    if a.dtype == np.float16:
        return a.copy()
    return a.astype(np.float32).astype(np.float16)


# --------------------------------------------------------------------------------------
# A7/A8  blur_image_list / manual_blur       (reference models/blur_functions.py:11-100)
# --------------------------------------------------------------------------------------

def half_sum_exact(psf_h):
    """psf.sum() of a Half tensor, defined here as the exactly-rounded sum: every finite fp16
    is a multiple of 2^-24, so the sum is exact in integers and rounded once to fp16
    (blur_functions.py:98).  torch accumulates in fp32 in an implementation-defined order and
    rounds to fp16; the two agree unless the fp32 error straddles an fp16 rounding boundary
    (never observed on the golden PSFs)."""
    q = np.asarray(psf_h, dtype=np.float16).astype(np.float64) * float(1 << 24)
    total = int(np.sum(q.astype(np.int64)))
    return np.float16(total / float(1 << 24)) if abs(total) < (1 << 53) else _round_int_to_half(total)


def _round_int_to_half(total):
    # exact integer (units of 2^-24) -> fp16 round-to-nearest-even without going through fp64
    sign = -1 if total < 0 else 1
    m = abs(total)
    if m == 0:
        return np.float16(0)
    nbits = m.bit_length()
    if nbits <= 11:
        return np.float16(sign * m * 2.0 ** -24)
    shift = nbits - 11
    q, rem = m >> shift, m & ((1 << shift) - 1)
    half = 1 << (shift - 1)
    if rem > half or (rem == half and (q & 1)):
        q += 1
    return np.float16(sign * float(q) * 2.0 ** (shift - 24))


def normalize_psf(psf):
    """psf / psf.sum() in the PSF's own dtype (blur_functions.py:98, utils.py:372)."""
    psf = np.asarray(psf)
    if psf.dtype == np.float16:
        return (psf / half_sum_exact(psf)).astype(np.float16)
    return (psf / psf.sum(dtype=psf.dtype)).astype(psf.dtype)


def taps_of(psf_norm):
    """Row-major non-zero list of a normalised PSF: (rows, cols, weights).
    blur_functions.py:63 (`nonzero(as_tuple=False)` order = ascending row, then column)."""
    rows, cols = np.nonzero(psf_norm)
    return rows.astype(np.int64), cols.astype(np.int64), psf_norm[rows, cols]


def _pad_index(s, n, mode):
    """Maps an un-padded coordinate s (may lie outside [0,n)) to a source index, or -1 = zero.
    'reflect' has no edge repeat (appendix A.1); 'replicate' clamps; 'constant' zero-fills."""
    s = np.asarray(s)
    if mode == "reflect":
        s = np.where(s < 0, -s, s)
        s = np.where(s > n - 1, 2 * (n - 1) - s, s)
        return s
    if mode == "replicate":
        return np.clip(s, 0, n - 1)
    return np.where((s < 0) | (s > n - 1), -1, s)


SEG_ROWS, SEG_COLS = 12, 24      # a tap segment's bounding box: at most 13 PSF rows x 25 columns (include/dib.h, `Tap tables`)


def tap_segments(rows, cols, seg_rows=SEG_ROWS, seg_cols=SEG_COLS):
    """The library's cut of the row-major tap list into the segments its blur stages in LDS (include/dib.h): greedy runs of
    consecutive taps whose rows span at most seg_rows + 1 and whose columns span at most seg_cols + 1.  Returns [(first, end)].
    Not the reference's arithmetic -- the reference has no segments -- but what fixes the ORDER in which DIB_ACC_FAST16 adds."""
    out, start = [], 0
    n = len(rows)
    while start < n:
        r0, cmin, cmax, end = rows[start], cols[start], cols[start], start
        while end < n:
            lo, hi = min(cmin, cols[end]), max(cmax, cols[end])
            if rows[end] - r0 > seg_rows or hi - lo > seg_cols:
                break
            cmin, cmax, end = lo, hi, end + 1
        out.append((start, end))
        start = end
    return out


def tap_order_vruns(rows, cols):
    """The order in which DIB_ACC_FAST16 (include/dib.h) accumulates the taps: segment by segment; inside a segment the taps of one
    PSF column in consecutive rows form a vertical run; a run of L taps is cut, from its lowest row up, into L // 4 groups of four
    and one group of L % 4; the segment's groups are taken by size -- all fours, then the threes, the twos, the singles -- inside a
    size in the row-major order of the runs' first taps (a run's fours from its lowest rows up), a group from its lowest row to its
    highest.  Returns a permutation of range(len(rows))."""
    order = []
    for a, b in tap_segments(rows, cols):
        index = {(int(rows[j]), int(cols[j])): j for j in range(a, b)}
        by_size = {4: [], 3: [], 2: [], 1: []}
        for j in range(a, b):
            r, c = int(rows[j]), int(cols[j])
            if (r - 1, c) in index:
                continue
            run = []
            while (r, c) in index:
                run.append(index[(r, c)])
                r += 1
            for k in range(0, len(run) - len(run) % 4, 4):
                by_size[4].append(run[k:k + 4])
            if len(run) % 4:
                by_size[len(run) % 4].append(run[len(run) - len(run) % 4:])
        for n in (4, 3, 2, 1):
            for g in by_size[n]:
                order.extend(g)
    assert sorted(order) == list(range(len(rows)))
    return order


def manual_blur(image, psf_norm, fp32_accumulate=False, fma16=False, tap_order=None):
    """models/blur_functions.py:11-69 (both canvas branches), post-ops excluded.

    fp32_accumulate=True restates the library's DIB_ACC_FP32 mode instead of the reference
    arithmetic (fp16 images only): the fp16 x fp16 products are exact in float32, the running sum is
    float32, taps in the same order, one rounding to fp16 at the end.
    fma16=True restates DIB_ACC_FMA16: acc = fp16(acc + P * w) with ONE rounding per tap (a fused
    multiply-add in fp16).  float64 holds acc + P * w exactly whenever the rounding could go either way
    (22-bit product, 11-bit accumulator), so rounding the float64 sum to fp16 is the fused result.
    tap_order: a permutation of the row-major tap list, the order to accumulate in (fma16=True with
    tap_order=tap_order_vruns(rows, cols) restates DIB_ACC_FAST16).

    image: C x H x W float16 or float32;  psf_norm: K x K, same dtype, already normalised.
    out[ch,y,x] = sum over taps (r,c), row-major, of  rnd(rnd(P[(y+2pb-r) mod Hp, (x+2pb-c) mod Wp] * w) + acc)
    with pb = K/2-1, pa = K/2, Hp = H+K-1 and P the padded image (appendix A.2/A.3)."""
    image = np.asarray(image)
    dt = image.dtype
    squeeze_c = False
    if image.ndim == 2:
        image = image[None]
        squeeze_c = True
    C, H, W = image.shape
    K = psf_norm.shape[0]
    if K > 129:                                                           # :17
        big, mode = 256, "replicate"                                      # :24-31
    else:
        big = 128
        mode = "constant" if (H < 64 or W < 64) else "reflect"            # :55-58
    pb, pa = big // 2 - 1, big // 2                                       # :52 / :26
    if mode == "reflect" and (H <= pa or W <= pa):
        raise RuntimeError("Padding size should be less than the corresponding input dimension")
    Hp, Wp = H + pb + pa, W + pb + pa
    src_r = _pad_index(np.arange(Hp) - pb, H, mode)
    src_c = _pad_index(np.arange(Wp) - pb, W, mode)
    rows, cols, wts = taps_of(np.asarray(psf_norm).astype(dt))
    acc = np.zeros((C, H, W), dtype=np.float32 if fp32_accumulate else dt)   # :61
    ys, xs = np.arange(H), np.arange(W)
    if tap_order is not None:
        rows, cols, wts = rows[list(tap_order)], cols[list(tap_order)], wts[list(tap_order)]
    for r, c, w in zip(rows, cols, wts):                                  # :66-67
        pr = src_r[(ys + 2 * pb - r) % Hp]
        pc = src_c[(xs + 2 * pb - c) % Wp]
        g = image[:, np.maximum(pr, 0)][:, :, np.maximum(pc, 0)]
        if mode == "constant":
            g = g * ((pr >= 0)[None, :, None] & (pc >= 0)[None, None, :]).astype(dt)
        if fma16:
            acc = (acc.astype(np.float64) + g.astype(np.float64) * np.float64(w)).astype(dt)
        elif fp32_accumulate:
            acc = acc + g.astype(np.float32) * np.float32(w)
        else:
            acc = (acc + (g * dt.type(w)).astype(dt)).astype(dt)
    out = acc.astype(dt)
    # :69 `.squeeze()` drops every size-1 dim
    return np.squeeze(out)


def blur_image_list(images, blur_dicts, psfs):
    """models/blur_functions.py:92-100: in-place replacement of the blurred entries."""
    for i, (img, bd, psf) in enumerate(zip(images, blur_dicts, psfs)):
        if not bd["blurring"]:
            continue
        images[i] = manual_blur(img, normalize_psf(psf))


# --------------------------------------------------------------------------------------
# A9/A10  expand_targets / fix_bounding_box_squeeze   (reference utils.py:360-434)
# --------------------------------------------------------------------------------------

def psf_extents(psf):
    """(left, top, right, bottom) = (min col, min row, max col, max row) - 63 over the
    non-zeros of the normalised PSF.  utils.py:372-380."""
    if psf.shape[0] != 128:
        raise Exception("Trying to expand with filters that are not 128 wide!")   # :369-370
    rows, cols, _ = taps_of(normalize_psf(psf))
    return int(cols.min()) - 63, int(rows.min()) - 63, int(cols.max()) - 63, int(rows.max()) - 63


def clamp_boxes(boxes, H, W):
    """utils.py:395-434 on an N x 4 float32 xyxy array (returns a new array)."""
    b = np.array(boxes, dtype=np.float32, copy=True)

    def clamp():
        b[:, 0] = np.where(b[:, 0] > W - 1, np.float32(W - 1), b[:, 0])   # :398
        b[:, 1] = np.where(b[:, 1] > H - 1, np.float32(H - 1), b[:, 1])   # :399
        b[:, 2] = np.where(b[:, 2] > W - 1, np.float32(W - 1), b[:, 2])   # :401
        b[:, 3] = np.where(b[:, 3] > H - 1, np.float32(H - 1), b[:, 3])   # :402
        b[...] = np.where(b < 0, np.float32(0), b)                        # :405-409

    clamp()
    bad = b[:, 0] >= b[:, 2]                                              # :412-414
    b[bad, 2] += np.float32(1)
    b[bad, 0] -= np.float32(1)
    bad = b[:, 1] >= b[:, 3]                                              # :416-418
    b[bad, 3] += np.float32(1)
    b[bad, 1] -= np.float32(1)
    clamp()                                                               # :421-432
    return b


def expand_boxes(boxes, psf, H, W):
    """utils.py:360-392 for one image: grow by the PSF extents, then clamp."""
    left, top, right, bottom = psf_extents(psf)
    b = np.array(boxes, dtype=np.float32, copy=True)
    b[:, 0] = b[:, 0] + np.float32(left)                                  # :382
    b[:, 2] = b[:, 2] + np.float32(right)                                 # :383
    b[:, 1] = b[:, 1] + np.float32(top)                                   # :385
    b[:, 3] = b[:, 3] + np.float32(bottom)                                # :386
    return clamp_boxes(b, H, W)


# --------------------------------------------------------------------------------------
# A4(vii)  PSF principal-axis statistics      (reference transforms.py:366-385)
# --------------------------------------------------------------------------------------

def psf_axis_stats(psf):
    """Returns (theta_rad, scale_factor_lambda1, scale_factor_lambda2)."""
    ys, xs = np.nonzero(psf > 0)
    yp = ys - ys.mean()
    xp = xs - xs.mean()
    cov = (yp * xp).mean()
    var_x = (xp * xp).mean()
    var_y = (yp * yp).mean()
    root = math.sqrt(math.pow((var_x - var_y) / 2, 2) + math.pow(cov, 2))
    lam1 = (var_x + var_y) / 2 + root
    lam2 = (var_x + var_y) / 2 - root

    def sig(v):
        return 1 / (1 + math.exp(-v))

    s1 = 1 - (sig(math.sqrt(lam1) / 10) - 0.5) * 0.6
    s2 = 1 - (sig(math.sqrt(lam2) / 10) - 0.5) * 0.6
    theta = -math.atan2(lam1 - var_x, -cov)
    return theta, s1, s2


# --------------------------------------------------------------------------------------
# A11  get_norm_params                        (reference utils.py:219-273)
# --------------------------------------------------------------------------------------

_CANON_MEAN = [0.485, 0.456, 0.406]
_CANON_STD = [0.229, 0.224, 0.225]
# per-exposure std tables (columns: clean, E0..E4), utils.py:228-230
_STD = {
    0: [[0.2384, 0.2334, 0.2370], [0.2337, 0.2288, 0.2325], [0.2270, 0.2221, 0.2261],
        [0.2209, 0.2161, 0.2203], [0.2127, 0.2082, 0.2126], [0.2087, 0.2043, 0.2088]],
    1: [[0.2384, 0.2334, 0.2370], [0.2337, 0.2287, 0.2325], [0.2267, 0.2218, 0.2258],
        [0.2184, 0.2137, 0.2180], [0.2048, 0.2006, 0.2051], [0.1950, 0.1911, 0.1957]],
    2: [[0.2384, 0.2334, 0.2370], [0.2337, 0.2287, 0.2325], [0.2266, 0.2217, 0.2258],
        [0.2182, 0.2136, 0.2178], [0.2012, 0.1972, 0.2017], [0.1824, 0.1790, 0.1838]],
}


def norm_params(blur_dicts, use_custom_image_norm):
    if blur_dicts is None:
        return np.array([_CANON_MEAN]), np.array([_CANON_STD])
    means = np.zeros((len(blur_dicts), 3))
    stds = np.zeros((len(blur_dicts), 3))
    for i, bd in enumerate(blur_dicts):
        if use_custom_image_norm and bd["blurring"] and bd["param_index"] is not None:   # :254
            fi = bd["fraction_index"]
            pi = bd["param_index"]
            if fi == -1:                                                  # :258-261
                means[i], stds[i] = _CANON_MEAN, _CANON_STD
            elif pi in (0, 1, 2):                                         # :263-271
                means[i] = _CANON_MEAN
                stds[i] = ((np.asarray(_STD[pi]).T * 0.229) / 0.2384)[:, fi + 1]
            # any other param_index (e.g. -1 from the stored-PSF off-by-one, transforms.py:427-428)
            # leaves the row at its initial zeros, as the reference does
        else:
            means[i], stds[i] = _CANON_MEAN, _CANON_STD
    return means, stds


# --------------------------------------------------------------------------------------
# A18  BlurImageHandler (--cpu_blur, FFT)     (reference motion_blur/blur_image.py:23-154)
# --------------------------------------------------------------------------------------

def _minmax01(a):
    """cv2.normalize(..., 0, 1, NORM_MINMAX, CV_32F): global min/max over all channels."""
    a = np.asarray(a, dtype=np.float64)
    lo, hi = float(a.min()), float(a.max())
    scale = 1.0 / (hi - lo) if hi > lo else 0.0
    return ((a - lo) * scale).astype(np.float32)


def cpu_fft_blur(image_u8, psf):
    """`--cpu_blur`: image_u8 H x W x 3 uint8, psf k x k (k <= H, W) -> H x W x 3 uint8.
    blur_image.py:78-85 (edge pad k/2), :113-123 (zero-pad the PSF to the image), :128-134
    (min-max, fftconvolve 'same' per channel, min-max), :137-147 (un-pad, x255 -> uint8).
    The upscale branch for images smaller than the PSF (:56-69) is not restated (bicubic PIL
    resize + cv2 Lanczos); such inputs raise."""
    from scipy import signal
    img = np.asarray(image_u8)
    H, W = img.shape[:2]
    k = psf.shape[0]
    if H < k or W < k:
        raise NotImplementedError("image smaller than the PSF: resize branch not restated")
    pr = int(round(k / 2))
    orig = np.pad(img, ((pr, pr), (pr, pr), (0, 0)), mode="edge")
    yN, xN = orig.shape[:2]
    dY, dX = yN - k, xN - k
    tmp = np.pad(np.asarray(psf, dtype=np.float32),
                 ((dY // 2, math.ceil(dY / 2)), (math.ceil(dX / 2), dX // 2)), "constant")
    tmp = _minmax01(tmp)
    blurred = _minmax01(orig)
    for ch in range(3):
        blurred[:, :, ch] = signal.fftconvolve(blurred[:, :, ch], tmp, "same")
    blurred = _minmax01(blurred)
    blurred = blurred[pr:blurred.shape[0] - pr, pr:blurred.shape[1] - pr, :]
    return (blurred * 255).astype(np.uint8)
