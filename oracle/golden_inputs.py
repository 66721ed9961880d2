"""Seeded input recipes shared by `oracle/gen_goldens.py` (which feeds them to the real
reference) and by the tests (which feed them to the oracle and to the HIP path).

TEST INFRASTRUCTURE.  Everything here is deterministic: legacy `np.random.RandomState`
streams (stable across numpy versions) or data read back from tests/golden/psf.npz.
"""
import os
import random

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

PARAMS = [0.005, 0.001, 0.00005]                 # transforms.py:249
FRACTIONS = [1 / 18, 1 / 10, 1 / 5, 1 / 2, 1, 1 / 25]   # transforms.py:250 + evaluate.py:303 (1/25)
TRAJ_SEEDS = [0, 1337, 2674]


def psf_seed(param, fi):
    return 1000 + 10 * PARAMS.index(param) + fi


_psf_store = None


def psf_store():
    global _psf_store
    if _psf_store is None:
        _psf_store = np.load(os.path.join(GOLDEN_DIR, "psf.npz"))
    return _psf_store


def golden_psf(param, fi, which="crop"):
    """Rebuilds a golden PSF array from its stored non-zero list.
    which: raw|cen (256x256 f64), crop (128x128 f64), half (128x128 f16), norm (128x128 f16)."""
    s = psf_store()
    key = "psf_p%g_f%d" % (param, fi)
    if which == "norm":
        rc = s[key + "_norm_rc"].astype(np.int64)
        a = np.zeros((128, 128), dtype=np.float16)
        a[rc[:, 0], rc[:, 1]] = s[key + "_norm_w"].view(np.float16)
        return a
    n = 256 if which in ("raw", "cen") else 128
    v = s[key + "_%s_v" % which]
    if which == "half":
        v = v.view(np.float16)
    a = np.zeros((n, n), dtype=v.dtype)
    a[s[key + "_%s_r" % which].astype(np.int64), s[key + "_%s_c" % which].astype(np.int64)] = v
    return a


# ------------------------------------------------------------------ blur cases

def blur_cases():
    c = []

    def add(name, shape, dtype, seed, psf, **kw):
        c.append(dict(name=name, shape=shape, dtype=dtype, seed=seed, psf=psf, **kw))

    add("reflect_f16", (3, 96, 130), "float16", 1, ("golden", 0.005, 1))
    add("reflect_f32", (3, 96, 130), "float32", 1, ("golden", 0.005, 1))
    add("zero_both_f16", (3, 50, 70), "float16", 2, ("golden", 0.001, 2))
    add("zero_h_f16", (3, 63, 200), "float16", 3, ("golden", 0.001, 0))
    add("zero_w_f16", (3, 200, 40), "float16", 4, ("golden", 0.00005, 1))
    add("medium_e4_f16", (3, 200, 300), "float16", 5, ("golden", 0.005, 4))
    add("medium_e3_f32", (3, 120, 150), "float32", 6, ("golden", 0.00005, 3))
    add("onechan_f16", (1, 80, 90), "float16", 7, ("golden", 0.001, 1))
    add("min_reflect_f16", (3, 65, 65), "float16", 8, ("golden", 0.005, 3))
    for (r, cc) in ((0, 0), (63, 63), (127, 127), (127, 0), (0, 127), (64, 62), (127, 63)):
        add("tap_%d_%d_f16" % (r, cc), (3, 70, 75), "float16", 9, ("single", 128, r, cc))
    add("tap_127_127_zero_f16", (2, 40, 90), "float16", 10, ("single", 128, 127, 127))
    add("corners_f16", (3, 66, 81), "float16", 11, ("random", 128, 40, 12, "corners"))
    add("dense_f16", (3, 70, 80), "float16", 12, ("random", 128, 420, 13, "blob"))
    add("subnormal_f16", (3, 70, 80), "float16", 14, ("random", 128, 30, 15, "tiny"), scale=900.0)
    add("canvas256_f16", (3, 150, 140), "float16", 16, ("random", 256, 60, 17, "corners"))
    add("canvas256_small_f16", (2, 40, 50), "float16", 18, ("random", 256, 25, 19, "blob"))
    add("canvas256_f32", (3, 130, 129), "float32", 20, ("random", 256, 33, 21, "blob"))
    add("full_e1_f16", (3, 800, 1333), "float16", 1337, ("golden", 0.005, 1), digest_only=True)
    add("full_e2_f16", (3, 800, 1333), "float16", 1338, ("golden", 0.005, 2), digest_only=True)
    add("coco_e2_f16", (3, 480, 640), "float16", 1339, ("golden", 0.001, 2), digest_only=True)
    # configs[4]'s heaviest cells: full exposure, ~200-250 taps in several LDS segments (round 3)
    add("full_e4_f16", (3, 800, 1333), "float16", 1340, ("golden", 0.005, 4), digest_only=True)
    add("full_e4_p3_f16", (3, 800, 1333), "float16", 1341, ("golden", 0.00005, 4), digest_only=True)
    return c


def make_image(case):
    rs = np.random.RandomState(case["seed"])
    img = rs.random_sample(case["shape"]).astype(np.float32) * case.get("scale", 1.0)
    return img.astype(case["dtype"])


def _random_psf(K, ntaps, seed, style):
    rs = np.random.RandomState(seed)
    a = np.zeros((K, K), dtype=np.float64)
    if style == "corners":
        # taps on the outermost rows/cols (wrap quirk) plus a few interior ones
        for (r, c) in ((0, 0), (0, K - 1), (K - 1, 0), (K - 1, K - 1), (K - 1, K // 2), (K // 2, K - 1),
                       (K // 2 - 1, K // 2 - 1), (0, K // 2)):
            a[r, c] = rs.random_sample() + 0.05
        rr = rs.randint(0, K, size=ntaps)
        cc = rs.randint(0, K, size=ntaps)
        a[rr, cc] = rs.random_sample(ntaps) + 0.05
    elif style == "blob":
        rr = np.clip((rs.randn(ntaps) * 9 + K // 2 - 1).astype(int), 0, K - 1)
        cc = np.clip((rs.randn(ntaps) * 9 + K // 2 - 1).astype(int), 0, K - 1)
        a[rr, cc] = rs.random_sample(ntaps) + 0.01
    elif style == "tiny":
        rr = np.clip((rs.randn(ntaps) * 5 + K // 2 - 1).astype(int), 0, K - 1)
        cc = np.clip((rs.randn(ntaps) * 5 + K // 2 - 1).astype(int), 0, K - 1)
        a[rr, cc] = np.exp(rs.uniform(np.log(6e-8), np.log(1e-3), ntaps))   # fp16 subnormal range
        a[K // 2 - 1, K // 2 - 1] = 0.97
        return a     # deliberately NOT normalised: keeps the subnormal weights
    return a / a.sum()


def make_case_psf(case):
    """Normalised PSF (numpy, dtype of the case) handed to manual_blur."""
    spec = case["psf"]
    dt = np.dtype(case["dtype"])
    if spec[0] == "golden":
        if dt == np.float16:
            return golden_psf(spec[1], spec[2], "norm")
        p = golden_psf(spec[1], spec[2], "crop").astype(np.float32)
        return (p / np.float32(p.sum(dtype=np.float64))).astype(np.float32)
    if spec[0] == "single":
        a = np.zeros((spec[1], spec[1]), dtype=dt)
        a[spec[2], spec[3]] = 1
        return a
    if spec[0] == "random":
        return _random_psf(spec[1], spec[2], spec[3], spec[4]).astype(dt)
    raise ValueError(spec)


def make_list_case():
    """blur_image_list inputs: ragged fp16 images, UN-normalised fp16 PSFs (as engine.py:84
    produces them), one non-blurred entry carrying the [0] placeholder PSF."""
    shapes = [(3, 90, 120), (3, 70, 64 + 3), (3, 48, 100), (3, 101, 67)]
    flags = [True, False, True, True]
    specs = [(0.005, 0), None, (0.001, 2), (0.00005, 5)]
    imgs, dicts, psfs = [], [], []
    for i, (sh, fl, sp) in enumerate(zip(shapes, flags, specs)):
        imgs.append(np.random.RandomState(100 + i).random_sample(sh).astype(np.float16))
        dicts.append({"blurring": fl})
        psfs.append(golden_psf(sp[0], sp[1], "half") if sp else np.zeros(1, dtype=np.float16))
    return imgs, dicts, psfs


# ------------------------------------------------------------------ box cases

def box_cases():
    return [dict(name="e1", seed=31, n=12, shape=(3, 240, 320), psf=(0.005, 1)),
            dict(name="e4", seed=32, n=20, shape=(3, 120, 160), psf=(0.001, 4)),
            dict(name="e3_small", seed=33, n=9, shape=(3, 40, 30), psf=(0.00005, 3)),
            dict(name="one", seed=34, n=1, shape=(3, 300, 500), psf=(0.005, 2))]


def make_box_case(case):
    rs = np.random.RandomState(case["seed"])
    _, H, W = case["shape"]
    n = case["n"]
    x1 = rs.uniform(0, W - 1, n)
    y1 = rs.uniform(0, H - 1, n)
    w = rs.uniform(0, W / 2, n)
    h = rs.uniform(0, H / 2, n)
    boxes = np.stack([x1, y1, np.minimum(x1 + w, W), np.minimum(y1 + h, H)], 1).astype(np.float32)
    if n >= 6:   # border-hugging, degenerate and inverted boxes
        boxes[0] = [0, 0, W, H]
        boxes[1] = [W - 1, H - 1, W - 1, H - 1]
        boxes[2] = [5.5, 7.25, 5.5, 7.25]
        boxes[3] = [0, 0, 0.5, 0.5]
        boxes[4] = [W - 2, 3, W + 40, 9]
        boxes[5] = [30, H + 10, 35, H + 20]
    psf = golden_psf(case["psf"][0], case["psf"][1], "half")
    return boxes, psf, case["shape"]


def make_squeeze_boxes():
    return np.array([[10, 10, 20, 20], [-5, -7, 3, 4], [149, 99, 149, 99], [160, 120, 170, 130],
                     [50, 60, 40, 30], [0, 0, 0, 0], [148.5, 20, 149.5, 21], [-3, 50, -1, 55],
                     [70.25, 80.5, 70.25, 99.75]], dtype=np.float32)


# ------------------------------------------------------------------ BlurImage cases

def blurimage_modes():
    return [
        dict(name="low", seed=1337, calls=6, kwargs=dict(prob=0.75, low_exposure=True)),
        dict(name="high", seed=1338, calls=3, kwargs=dict(prob=1, high_exposure=True)),
        dict(name="all", seed=1339, calls=5, kwargs=dict(prob=0.9)),
        dict(name="fixed", seed=1340, calls=2, kwargs=dict(prob=1, blur_type=0.001, blur_exposure=1 / 25)),
        dict(name="fixed_tiny", seed=1341, calls=1, kwargs=dict(prob=1, blur_type=0.00005, blur_exposure=1 / 100)),
        dict(name="lehe", seed=1342, calls=4, kwargs=dict(prob=0.5, LEHE_blur_seg=True)),
        dict(name="nocenter", seed=1343, calls=2, kwargs=dict(prob=1, low_exposure=True, dont_center_psf=True)),
        dict(name="stored_low", seed=1344, calls=6, kwargs=dict(prob=0.75, low_exposure=True, use_stored_psfs=True)),
        dict(name="stored_fixed", seed=1345, calls=3,
             kwargs=dict(prob=1, blur_type=2, blur_exposure=3, use_stored_psfs=True)),
        dict(name="stored_all", seed=1346, calls=4, kwargs=dict(prob=0.9, use_stored_psfs=True)),
        dict(name="stored_high", seed=1347, calls=3, kwargs=dict(prob=1, high_exposure=True, use_stored_psfs=True)),
        # defocus: a Gaussian of random width over the PSF, then division by the maximum (reference transforms.py:338-342)
        dict(name="dilate", seed=1348, calls=2, kwargs=dict(prob=1, low_exposure=True, dilate_psf=True)),
    ]


def predict_stored_draw(kw):
    """Replays BlurImage's Python-`random` draw order on the CURRENT
    global `random` state and returns the (param_dir, fraction_dir, psf_index) the reference
    will open, or None when the call will not blur.  The caller restores the state."""
    thr = (1 - 0.0625) if kw.get("LEHE_blur_seg") else kw.get("prob", 0.5)
    if not random.random() < thr:
        return None
    if kw.get("blur_exposure") is None:
        if kw.get("high_exposure"):
            random.choice(range(2))
        elif kw.get("low_exposure"):
            random.choice(range(3))
        elif kw.get("LEHE_blur_seg"):
            random.choices(range(5), weights=[0.0625, 0.0625, 0.0625, 0.375, 0.375])
        else:
            random.choice(range(5))
    if kw.get("blur_type") is None:
        random.choice(range(3))
    p = kw["blur_type"] if kw.get("blur_type") is not None else random.choice([1, 2, 3])
    if kw.get("blur_exposure") is not None:
        e = kw["blur_exposure"]
    elif kw.get("high_exposure"):
        e = random.choice([3, 4])
    elif kw.get("low_exposure"):
        e = random.choice([0, 1, 2])
    elif kw.get("LEHE_blur_seg"):
        e = random.choices([0, 1, 2, 3, 4], weights=[0.0625, 0.0625, 0.0625, 0.375, 0.375])[0]
    else:
        e = random.choice([0, 1, 2, 3, 4])
    return p, e, random.randint(0, 12000 - 1)


def stored_psf(p, e, idx):
    """Synthetic float16 256x256 stored PSF (format A19), deterministic in (p, e, idx)."""
    rs = np.random.RandomState((p * 7919 + e * 104729 + idx) % (2 ** 31))
    n = 10 + 12 * e
    a = np.zeros((256, 256), dtype=np.float64)
    t = np.cumsum(rs.randn(n, 2) * (0.6 + 0.5 * e), axis=0)
    t -= t.mean(axis=0)
    rr = np.clip(np.round(t[:, 0]).astype(int) + 128, 70, 185)
    cc = np.clip(np.round(t[:, 1]).astype(int) + 128, 70, 185)
    a[rr, cc] = rs.random_sample(n) + 0.1
    return (a / a.sum()).astype(np.float16)


# ------------------------------------------------------------------ norm / fft

def norm_dicts():
    d = [{"blurring": False, "param_index": None, "fraction_index": None}]
    for pi in (0, 1, 2, -1, None):
        for fi in (-1, 0, 1, 2, 3, 4):
            d.append({"blurring": True, "param_index": pi, "fraction_index": fi})
    d.append({"blurring": False, "param_index": 1, "fraction_index": 2})
    return d


def make_fft_image():
    return (np.random.RandomState(77).random_sample((200, 260, 3)) * 255).astype(np.uint8)


def make_fft_psf():
    return golden_psf(0.005, 2, "crop")
