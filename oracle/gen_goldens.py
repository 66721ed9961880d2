#!/usr/bin/env python3
"""Golden-vector generator: runs the REAL reference (imported read-only from /root/reference
through oracle/ref_harness.py) on seeded inputs and writes small fixtures to tests/golden/.

Run in the build container only:   python oracle/gen_goldens.py
The fixtures are data (inputs are regenerated from legacy numpy RandomState seeds, outputs are
stored); no reference source text is written anywhere.  Input recipes live in
`oracle/golden_inputs.py` so that tests regenerate exactly the same inputs.
"""
import hashlib
import json
import os
import random
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import golden_inputs as GI  # noqa: E402
import ref_harness  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def nz_list(a):
    rows, cols = np.nonzero(a)
    return rows.astype(np.int16), cols.astype(np.int16), a[rows, cols]


def gen_trajectories(ns, store):
    for param in GI.PARAMS:
        for seed in GI.TRAJ_SEEDS:
            np.random.seed(seed)
            tr = ns.generate_trajectory.Trajectory(canvas=256, max_len=96, expl=param).fit()
            first = tr.x.copy()
            tr = tr.fit()
            key = "traj_p%g_s%d" % (param, seed)
            store[key + "_fit1"] = first
            store[key + "_fit2"] = tr.x.copy()
            store[key + "_len"] = np.array([tr.tot_length, tr.big_expl_count], dtype=np.float64)
            store[key + "_next"] = np.array([np.random.uniform(), np.random.randn()])
    # expl=None constructor draw + a parameter with frequent big shakes (exercises the cexp branch)
    np.random.seed(7)
    tr = ns.generate_trajectory.Trajectory(canvas=64, iters=500, max_len=60).fit()
    store["traj_none_s7"] = tr.x.copy()
    store["traj_none_s7_expl"] = np.array([tr.expl, tr.tot_length, tr.big_expl_count])
    np.random.seed(11)
    tr = ns.generate_trajectory.Trajectory(canvas=256, iters=2000, max_len=96, expl=0.9).fit()
    store["traj_big_s11"] = tr.x.copy()
    store["traj_big_s11_len"] = np.array([tr.tot_length, tr.big_expl_count])


def gen_psfs(ns, store):
    for param in GI.PARAMS:
        for fi, frac in enumerate(GI.FRACTIONS):
            seed = GI.psf_seed(param, fi)
            np.random.seed(seed)
            tr = ns.generate_trajectory.Trajectory(canvas=256, max_len=96, expl=param).fit()
            tr = tr.fit()
            p = ns.generate_PSF.PSF(canvas=256, trajectory=tr, fraction=[frac])
            raw = p.fit()[0].copy()
            p.centerPSF()
            cen = p.PSFs[0].copy()
            crop = cen[64:192, 64:192].copy()
            half = torch.HalfTensor(crop)
            norm = half / half.sum()
            key = "psf_p%g_f%d" % (param, fi)
            for nm, arr in (("raw", raw), ("cen", cen), ("crop", crop)):
                r, c, v = nz_list(arr)
                store[key + "_%s_r" % nm], store[key + "_%s_c" % nm], store[key + "_%s_v" % nm] = r, c, v
            r, c, v = nz_list(half.numpy())
            store[key + "_half_r"], store[key + "_half_c"] = r, c
            store[key + "_half_v"] = v.view(np.uint16)
            nzp = norm.nonzero(as_tuple=False).numpy()
            store[key + "_norm_rc"] = nzp.astype(np.int16)
            store[key + "_norm_w"] = norm[nzp[:, 0], nzp[:, 1]].numpy().view(np.uint16)
            store[key + "_sum"] = half.sum().numpy().reshape(1).view(np.uint16)
    # multi-fraction cumulative list (PSF API parity, generate_PSF.py:39-77)
    np.random.seed(5)
    tr = ns.generate_trajectory.Trajectory(canvas=128, iters=300, max_len=40, expl=0.005).fit()
    p = ns.generate_PSF.PSF(canvas=128, trajectory=tr, fraction=[1 / 100, 1 / 10, 1 / 2, 1])
    for i, a in enumerate(p.fit()):
        store["psf_multi_%d" % i] = a.copy()
    store["psf_multi_traj"] = tr.x.copy()


def gen_blur(ns, store, meta):
    mb = ns.blur_functions.manual_blur
    for case in GI.blur_cases():
        img = GI.make_image(case)
        psf = torch.from_numpy(GI.make_case_psf(case))   # normalised, dtype of the case
        t_img = torch.from_numpy(img)
        out = mb(t_img, psf).numpy()
        name = "blur_" + case["name"]
        if case.get("digest_only"):
            meta[name] = {"sha256": hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest(),
                          "shape": list(out.shape), "dtype": str(out.dtype)}
            store[name + "_sample"] = np.ascontiguousarray(out[..., ::37, ::41]).view(
                np.uint16 if out.dtype == np.float16 else np.uint32)
        else:
            store[name] = out.view(np.uint16 if out.dtype == np.float16 else np.uint32)
        print("  blur case", case["name"], out.shape, out.dtype)
    # blur_image_list: un-normalised PSFs, mixed blurring flags, ragged sizes
    imgs, dicts, psfs = GI.make_list_case()
    t_imgs = [torch.from_numpy(a) for a in imgs]
    t_psfs = [torch.from_numpy(a) for a in psfs]
    ret = ns.blur_functions.blur_image_list(t_imgs, dicts, t_psfs)
    assert ret is None
    for i, t in enumerate(t_imgs):
        store["blurlist_%d" % i] = t.numpy().view(np.uint16)


def gen_boxes(ns, store):
    for case in GI.box_cases():
        boxes, psf, shape = GI.make_box_case(case)
        tgt = [{"boxes": torch.from_numpy(boxes.copy())}]
        out = ns.utils.expand_targets(tgt, [{"blurring": True}], [torch.from_numpy(psf)],
                                      [torch.zeros(shape, dtype=torch.float16)])
        store["boxes_" + case["name"]] = out[0]["boxes"].numpy().view(np.uint32)
    # fix_bounding_box_squeeze alone
    b = GI.make_squeeze_boxes()
    t = {"boxes": torch.from_numpy(b.copy())}
    ns.utils.fix_bounding_box_squeeze(t, (3, 100, 150))
    store["boxes_squeeze"] = t["boxes"].numpy().view(np.uint32)


class _PsfTree:
    """Creates, on demand, only the stored-PSF files a seeded reference call will open."""

    def __init__(self, root):
        self.root = root

    def ensure(self, p, e, idx):
        d = os.path.join(self.root, "P%dE%d" % (p, e))
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, "I%06d" % idx)
        if not os.path.exists(path):
            with open(path, "wb") as f:
                np.save(f, GI.stored_psf(p, e, idx))
        return path


def gen_blurdicts(ns, meta, store):
    BlurImage = ns.transforms.BlurImage
    out = {}
    tmp = tempfile.mkdtemp(prefix="dib_psfs_")
    tree = _PsfTree(tmp)
    for mode in GI.blurimage_modes():
        kw = dict(mode["kwargs"])
        if kw.get("use_stored_psfs"):
            kw["stored_psf_directory"] = tmp
        recs = []
        random.seed(mode["seed"])
        np.random.seed(mode["seed"])
        bi = BlurImage(blur_image_in_transform=False, **kw)
        for call in range(mode["calls"]):
            if kw.get("use_stored_psfs"):
                # the reference opens P{p}E{e}/I{idx}: pre-create exactly that file by
                # replaying the documented draw order on a copy of the RNG state
                st = random.getstate()
                pred = GI.predict_stored_draw(kw)
                random.setstate(st)
                if pred is not None:
                    tree.ensure(*pred)
            img, tgt, bd = bi("IMG", {"t": call}, {})
            rec = {"blurring": bool(bd["blurring"])}
            if bd["blurring"]:
                psf = np.asarray(bd["psf"])
                r, c, v = nz_list(psf)
                k = "bd_%s_%d" % (mode["name"], call)
                store[k + "_r"], store[k + "_c"] = r, c
                store[k + "_v"] = v
                rec.update(psf_dtype=str(psf.dtype), psf_shape=list(psf.shape),
                           theta_rad=float(bd["theta_rad"]).hex(),
                           scale_factor_lambda1=float(bd["scale_factor_lambda1"]).hex(),
                           scale_factor_lambda2=float(bd["scale_factor_lambda2"]).hex(),
                           param_index=None if bd["param_index"] is None else int(bd["param_index"]),
                           fraction_index=None if bd["fraction_index"] is None else int(bd["fraction_index"]))
            else:
                rec.update(psf=list(bd["psf"]), theta_rad=bd["theta_rad"],
                           scale_factor_lambda1=bd["scale_factor_lambda1"],
                           scale_factor_lambda2=bd["scale_factor_lambda2"],
                           param_index=bd["param_index"], fraction_index=bd["fraction_index"])
            assert img == "IMG" and tgt == {"t": call}
            recs.append(rec)
        out[mode["name"]] = {"records": recs, "next_random": random.random(),
                             "next_np": float(np.random.uniform())}
    # preBlurred pass-through (transforms.py:225-235)
    bi = BlurImage(prob=1.0, blur_image_in_transform=False)
    _, _, bd = bi("IMG", None, {"preBlurred": True})
    out["preblurred"] = {k: (v if not isinstance(v, np.generic) else v.item()) for k, v in bd.items()}
    meta["blurimage"] = out


def gen_norm(ns, store):
    dicts = GI.norm_dicts()
    for flag in (False, True):
        m, s = ns.utils.get_norm_params(dicts, flag)
        store["norm_means_%d" % flag] = m
        store["norm_stds_%d" % flag] = s
    m, s = ns.utils.get_norm_params(None, True)
    store["norm_none_means"], store["norm_none_stds"] = m, s


def gen_fft(ns, store):
    from PIL import Image
    img = GI.make_fft_image()
    psf = GI.make_fft_psf()
    h = ns.blur_image.BlurImageHandler(image_path=None, PSFs=[psf.astype(np.float32)],
                                       pillowImage=Image.fromarray(img))
    assert h.blur_image()
    store["fft_out"] = np.array(h.pilImageResult)


def warper_inputs():
    g = torch.Generator().manual_seed(99)
    x = torch.rand(3, 3, 40, 56, generator=g)
    feat = torch.randn(3, 8, 10, 14, generator=g)
    th = torch.tensor([0.3, -1.1, 2.4]).half()
    l1 = torch.tensor([0.9, 0.8, 1.0]).half()
    l2 = torch.tensor([1.0, 0.95, 0.72]).half()
    return x, feat, th, l1, l2


def warper_smooth_input():
    """A low-frequency image (11 x 11 box blur of noise, contrast restored): Half grid coordinates jitter the sampling
    position by ~1e-2 px, which moves a smooth image by ~1e-3 -- while a warp that is off by a pixel moves it by ~5e-2."""
    g = torch.Generator().manual_seed(123)
    x = torch.rand(3, 3, 40, 56, generator=g)
    x = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(x, (5, 5, 5, 5), mode="reflect"), 11, 1)
    return ((x - x.amin(dim=(1, 2, 3), keepdim=True)) / (x.amax(dim=(1, 2, 3), keepdim=True) - x.amin(dim=(1, 2, 3), keepdim=True))).contiguous()


def gen_warper(ns, store):
    """Reference models/warper.py on CPU: image warp and the inverse-scale feature warp."""
    import importlib
    w = importlib.import_module("models.warper").Warper()
    x, feat, th, l1, l2 = warper_inputs()
    store["warp_image"] = w(x, th, l1, l2).numpy()
    store["warp_feature"] = w(feat, th, 1 / l1, 1 / l2).numpy()
    store["warp_smooth"] = w(warper_smooth_input(), th, l1, l2).numpy()


def net_transform_inputs():
    """Two ragged images with boxes, per-image statistics as engine.py passes them (float64 numpy rows)."""
    g = torch.Generator().manual_seed(4711)
    imgs = [torch.rand(3, 50, 70, generator=g), torch.rand(3, 60, 45, generator=g)]
    tgts = [{"boxes": torch.tensor([[3.5, 4.25, 40.0, 44.5], [10.0, 0.0, 69.0, 49.0]]), "labels": torch.tensor([3, 7])},
            {"boxes": torch.tensor([[1.0, 2.0, 30.5, 59.0]]), "labels": torch.tensor([1])}]
    means = np.array([[0.485, 0.456, 0.406], [0.485, 0.456, 0.406]])
    stds = np.array([[0.229, 0.224, 0.225], [0.11716, 0.11548, 0.11734]])
    return imgs, tgts, means, stds


def gen_net_transforms(ns, store):
    """Reference models/net_transforms.py: normalise -> resize -> batch (+ box rescaling, crop mode, postprocess)."""
    NT = ref_harness.load_net_transforms()
    imgs, tgts, means, stds = net_transform_inputs()
    for tag, training, use_stats in (("train", True, True), ("eval", False, False)):
        t = NT.GeneralizedRCNNTransform(64, 100, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], training=training)
        torch.manual_seed(5)
        il, out = t([i.clone() for i in imgs], [{k: v.clone() for k, v in d.items()} for d in tgts],
                    newMeans=means if use_stats else None, newSTDs=stds if use_stats else None)
        store["nt_%s_batch" % tag] = il.tensors.numpy()
        store["nt_%s_sizes" % tag] = np.array(il.image_sizes)
        for k, d in enumerate(out):
            store["nt_%s_boxes%d" % (tag, k)] = d["boxes"].numpy()
    # an image that already has the target size: scale factor exactly 1
    t = NT.GeneralizedRCNNTransform(64, 100, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], training=False)
    one = torch.rand(3, 64, 100, generator=torch.Generator().manual_seed(8))
    il, _ = t([one.clone()], None)
    store["nt_unit_batch"] = il.tensors.numpy()
    # the blur estimator's batcher (engine.py:262-264): crop to the smallest image, floored to a multiple of 32
    t = NT.GeneralizedRCNNTransform(64, 100, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], crop_images=True)
    torch.manual_seed(5)
    il, _ = t([i.clone() for i in imgs], None)
    store["nt_crop_batch"] = il.tensors.numpy()
    store["nt_crop_sizes"] = np.array(il.image_sizes)
    # eval-mode postprocess: boxes back to the original image sizes
    t = NT.GeneralizedRCNNTransform(64, 100, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], training=False)
    res = t.postprocess([{"boxes": torch.tensor([[4.0, 8.0, 60.0, 50.0]])}, {"boxes": torch.tensor([[2.0, 2.0, 40.0, 90.0]])}],
                        [(64, 90), (100, 75)], [(50, 70), (60, 45)])
    for k, d in enumerate(res):
        store["nt_post_boxes%d" % k] = d["boxes"].numpy()


def postop_input():
    x = torch.rand(3, 40, 56, generator=torch.Generator().manual_seed(77))
    delta = torch.zeros(128, 128)
    delta[63, 63] = 1.0            # manual_blur with this PSF returns its input: what follows is the post-op chain alone
    return x, delta


POSTOP_SEEDS = (0, 1, 3, 5)


def gen_postops(ns, store):
    """Reference models/blur_functions.py:72-81 on the CPU: Gaussian noise + clamp, nearest-neighbour "block"
    down/up-sampling, each after the reference's own numpy draws (seeded per case)."""
    x, delta = postop_input()
    for seed in POSTOP_SEEDS:
        np.random.seed(seed); torch.manual_seed(seed)
        store["noise_%d" % seed] = ns.blur_functions.manual_blur(x.clone(), delta, add_noise=True, noise_level=0.01).numpy()
        np.random.seed(seed); torch.manual_seed(seed)
        store["block_%d" % seed] = ns.blur_functions.manual_blur(x.clone(), delta, add_block=True).numpy()
        np.random.seed(seed); torch.manual_seed(seed)
        store["both_%d" % seed] = ns.blur_functions.manual_blur(x.clone(), delta, add_noise=True, noise_level=0.004, add_block=True).numpy()
        store["rng_after_%d" % seed] = np.array([np.random.uniform()])      # the draw order leaves numpy's stream here


def jpeg_input():
    g = torch.Generator().manual_seed(2024)
    x = torch.rand(2, 3, 32, 48, generator=g)
    x[1] = torch.nn.functional.avg_pool2d(x[1:2], 3, 1, 1)[0]       # one smooth image
    return x


JPEG_QUALITIES = (20, 49, 50, 75, 90)


def gen_jpeg(ns, store):
    """Reference models/jpeg/DiffJPEG.py (differentiable=False) round trips on CPU."""
    import importlib
    J = importlib.import_module("models.jpeg.DiffJPEG")
    x = jpeg_input()
    m = J.DiffJPEG(height=100, width=100, differentiable=False, quality=10)
    m.setRes(32, 48)
    for q in JPEG_QUALITIES:
        m.setQuality(q)
        with torch.no_grad():
            store["jpeg_q%d" % q] = m(x).numpy()


def coco_eval_inputs():
    """Synthetic COCO-shaped ground truth and detections: 12 images, 6 categories, crowd boxes, all three
    size classes, duplicates, misses and false positives."""
    rs = np.random.RandomState(77)
    gt, dt = {}, {}
    for img in range(1, 13):
        n = rs.randint(0, 9)
        side = np.exp(rs.uniform(np.log(6), np.log(300), (n, 2)))
        xy = rs.uniform(0, 400, (n, 2))
        boxes = np.concatenate([xy, xy + side], 1)
        labels = rs.randint(1, 7, n)
        crowd = (rs.random_sample(n) < 0.12).astype(np.int64)
        gt[img] = dict(boxes=boxes, labels=labels, iscrowd=crowd, area=side[:, 0] * side[:, 1] * rs.uniform(0.5, 1.0, n))
        db, dl, ds = [], [], []
        for b, l in zip(boxes, labels):
            for _ in range(rs.randint(0, 3)):                      # 0..2 detections per object, jittered
                j = b + rs.normal(0, 0.12, 4) * np.tile(b[2:] - b[:2], 2)
                db.append([min(j[0], j[2]), min(j[1], j[3]), max(j[0], j[2]) + 1, max(j[1], j[3]) + 1])
                dl.append(l if rs.random_sample() < 0.85 else rs.randint(1, 7)); ds.append(rs.random_sample())
        for _ in range(rs.randint(0, 6)):                          # false positives
            p = rs.uniform(0, 400, 2); s = np.exp(rs.uniform(np.log(6), np.log(200), 2))
            db.append([p[0], p[1], p[0] + s[0], p[1] + s[1]]); dl.append(rs.randint(1, 7)); ds.append(rs.random_sample() * 0.8)
        dt[img] = dict(boxes=np.asarray(db, dtype=np.float64).reshape(-1, 4), labels=np.asarray(dl, dtype=np.int64),
                       scores=np.round(np.asarray(ds, dtype=np.float64), 3))      # rounded: ties happen
    return gt, dt


def gen_coco_eval(ns, store):
    """The reference's own COCOeval (cocoapi/PythonAPI/pycocotools/{coco,cocoeval}.py, pure Python) with its
    compiled _mask extension replaced by the reference's C bbIou (oracle/_ref/libmaskapi.so)."""
    import importlib
    import types
    import ref_maskapi
    if not hasattr(np, "float"):
        np.float = float            # alias removed in numpy 1.24; cocoeval.py:378-379 still uses it
    pkg_root = os.path.join(ref_harness.REFERENCE_ROOT, "cocoapi", "PythonAPI")
    for k in [k for k in sys.modules if k == "pycocotools" or k.startswith("pycocotools.")]:
        del sys.modules[k]
    sys.path.insert(0, pkg_root)
    fake = types.ModuleType("pycocotools._mask")
    fake.iou = lambda d, g, c: ref_maskapi.bb_iou(d, g, c)
    fake.merge = fake.frPyObjects = fake.encode = fake.decode = fake.area = fake.toBbox = None    # segmentation only
    sys.modules["pycocotools._mask"] = fake
    coco_mod = importlib.import_module("pycocotools.coco")
    eval_mod = importlib.import_module("pycocotools.cocoeval")
    gt, dt = coco_eval_inputs()
    images, anns, res, aid = [], [], [], 1
    for img, g in gt.items():
        images.append({"id": img, "height": 800, "width": 800})
        for b, l, c, a in zip(g["boxes"], g["labels"], g["iscrowd"], g["area"]):
            anns.append({"id": aid, "image_id": img, "category_id": int(l), "bbox": [b[0], b[1], b[2] - b[0], b[3] - b[1]],
                         "area": float(a), "iscrowd": int(c)}); aid += 1
    for img, d in dt.items():
        for b, l, s in zip(d["boxes"], d["labels"], d["scores"]):
            res.append({"image_id": img, "category_id": int(l), "bbox": [b[0], b[1], b[2] - b[0], b[3] - b[1]], "score": float(s)})
    cg = coco_mod.COCO()
    cg.dataset = {"images": images, "annotations": anns, "categories": [{"id": i} for i in range(1, 7)]}
    cg.createIndex()
    cd = cg.loadRes(res)
    ev = eval_mod.COCOeval(cg, cd, "bbox")
    ev.evaluate(); ev.accumulate(); ev.summarize()
    store["coco_stats"] = np.asarray(ev.stats, dtype=np.float64)
    store["coco_precision"] = ev.eval["precision"]
    store["coco_recall"] = ev.eval["recall"]
    rs = np.random.RandomState(5)
    d = np.concatenate([rs.uniform(0, 300, (40, 2)), np.exp(rs.uniform(0, 5.5, (40, 2)))], 1)
    g = np.concatenate([rs.uniform(0, 300, (23, 2)), np.exp(rs.uniform(0, 5.5, (23, 2)))], 1)
    g[:5] = d[:5]
    c = (rs.random_sample(23) < 0.3).astype(np.uint8)
    store["iou_dt"], store["iou_gt"], store["iou_crowd"] = d, g, c
    store["iou_out"] = ref_maskapi.bb_iou(d, g, c)
    store["iou_out_nocrowd"] = ref_maskapi.bb_iou(d, g, None)


PSF_STORE_RUNS = {"w0_of1_n2": (0, 1, 2), "w1_of2_n4": (1, 2, 4)}   # name -> (worker_index, num_workers, total_num_psfs)


def gen_psf_store(ns, meta):
    """Runs the reference's dataset_utils/generate_PSFs.py main() into a temp dir and records the
    SHA-256 of every file it writes (15 directories x slice): the store builder must reproduce the
    files byte for byte."""
    import argparse
    import importlib
    import shutil
    import tempfile
    ref = importlib.import_module("dataset_utils.generate_PSFs")
    out = {}
    for name, (w, nw, tot) in PSF_STORE_RUNS.items():
        d = tempfile.mkdtemp() + "/"
        ref.main(argparse.Namespace(destination_path=d, worker_index=w, num_workers=nw, total_num_psfs=tot))
        digests = {}
        for root, _, files in os.walk(d + "psfs"):
            for fn in files:
                full = os.path.join(root, fn)
                with open(full, "rb") as f:
                    digests[os.path.relpath(full, d)] = hashlib.sha256(f.read()).hexdigest()
        out[name] = digests
        shutil.rmtree(d)
    meta["psf_store"] = out


def main():
    ns = ref_harness.load()
    os.makedirs(OUT, exist_ok=True)
    if "--only-coco" in sys.argv:
        store = {}
        gen_coco_eval(ns, store)
        np.savez_compressed(os.path.join(OUT, "coco.npz"), **store)
        print("coco", store["coco_stats"])
        return
    if "--only-jpeg" in sys.argv:
        store = {}
        gen_jpeg(ns, store)
        np.savez_compressed(os.path.join(OUT, "jpeg.npz"), **store)
        print("jpeg", {k: v.shape for k, v in store.items()})
        return
    if "--only-net-transforms" in sys.argv:
        store = {}
        gen_net_transforms(ns, store)
        np.savez_compressed(os.path.join(OUT, "net_transforms.npz"), **store)
        print("net_transforms", {k: v.shape for k, v in store.items()})
        return
    if "--only-postops" in sys.argv:
        store = {}
        gen_postops(ns, store)
        np.savez_compressed(os.path.join(OUT, "postops.npz"), **store)
        print("postops", len(store), "arrays; block cases changed:", [int(not np.array_equal(store["block_%d" % k], postop_input()[0].numpy())) for k in POSTOP_SEEDS])
        return
    if "--only-warper" in sys.argv:
        store = {}
        gen_warper(ns, store)
        np.savez_compressed(os.path.join(OUT, "warper.npz"), **store)
        print("warper", {k: v.shape for k, v in store.items()})
        return
    if "--only-psf-store" in sys.argv:      # incremental: adds one key to the committed meta.json
        with open(os.path.join(OUT, "meta.json")) as f:
            meta = json.load(f)
        gen_psf_store(ns, meta)
        with open(os.path.join(OUT, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
        print("psf_store:", {k: len(v) for k, v in meta["psf_store"].items()})
        return
    meta = {"numpy": np.__version__, "torch": torch.__version__}
    gen_psf_store(ns, meta)
    for name, fn in (("traj", gen_trajectories), ("psf", gen_psfs), ("boxes", gen_boxes),
                     ("norm", gen_norm), ("fft", gen_fft), ("warper", gen_warper), ("jpeg", gen_jpeg), ("coco", gen_coco_eval),
                     ("net_transforms", gen_net_transforms), ("postops", gen_postops)):
        store = {}
        fn(ns, store)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **store)
        print(name, len(store), "arrays")
    store = {}
    gen_blur(ns, store, meta)
    np.savez_compressed(os.path.join(OUT, "blur.npz"), **store)
    store = {}
    gen_blurdicts(ns, meta, store)
    np.savez_compressed(os.path.join(OUT, "blurdict.npz"), **store)
    with open(os.path.join(OUT, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
