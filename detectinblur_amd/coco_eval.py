"""COCO bounding-box evaluation (AP / AR, the 12 numbers of COCOeval.summarize) without pycocotools --
SURVEY.md section 8f-4.  The protocol is pycocotools' (reference cocoapi/PythonAPI/pycocotools/cocoeval.py:
computeIoU :158-187, evaluateImg :225-304, accumulate :306-403, summarize :405-470); the detection x
ground-truth IoU, its inner loop (cocoapi/common/maskApi.c:109-120), runs on the GPU through
`dib_coco_box_iou` when a CUDA device is given, one launch per image for all categories at once.

    ev = CocoBoxEvaluator(ground_truth, device="cuda")      # {image_id: {"boxes" xyxy, "labels", "iscrowd"?, "area"?}}
    ev.update({image_id: {"boxes" xyxy, "labels", "scores"}})   # as engine.evaluate returns them
    stats = ev.summarize()                                  # AP, AP50, AP75, APs, APm, APl, AR1, AR10, AR100, ARs, ARm, ARl
"""
import numpy as np
import torch

IOU_THRS = np.linspace(0.5, 0.95, int(np.round((0.95 - 0.5) / 0.05)) + 1, endpoint=True)
REC_THRS = np.linspace(0.0, 1.00, int(np.round((1.00 - 0.0) / 0.01)) + 1, endpoint=True)
MAX_DETS = [1, 10, 100]
AREA_RNG = [[0 ** 2, 1e5 ** 2], [0 ** 2, 32 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]


def _xywh(boxes):
    b = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
    return np.stack([b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], axis=1)


def _iou_cpu(dt, gt, crowd):
    """bbIou restated with numpy float64 (used when no GPU is given; same expression order)."""
    out = np.zeros((dt.shape[0], gt.shape[0]), dtype=np.float64)
    for g in range(gt.shape[0]):
        G = gt[g]
        ga = G[2] * G[3]
        w = np.minimum(dt[:, 2] + dt[:, 0], G[2] + G[0]) - np.maximum(dt[:, 0], G[0])
        h = np.minimum(dt[:, 3] + dt[:, 1], G[3] + G[1]) - np.maximum(dt[:, 1], G[1])
        ok = (w > 0) & (h > 0)
        i = w * h
        da = dt[:, 2] * dt[:, 3]
        u = da if crowd[g] else da + ga - i
        out[ok, g] = i[ok] / u[ok]
    return out


class _CatRecord(object):
    """Matching results of one (image, category): scores [D], dtm / dt_ig [A, T, D] bool, n_gt [A].  Indexing by area range gives
    the per-range dict the interpreted matcher returns (`rec[a]["dtm"]` is [T, D])."""
    __slots__ = ("scores", "dtm", "dt_ig", "n_gt")

    def __init__(self, scores, dtm, dt_ig, n_gt):
        self.scores, self.dtm, self.dt_ig, self.n_gt = scores, dtm, dt_ig, n_gt

    @classmethod
    def of(cls, x):
        if isinstance(x, cls):
            return x
        return cls(x[0]["scores"], np.stack([r["dtm"] for r in x]), np.stack([r["dt_ig"] for r in x]), np.asarray([r["n_gt"] for r in x]))

    def __len__(self):
        return len(self.n_gt)

    def __getitem__(self, a):
        if not 0 <= a < len(self.n_gt):
            raise IndexError(a)
        return dict(scores=self.scores, dtm=self.dtm[a], dt_ig=self.dt_ig[a], n_gt=int(self.n_gt[a]))

    def __getstate__(self):
        return (self.scores, np.ascontiguousarray(self.dtm), np.ascontiguousarray(self.dt_ig), np.ascontiguousarray(self.n_gt))

    def __setstate__(self, st):
        self.scores, self.dtm, self.dt_ig, self.n_gt = st


class CocoBoxEvaluator(object):
    def __init__(self, ground_truth, device=None, cats=None):
        self.device = torch.device(device) if device is not None else None
        self.gt = {}
        for img, t in ground_truth.items():
            boxes = _xywh(torch.as_tensor(t["boxes"]).cpu().numpy())
            n = boxes.shape[0]
            crowd = np.asarray(torch.as_tensor(t.get("iscrowd", np.zeros(n))).cpu().numpy(), dtype=np.int64).reshape(-1)
            area = (np.asarray(torch.as_tensor(t["area"]).cpu().numpy(), dtype=np.float64).reshape(-1)
                    if "area" in t else boxes[:, 2] * boxes[:, 3])
            self.gt[img] = dict(boxes=boxes, labels=np.asarray(torch.as_tensor(t["labels"]).cpu().numpy()).reshape(-1),
                                crowd=crowd, area=area)
        self.cats = sorted(cats) if cats is not None else sorted({int(c) for g in self.gt.values() for c in g["labels"]})
        self.results = {}      # (image, category) -> per-area-range records
        self.images = []

    def set_ground_truth_xywh(self, img, boxes_xywh, labels, crowd, area):
        """Ground truth of one image in COCO's own layout (bbox = x, y, w, h), installed or replaced just
        before the image is evaluated: `engine.evaluate` edits the boxes of an image (expanded targets,
        reference engine.py:325-342) before it scores it."""
        self.gt[img] = dict(boxes=np.asarray(boxes_xywh, dtype=np.float64).reshape(-1, 4),
                            labels=np.asarray(labels, dtype=np.int64).reshape(-1),
                            crowd=np.asarray(crowd, dtype=np.int64).reshape(-1),
                            area=np.asarray(area, dtype=np.float64).reshape(-1))

    # ---- per image ------------------------------------------------------------------------------------
    def _iou(self, dt, gt, crowd):
        if dt.shape[0] == 0 or gt.shape[0] == 0:
            return np.zeros((dt.shape[0], gt.shape[0]))
        if self.device is not None and self.device.type == "cuda":
            from .models.detector_ops import coco_box_iou
            return coco_box_iou(torch.from_numpy(dt).to(self.device), torch.from_numpy(gt).to(self.device),
                                torch.from_numpy(crowd.astype(np.uint8)).to(self.device)).cpu().numpy()
        return _iou_cpu(dt, gt, crowd)

    def update(self, detections):
        for img, d in detections.items():
            g = self.gt[img]
            boxes = _xywh(torch.as_tensor(d["boxes"]).cpu().numpy())
            scores = np.asarray(torch.as_tensor(d["scores"]).cpu().numpy(), dtype=np.float64).reshape(-1)
            labels = np.asarray(torch.as_tensor(d["labels"]).cpu().numpy()).reshape(-1)
            iou_all = self._iou(boxes, g["boxes"], g["crowd"])            # every detection x every ground truth
            if img not in self._seen():
                self.images.append(img)
                self._seen_set.add(img)
            self._match_image(img, boxes, scores, labels, g, iou_all)

    def _match_image(self, img, boxes, scores, labels, g, iou_all):
        """Every category of one image in one native call (csrc/host/dib_host.c: dib_coco_match_image), outside the interpreter
        lock; `_match_image_py` is the same thing category by category in Python (the checker)."""
        import ctypes
        from . import _hostlib
        h = _hostlib
        D, G, K, A, T = len(scores), len(g["labels"]), len(self.cats), len(AREA_RNG), len(IOU_THRS)
        if not hasattr(self, "_cats_arr"):
            self._cats_arr = np.ascontiguousarray(self.cats, dtype=np.int64)
        io = np.ascontiguousarray(iou_all, dtype=np.float64)
        lab = np.ascontiguousarray(labels, dtype=np.int64)
        sc = np.ascontiguousarray(scores, dtype=np.float64)
        da = np.ascontiguousarray(boxes[:, 2] * boxes[:, 3], dtype=np.float64)
        gl, gc, ga = (np.ascontiguousarray(g["labels"], dtype=np.int64), np.ascontiguousarray(g["crowd"], dtype=np.int64),
                      np.ascontiguousarray(g["area"], dtype=np.float64))
        order = np.zeros(max(D, 1), dtype=np.int32)
        start = np.zeros(K + 1, dtype=np.int32)
        dtm = np.zeros((A, T, max(D, 1)), dtype=np.uint8)
        dt_ig = np.zeros((A, T, max(D, 1)), dtype=np.uint8)
        n_gt = np.zeros((max(K, 1), A), dtype=np.int32)
        n_cat_gt = np.zeros(max(K, 1), dtype=np.int32)
        ll = lambda a: a.ctypes.data_as(h._llp)
        rc = h.lib().dib_coco_match_image(h.dptr(io), D, G, ll(lab), h.dptr(sc), h.dptr(da), ll(gl), ll(gc), h.dptr(ga), ll(self._cats_arr), K,
                                          MAX_DETS[-1], h.dptr(CocoBoxEvaluator._AREA), A, h.dptr(IOU_THRS), T, order.ctypes.data_as(h._ip),
                                          start.ctypes.data_as(h._ip), dtm.ctypes.data_as(h._u8p), dt_ig.ctypes.data_as(h._u8p),
                                          n_gt.ctypes.data_as(h._ip), n_cat_gt.ctypes.data_as(h._ip))
        if rc != 0:
            raise RuntimeError("dib_coco_match_image failed (%d)" % rc)
        dtm, dt_ig = dtm.view(bool), dt_ig.view(bool)
        st = start.tolist()
        sk = sc[order[:st[K]]]                                   # every category's scores in its own order, one gather
        cats, res = self.cats, self.results
        for k in np.nonzero((start[1:] > start[:-1]) | (n_cat_gt[:K] > 0))[0].tolist():       # plain ints and basic slices from here
            s, e = st[k], st[k + 1]
            res[(img, cats[k])] = _CatRecord(sk[s:e], dtm[:, :, s:e], dt_ig[:, :, s:e], n_gt[k])

    def _match_image_py(self, img, boxes, scores, labels, g, iou_all):
            for cat in self.cats:
                di = np.nonzero(labels == cat)[0]
                gi = np.nonzero(g["labels"] == cat)[0]
                if di.size == 0 and gi.size == 0:
                    continue
                di = di[np.argsort(-scores[di], kind="mergesort")][:MAX_DETS[-1]]
                self.results[(img, cat)] = self._match_py(iou_all[np.ix_(di, gi)], scores[di], boxes[di, 2] * boxes[di, 3],
                                                          g["crowd"][gi], g["area"][gi])

    def _seen(self):
        if not hasattr(self, "_seen_set") or len(self._seen_set) != len(self.images):
            self._seen_set = set(self.images)
        return self._seen_set

    _AREA = np.ascontiguousarray(np.asarray(AREA_RNG, dtype=np.float64))

    @staticmethod
    def _match(ious, scores, dt_area, crowd, gt_area):
        """evaluateImg for the four area ranges: greedy matching per IoU threshold, detections by score.  The loop nest runs in
        the native host library (csrc/host/dib_host.c: dib_coco_match -- the same statements; `_match_py` below is the checker,
        tests/test_coco_eval.py), outside the interpreter lock: engine.evaluate scores on a worker thread while the main thread
        launches the next image."""
        import ctypes
        from . import _hostlib
        T, D, G, A = len(IOU_THRS), len(scores), len(crowd), len(AREA_RNG)
        io = np.ascontiguousarray(ious, dtype=np.float64).reshape(D, G) if D and G else np.zeros((D, G))
        da = np.ascontiguousarray(dt_area, dtype=np.float64)
        cr = np.ascontiguousarray(crowd, dtype=np.int64)
        ga = np.ascontiguousarray(gt_area, dtype=np.float64)
        dtm = np.zeros((A, T, D), dtype=np.uint8)
        dt_ig = np.zeros((A, T, D), dtype=np.uint8)
        n_gt = np.zeros(A, dtype=np.int32)
        u8 = ctypes.POINTER(ctypes.c_ubyte)
        rc = _hostlib.lib().dib_coco_match(_hostlib.dptr(io), D, G, _hostlib.dptr(da), cr.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)),
                                           _hostlib.dptr(ga), _hostlib.dptr(CocoBoxEvaluator._AREA), A, _hostlib.dptr(IOU_THRS), T,
                                           dtm.ctypes.data_as(u8), dt_ig.ctypes.data_as(u8), n_gt.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
        if rc != 0:
            raise RuntimeError("dib_coco_match failed (%d)" % rc)
        return [dict(scores=scores, dtm=dtm[a].astype(bool), dt_ig=dt_ig[a].astype(bool), n_gt=int(n_gt[a])) for a in range(A)]

    @staticmethod
    def _match_py(ious, scores, dt_area, crowd, gt_area):
        """`_match` as the interpreted loop nest (the reference's own form): the checker of the native routine."""
        out = []
        T, D = len(IOU_THRS), len(scores)
        for lo, hi in AREA_RNG:
            ignore = (crowd != 0) | (gt_area < lo) | (gt_area > hi)
            order = np.argsort(ignore.astype(np.uint8), kind="mergesort")       # counted ground truth first
            ig, cr = ignore[order], crowd[order]
            io = ious[:, order] if ious.size else ious
            G = len(order)
            gtm = np.zeros((T, G), dtype=bool)
            dtm = np.zeros((T, D), dtype=bool)
            dt_ig = np.zeros((T, D), dtype=bool)
            for t, thr in enumerate(IOU_THRS):
                for d in range(D):
                    best, m = min(thr, 1 - 1e-10), -1
                    for k in range(G):
                        if gtm[t, k] and not cr[k]:
                            continue
                        if m > -1 and not ig[m] and ig[k]:
                            break
                        if io[d, k] < best:
                            continue
                        best, m = io[d, k], k
                    if m == -1:
                        continue
                    dt_ig[t, d] = ig[m]
                    dtm[t, d] = True
                    gtm[t, m] = True
            outside = (dt_area < lo) | (dt_area > hi)
            dt_ig = dt_ig | (~dtm & outside[None, :])
            out.append(dict(scores=scores, dtm=dtm, dt_ig=dt_ig, n_gt=int((~ig).sum())))
        return out

    # ---- accumulate + summarize ---------------------------------------------------------------------------
    def accumulate(self):
        """COCOeval.accumulate (reference cocoapi/PythonAPI/pycocotools/cocoeval.py:315-420).  Per category the records of all
        images are laid end to end once and the ranking, the running counts and the 101-point precision curves for every
        (maxDets, area range, IoU threshold) come from one native call (csrc/host/dib_host.c: dib_coco_accumulate_cat);
        `accumulate_py` is the interpreted form, the checker."""
        from . import _hostlib as h
        T, R, K, A, M = len(IOU_THRS), len(REC_THRS), len(self.cats), len(AREA_RNG), len(MAX_DETS)
        precision = -np.ones((T, R, K, A, M))
        recall = -np.ones((T, K, A, M))
        max_dets = np.asarray(MAX_DETS, dtype=np.int32)
        rec_thrs = np.ascontiguousarray(REC_THRS, dtype=np.float64)
        for k, cat in enumerate(self.cats):
            recs = [_CatRecord.of(self.results[(img, cat)]) for img in self.images if (img, cat) in self.results]
            if not recs:
                continue
            lens = np.asarray([len(r.scores) for r in recs], dtype=np.int32)
            scores = np.ascontiguousarray(np.concatenate([r.scores for r in recs]), dtype=np.float64)
            dtm = np.ascontiguousarray(np.concatenate([r.dtm for r in recs], axis=2)).view(np.uint8)
            dt_ig = np.ascontiguousarray(np.concatenate([r.dt_ig for r in recs], axis=2)).view(np.uint8)
            n_gt = np.ascontiguousarray(np.sum([r.n_gt for r in recs], axis=0), dtype=np.int32)
            p_cat, r_cat = -np.ones((A, M, T, R)), -np.ones((A, M, T))
            rc = h.lib().dib_coco_accumulate_cat(h.dptr(scores), lens.ctypes.data_as(h._ip), len(recs), dtm.ctypes.data_as(h._u8p),
                                                 dt_ig.ctypes.data_as(h._u8p), n_gt.ctypes.data_as(h._ip), A, T, max_dets.ctypes.data_as(h._ip), M,
                                                 h.dptr(rec_thrs), R, h.dptr(p_cat), h.dptr(r_cat))
            if rc != 0:
                raise RuntimeError("dib_coco_accumulate_cat failed (%d)" % rc)
            precision[:, :, k, :, :] = p_cat.transpose(2, 3, 0, 1)
            recall[:, k, :, :] = r_cat.transpose(2, 0, 1)
        self.precision, self.recall = precision, recall
        return precision, recall

    def accumulate_py(self):
        T, R, K, A, M = len(IOU_THRS), len(REC_THRS), len(self.cats), len(AREA_RNG), len(MAX_DETS)
        precision = -np.ones((T, R, K, A, M))
        recall = -np.ones((T, K, A, M))
        for k, cat in enumerate(self.cats):
            for a in range(A):
                recs = [_CatRecord.of(self.results[(img, cat)]) for img in self.images if (img, cat) in self.results]
                if not recs:
                    continue
                for m, max_det in enumerate(MAX_DETS):
                    scores = np.concatenate([r.scores[:max_det] for r in recs])
                    inds = np.argsort(-scores, kind="mergesort")
                    dtm = np.concatenate([r.dtm[a][:, :max_det] for r in recs], axis=1)[:, inds]
                    dt_ig = np.concatenate([r.dt_ig[a][:, :max_det] for r in recs], axis=1)[:, inds]
                    npig = sum(int(r.n_gt[a]) for r in recs)
                    if npig == 0:
                        continue
                    tps = np.cumsum(dtm & ~dt_ig, axis=1).astype(np.float64)
                    fps = np.cumsum(~dtm & ~dt_ig, axis=1).astype(np.float64)
                    for t in range(T):
                        tp, fp = tps[t], fps[t]
                        nd = len(tp)
                        rc = tp / npig
                        pr = tp / (fp + tp + np.spacing(1))
                        recall[t, k, a, m] = rc[-1] if nd else 0
                        pr = pr.tolist()
                        for i in range(nd - 1, 0, -1):          # precision envelope
                            if pr[i] > pr[i - 1]:
                                pr[i - 1] = pr[i]
                        q = np.zeros((R,))
                        pos = np.searchsorted(rc, REC_THRS, side="left")
                        for ri, pi in enumerate(pos):
                            if pi >= nd:
                                break
                            q[ri] = pr[pi]
                        precision[t, :, k, a, m] = q
        self.precision, self.recall = precision, recall
        return precision, recall

    def summarize(self):
        if not hasattr(self, "precision"):
            self.accumulate()

        def pick(ap, iou=None, area=0, max_det=2):
            s = self.precision if ap else self.recall
            if iou is not None:
                s = s[np.where(iou == IOU_THRS)[0]]
            s = s[..., area, max_det]
            return -1.0 if len(s[s > -1]) == 0 else float(np.mean(s[s > -1]))

        self.stats = np.array([pick(1), pick(1, 0.5), pick(1, 0.75), pick(1, area=1), pick(1, area=2), pick(1, area=3),
                               pick(0, max_det=0), pick(0, max_det=1), pick(0), pick(0, area=1), pick(0, area=2), pick(0, area=3)])
        return self.stats


_STAT_NAMES = ["AP @[ IoU=0.50:0.95 | area=   all | maxDets=100 ]", "AP @[ IoU=0.50      | area=   all | maxDets=100 ]",
               "AP @[ IoU=0.75      | area=   all | maxDets=100 ]", "AP @[ IoU=0.50:0.95 | area= small | maxDets=100 ]",
               "AP @[ IoU=0.50:0.95 | area=medium | maxDets=100 ]", "AP @[ IoU=0.50:0.95 | area= large | maxDets=100 ]",
               "AR @[ IoU=0.50:0.95 | area=   all | maxDets=  1 ]", "AR @[ IoU=0.50:0.95 | area=   all | maxDets= 10 ]",
               "AR @[ IoU=0.50:0.95 | area=   all | maxDets=100 ]", "AR @[ IoU=0.50:0.95 | area= small | maxDets=100 ]",
               "AR @[ IoU=0.50:0.95 | area=medium | maxDets=100 ]", "AR @[ IoU=0.50:0.95 | area= large | maxDets=100 ]"]


class CocoEvaluator(object):
    """The reference's `coco_eval.CocoEvaluator` (coco_eval.py:20-77) for the box task, same surface:
    `coco_gt` (a deep copy; `coco_gt.imgToAnns[image_id][k]["bbox"]` may be edited before the image is
    scored), `update({image_id: {"boxes", "labels", "scores"}})`, `synchronize_between_processes()`,
    `accumulate()`, `summarize()`, and `coco_eval["bbox"].stats` -- the 12 numbers reference train.py:350-387
    and evaluate.py:249-259 log.  Underneath: CocoBoxEvaluator (IoU on the GPU when `device` is one)."""

    def __init__(self, coco_gt, iou_types=("bbox",), device=None):
        import copy
        assert isinstance(iou_types, (list, tuple))
        other = [t for t in iou_types if t != "bbox"]
        if other:
            raise NotImplementedError("iou types %s are outside the built path (Faster R-CNN: boxes only)" % other)
        self.coco_gt = copy.deepcopy(coco_gt)
        self.iou_types = list(iou_types)
        cats = self.coco_gt.getCatIds() or sorted({a["category_id"] for a in self.coco_gt.dataset.get("annotations", [])})
        self._ev = CocoBoxEvaluator({}, device=device, cats=cats)
        self.coco_eval = {"bbox": self._ev}
        self.img_ids = []

    def update(self, predictions):
        img_ids = [int(i) for i in np.unique(list(predictions.keys()))]
        self.img_ids.extend(img_ids)
        for img in img_ids:
            anns = self.coco_gt.imgToAnns.get(img, [])
            self._ev.set_ground_truth_xywh(img, [a["bbox"] for a in anns], [a["category_id"] for a in anns],
                                           [a.get("ignore", 0) or a["iscrowd"] for a in anns], [a["area"] for a in anns])
        self._ev.update({int(k): v for k, v in predictions.items()})

    def synchronize_between_processes(self):
        """Every rank ends up with every rank's per-image records, images unique and ascending by id (the
        reference's np.unique merge, coco_eval.py:283-303).  The collective is entered unconditionally."""
        from . import utils
        merged = {}
        for part in utils.all_gather(self._ev.results):
            merged.update(part)
        self._ev.results = merged
        self._ev.images = sorted({img for img, _ in merged} | set(self._ev.images))
        self.img_ids = list(self._ev.images)

    def accumulate(self):
        self._ev.accumulate()

    def summarize(self):
        print("IoU metric: bbox")
        stats = self._ev.summarize()
        for name, v in zip(_STAT_NAMES, stats):
            print(" Average %s (%s) %s = %0.3f" % ("Precision" if name.startswith("AP") else "Recall   ", name[:2], name[3:], v))
        return stats
