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


class CocoBoxEvaluator(object):
    def __init__(self, ground_truth, device=None):
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
        self.cats = sorted({int(c) for g in self.gt.values() for c in g["labels"]})
        self.results = {}      # (image, category) -> per-area-range records
        self.images = []

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
            self.images.append(img)
            for cat in self.cats:
                di = np.nonzero(labels == cat)[0]
                gi = np.nonzero(g["labels"] == cat)[0]
                if di.size == 0 and gi.size == 0:
                    continue
                di = di[np.argsort(-scores[di], kind="mergesort")][:MAX_DETS[-1]]
                self.results[(img, cat)] = self._match(iou_all[np.ix_(di, gi)], scores[di], boxes[di, 2] * boxes[di, 3],
                                                       g["crowd"][gi], g["area"][gi])

    @staticmethod
    def _match(ious, scores, dt_area, crowd, gt_area):
        """evaluateImg for the four area ranges: greedy matching per IoU threshold, detections by score."""
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
        T, R, K, A, M = len(IOU_THRS), len(REC_THRS), len(self.cats), len(AREA_RNG), len(MAX_DETS)
        precision = -np.ones((T, R, K, A, M))
        recall = -np.ones((T, K, A, M))
        for k, cat in enumerate(self.cats):
            for a in range(A):
                recs = [self.results[(img, cat)][a] for img in self.images if (img, cat) in self.results]
                if not recs:
                    continue
                for m, max_det in enumerate(MAX_DETS):
                    scores = np.concatenate([r["scores"][:max_det] for r in recs])
                    inds = np.argsort(-scores, kind="mergesort")
                    dtm = np.concatenate([r["dtm"][:, :max_det] for r in recs], axis=1)[:, inds]
                    dt_ig = np.concatenate([r["dt_ig"][:, :max_det] for r in recs], axis=1)[:, inds]
                    npig = sum(r["n_gt"] for r in recs)
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
