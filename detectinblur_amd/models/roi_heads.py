"""Box head of Faster R-CNN: proposal sampling, multi-scale RoIAlign, two-FC head, per-class box
regression, and test-time post-processing (reference models/faster_rcnn.py:157-159, 204-229 feed these
numbers to torchvision's RoIHeads)."""
import torch
import torch.nn.functional as F
from torch import nn

from . import detector_ops as ops


class TwoMLPHead(nn.Module):
    def __init__(self, in_channels, representation_size):
        super().__init__()
        self.fc6 = nn.Linear(in_channels, representation_size)
        self.fc7 = nn.Linear(representation_size, representation_size)

    def forward(self, x):
        return F.relu(self.fc7(F.relu(self.fc6(x.flatten(start_dim=1)))))


class FastRCNNPredictor(nn.Module):
    def __init__(self, in_channels, num_classes):
        super().__init__()
        self.cls_score = nn.Linear(in_channels, num_classes)
        self.bbox_pred = nn.Linear(in_channels, num_classes * 4)

    def forward(self, x):
        x = x.flatten(start_dim=1)
        return self.cls_score(x), self.bbox_pred(x)


def _unit_boxes(n, like):
    """n x [0, 0, 1, 1] padding boxes, built from fill kernels (a `new_tensor([...])` literal would be a
    blocking host-to-device copy in the middle of the forward pass)."""
    return torch.cat((like.new_zeros((n, 2)), like.new_ones((n, 2))), dim=1)


def fastrcnn_loss(class_logits, box_regression, labels, regression_targets, ok=None):
    """labels [M] int64, regression_targets [M, 4]; `ok` [M] bool masks padding rows of the fixed-size
    sampler (None = every row counts).  Classification: mean cross-entropy over the sampled RoIs;
    boxes: smooth-L1 sum over sampled foreground RoIs / number sampled."""
    if isinstance(labels, (list, tuple)):
        labels, regression_targets = torch.cat(labels), torch.cat(regression_targets)
    M = labels.shape[0]
    if ok is None:
        ok = torch.ones_like(labels, dtype=torch.bool)
    n = ok.sum().clamp(min=1)
    safe = labels.clamp(min=0)
    ce = F.cross_entropy(class_logits, safe, reduction="none")
    cls_loss = torch.where(ok, ce, ce.new_zeros(())).sum() / n
    per_class = box_regression.reshape(M, -1, 4).gather(1, safe[:, None, None].expand(-1, 1, 4)).squeeze(1)
    l1 = F.smooth_l1_loss(per_class, regression_targets, beta=1 / 9, reduction="none")
    box_loss = torch.where((ok & (labels > 0))[:, None], l1, l1.new_zeros(())).sum() / n
    return cls_loss, box_loss


class RoIHeads(nn.Module):
    def __init__(self, box_roi_pool, box_head, box_predictor, fg_iou_thresh, bg_iou_thresh, batch_size_per_image,
                 positive_fraction, bbox_reg_weights, score_thresh, nms_thresh, detections_per_img):
        super().__init__()
        self.box_roi_pool, self.box_head, self.box_predictor = box_roi_pool, box_head, box_predictor
        self.matcher = ops.Matcher(fg_iou_thresh, bg_iou_thresh, allow_low_quality_matches=False)
        self.batch_size_per_image, self.positive_fraction = batch_size_per_image, positive_fraction
        self.box_coder = ops.BoxCoder(bbox_reg_weights or (10.0, 10.0, 5.0, 5.0))
        self.score_thresh, self.nms_thresh, self.detections_per_img = score_thresh, nms_thresh, detections_per_img

    def select_training_samples(self, proposals, targets):
        """proposals: (boxes [N, P, 4], ok [N, P]) from the RPN's padded training output, or a list of
        [P_i, 4] tensors.  Ground truth joins the pool, every candidate is labelled by IoU matching,
        and `batch_size_per_image` RoIs are drawn per image (<= positive_fraction foreground).
        Everything has a fixed shape -- [N, batch_size_per_image] -- so nothing waits on the device;
        `ok` marks the rows that are real samples (all of them unless an image is short of candidates).
        Returns (list of [S, 4] RoIs, labels [N*S], regression targets [N*S, 4], ok [N*S])."""
        if isinstance(proposals, (list, tuple)) and not (len(proposals) == 2 and proposals[1].dtype == torch.bool):
            width = max(p.shape[0] for p in proposals)
            boxes = torch.stack([torch.cat((p, _unit_boxes(width - p.shape[0], p))) for p in proposals])
            ok = torch.stack([torch.arange(width, device=p.device) < p.shape[0] for p in proposals])
        else:
            boxes, ok = proposals
        dtype, device = boxes.dtype, boxes.device
        gt_boxes = [t["boxes"].to(dtype) for t in targets]
        gt_labels = [t["labels"] for t in targets]
        if not self.batched:
            return self._select_per_image(boxes, ok, gt_boxes, gt_labels)
        # every image at once (ground truth padded to the longest list, padding rows [0, 0, 1, 1] as in the per-image form):
        # the per-image loop below is ~30 small launches per image that the host issues slower than the GPU runs them
        N = boxes.shape[0]
        if ops.hip_boxes_ok(boxes, *gt_boxes) and N <= 32 and max(g.shape[0] for g in gt_boxes) <= 256 and ok.dtype == torch.bool:
            return self._select_hip(boxes, ok, gt_boxes, gt_labels)
        gt, valid = ops.pad_boxes(gt_boxes)
        G = gt.shape[1]
        unit = _unit_boxes(1, boxes)[None].expand(N, G, 4)
        pool_gt = torch.where(valid[..., None], gt, unit)
        cands = torch.cat((boxes, pool_gt), dim=1)                                 # ground truth joins the pool
        live = torch.cat((ok, valid), dim=1)
        m = ops.match_batched(self.matcher, ops.box_iou_batched(gt, cands), valid)   # [N, P + G]
        lab_pad = torch.zeros((N, G), dtype=torch.int64, device=device)
        for i, gl in enumerate(gt_labels):
            if gl.shape[0]:
                lab_pad[i, :gl.shape[0]] = gl
        lab = lab_pad.gather(1, m.clamp(min=0))
        lab = torch.where(m == ops.Matcher.BELOW_LOW, lab.new_zeros(()), lab)
        lab = torch.where(m == ops.Matcher.BETWEEN, lab.new_full((), -1), lab)
        labels = torch.where(live, lab, lab.new_full((), -1))                       # padding is ignored by the sampler
        matched = m.clamp(min=0)
        S = self.batch_size_per_image
        pos_idx, pos_ok, neg_idx, neg_ok = ops.sample_pos_neg_fixed(labels, S, self.positive_fraction)
        sel, sel_ok = torch.cat([pos_idx, neg_idx], dim=1), torch.cat([pos_ok, neg_ok], dim=1)
        # real samples first (foreground, then background, each in draw order), padding last; keep S
        order = torch.argsort((~sel_ok).to(torch.int8), dim=1, stable=True)[:, :S]
        sel, sel_ok = sel.gather(1, order), sel_ok.gather(1, order)
        rois = cands.gather(1, sel[..., None].expand(-1, -1, 4))
        labs = labels.gather(1, sel)
        ref = gt.gather(1, matched.gather(1, sel)[..., None].expand(-1, -1, 4))      # images without boxes: zeros
        reg = self.box_coder.encode(ref.reshape(-1, 4), rois.reshape(-1, 4))
        return list(rois), labs.reshape(-1), reg, sel_ok.reshape(-1)

    def _select_hip(self, boxes, ok, gt_boxes, gt_labels):
        """The batched form above with pool, matching, labels and target encoding as one launch each (csrc/dib_detect.hip);
        the sampler in between is the same tensor code, fed the same shapes, so it draws the same rows."""
        P, S = boxes.shape[1], self.batch_size_per_image
        gt_cat, offs = ops.cat_boxes(gt_boxes)
        live_labels = [l for l in gt_labels if l.shape[0]]
        lab_cat = None if not live_labels else (live_labels[0] if len(live_labels) == 1 else torch.cat(live_labels)).to(torch.int64).contiguous()
        g_pad = max(max(g.shape[0] for g in gt_boxes), 1)
        cands = ops.pool_boxes_hip(boxes, gt_cat, offs, g_pad)                       # ground truth joins the pool
        m = ops.match_boxes_hip(self.matcher, gt_cat, offs, cands, shared=False)
        labels = ops.pool_labels_hip(m, lab_cat, offs, ok, P)
        pos_idx, pos_ok, neg_idx, neg_ok = ops.sample_pos_neg_fixed(labels, S, self.positive_fraction)
        sel, sel_ok = torch.cat([pos_idx, neg_idx], dim=1), torch.cat([pos_ok, neg_ok], dim=1)
        order = torch.argsort((~sel_ok).to(torch.int8), dim=1, stable=True)[:, :S]
        sel, sel_ok = sel.gather(1, order), sel_ok.gather(1, order)
        rois = cands.gather(1, sel[..., None].expand(-1, -1, 4))
        labs = labels.gather(1, sel)
        reg = ops.encode_matched_hip(self.box_coder, gt_cat, offs, m.gather(1, sel), rois, shared=False)[0]
        return list(rois), labs.reshape(-1), reg.reshape(-1, 4), sel_ok.reshape(-1)

    batched = True      # False: the per-image form below (the checker of tests/test_detector_ops.py)

    def _select_per_image(self, boxes, ok, gt_boxes, gt_labels):
        dtype, device = boxes.dtype, boxes.device
        g_max = max([g.shape[0] for g in gt_boxes] + [1])
        cands, labels, matched = [], [], []
        for b, o, g, gl in zip(boxes, ok, gt_boxes, gt_labels):
            G = g.shape[0]
            pad = g_max - G
            filler = _unit_boxes(pad, b)
            cand = torch.cat((b, g, filler))                                   # ground truth joins the pool
            live = torch.cat((o, o.new_ones(G), o.new_zeros(pad)))
            if G == 0:
                m = torch.zeros((cand.shape[0],), dtype=torch.int64, device=device)
                lab = torch.zeros((cand.shape[0],), dtype=torch.int64, device=device)
            else:
                m = self.matcher(ops.box_iou(g, cand))
                lab = gl[m.clamp(min=0)].to(torch.int64)
                lab = torch.where(m == ops.Matcher.BELOW_LOW, lab.new_zeros(()), lab)
                lab = torch.where(m == ops.Matcher.BETWEEN, lab.new_full((), -1), lab)
            cands.append(cand)
            labels.append(torch.where(live, lab, lab.new_full((), -1)))         # padding is ignored by the sampler
            matched.append(m.clamp(min=0))
        cands, labels, matched = torch.stack(cands), torch.stack(labels), torch.stack(matched)
        S = self.batch_size_per_image
        pos_idx, pos_ok, neg_idx, neg_ok = ops.sample_pos_neg_fixed(labels, S, self.positive_fraction)
        sel, sel_ok = torch.cat([pos_idx, neg_idx], dim=1), torch.cat([pos_ok, neg_ok], dim=1)
        # real samples first (foreground, then background, each in draw order), padding last; keep S
        order = torch.argsort((~sel_ok).to(torch.int8), dim=1, stable=True)[:, :S]
        sel, sel_ok = sel.gather(1, order), sel_ok.gather(1, order)
        rois = cands.gather(1, sel[..., None].expand(-1, -1, 4))
        labs = labels.gather(1, sel)
        midx = matched.gather(1, sel)
        reg = []
        for r, mi, g in zip(rois, midx, gt_boxes):
            g = g if g.numel() else torch.zeros((1, 4), dtype=dtype, device=device)
            reg.append(self.box_coder.encode(g[mi], r))
        return list(rois), labs.reshape(-1), torch.cat(reg), sel_ok.reshape(-1)

    def postprocess_detections(self, class_logits, box_regression, proposals, image_shapes):
        num_classes = class_logits.shape[-1]
        counts = [p.shape[0] for p in proposals]
        if (ops.hip_boxes_ok(class_logits, box_regression, *proposals) and num_classes <= 128 and max(counts + [0]) <= 2048
                and self.detections_per_img <= 2048 and (num_classes - 1) * self.detections_per_img <= ops.TOPK_SPLIT):
            out, start = [], 0
            for p, shape in zip(proposals, image_shapes):
                R = p.shape[0]
                out.append(self._detections_hip(class_logits[start:start + R], box_regression[start:start + R], p, shape))
                start += R
            return out
        boxes = self.box_coder.decode(box_regression, torch.cat(proposals)).split(counts, 0)
        scores = F.softmax(class_logits, -1).split(counts, 0)
        out = []
        for b, s, shape in zip(boxes, scores, image_shapes):
            b = ops.clip_boxes_to_image(b, shape)
            labels = torch.arange(num_classes, device=b.device).view(1, -1).expand_as(s)
            b, s, labels = b[:, 1:].reshape(-1, 4), s[:, 1:].reshape(-1), labels[:, 1:].reshape(-1)   # drop background
            keep = torch.where(s > self.score_thresh)[0]
            b, s, labels = b[keep], s[keep], labels[keep]
            keep = ops.remove_small_boxes(b, 1e-2)
            b, s, labels = b[keep], s[keep], labels[keep]
            keep = ops.batched_nms(b, s, labels, self.nms_thresh)[:self.detections_per_img]
            out.append({"boxes": b[keep], "labels": labels[keep], "scores": s[keep]})
        return out

    _class_ids = {}

    def _detections_hip(self, class_logits, box_regression, rois, shape, defer=False):
        """postprocess_detections of one image in ~20 launches and one host synchronisation (the final count): candidates
        class-major from one kernel, every class's candidates sorted by one top-k launch, ONE batched NMS over the classes as
        independent sets -- on boxes moved apart by class * (largest coordinate + 1), the arithmetic of torchvision's batched_nms, so
        the same pairs suppress each other --, and the classes' survivors (at most detections_per_img of each can matter) ranked by
        a last top-k.  Equal scores come out in (class, rank) order; the tensor form below leaves their order to an unstable sort."""
        R, C = class_logits.shape
        dev, D = class_logits.device, self.detections_per_img
        if R == 0:
            return {"boxes": class_logits.new_zeros((0, 4)), "labels": torch.zeros((0,), dtype=torch.int64, device=dev), "scores": class_logits.new_zeros((0,))}
        scores_cm, boxes_cm, stats = ops.det_candidates_hip(self.box_coder, class_logits, box_regression, rois, shape, self.score_thresh, 1e-2)
        s, _, b, v = ops.topk_levels_hip(scores_cm, [R], [R], R, boxes_cm, None, float("-inf"))          # [C - 1, 1, R]: per class, by score
        s, b, v = s[:, 0], b[:, 0], v[:, 0]
        key = (str(dev), C)
        cls = RoIHeads._class_ids.get(key)
        if cls is None:
            cls = RoIHeads._class_ids[key] = torch.arange(1, C, device=dev)
        offsets = cls.to(torch.float32) * (stats[1:2].view(torch.float32) + 1)                             # batched_nms: idxs * (boxes.max() + 1)
        keep, count = ops.nms_sets_sorted(b + offsets[:, None, None], v, self.nms_thresh)
        d = min(D, R)
        keep = keep[:, :d]
        kept = torch.arange(d, device=dev)[None, :] < count[:, None]
        fs = torch.where(kept, s.gather(1, keep), s.new_full((), float("-inf")))
        fb = b.gather(1, keep[..., None].expand(-1, -1, 4))
        k = min(D, (C - 1) * d)
        top_s, top_i, top_b, _ = ops.topk_levels_hip(fs.reshape(1, -1), [(C - 1) * d], [k], k, fb.reshape(1, -1, 4), want_index=True)
        if defer:          # padded to k rows, the number that count on the device: the caller synchronises when it suits it
            return top_b[0, 0], top_s[0, 0], top_i[0, 0] // d + 1, count.sum().clamp(max=D)
        n = int(count.sum().clamp(max=D))                                                                  # the one synchronisation
        return {"boxes": top_b[0, 0, :n], "labels": top_i[0, 0, :n] // d + 1, "scores": top_s[0, 0, :n]}

    def padded_detections(self, features, proposals, image_shapes):
        """Inference without any host synchronisation (engine.evaluate's pipelined loop): per image (boxes [k, 4], scores [k],
        labels [k], number of real rows as a 0-d device tensor) and None -- or (None, (class_logits, box_regression)) where the
        kernels of `_detections_hip` do not apply and the caller has to finish through `postprocess_detections`."""
        box_features = self.box_head(self.box_roi_pool(features, proposals, image_shapes))
        class_logits, box_regression = self.box_predictor(box_features)
        C = class_logits.shape[-1]
        counts = [p.shape[0] for p in proposals]
        if not (ops.hip_boxes_ok(class_logits, box_regression, *proposals) and C <= 128 and max(counts + [0]) <= 2048
                and self.detections_per_img <= 2048 and (C - 1) * self.detections_per_img <= ops.TOPK_SPLIT):
            return None, (class_logits, box_regression)
        out, start = [], 0
        for p, shape in zip(proposals, image_shapes):
            R = p.shape[0]
            if R == 0:                                                   # an image without a single proposal: one padding row, none real
                out.append((class_logits.new_zeros((1, 4)), class_logits.new_zeros((1,)), torch.zeros((1,), dtype=torch.int64, device=p.device),
                            torch.zeros((), dtype=torch.int64, device=p.device)))
                continue
            out.append(self._detections_hip(class_logits[start:start + R], box_regression[start:start + R], p, shape, defer=True))
            start += R
        return out, None

    def forward(self, features, proposals, image_shapes, targets=None):
        if self.training:
            proposals, labels, reg_targets, ok = self.select_training_samples(proposals, targets)
        box_features = self.box_head(self.box_roi_pool(features, proposals, image_shapes))
        class_logits, box_regression = self.box_predictor(box_features)
        if self.training:
            cls, box = fastrcnn_loss(class_logits, box_regression, labels, reg_targets, ok)
            return [], {"loss_classifier": cls, "loss_box_reg": box}
        return self.postprocess_detections(class_logits, box_regression, proposals, image_shapes), {}
