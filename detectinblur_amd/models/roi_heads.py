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


def fastrcnn_loss(class_logits, box_regression, labels, regression_targets):
    labels = torch.cat(labels)
    regression_targets = torch.cat(regression_targets)
    cls_loss = F.cross_entropy(class_logits, labels)
    pos = torch.where(labels > 0)[0]
    box_regression = box_regression.reshape(class_logits.shape[0], -1, 4)
    box_loss = F.smooth_l1_loss(box_regression[pos, labels[pos]], regression_targets[pos], beta=1 / 9, reduction="sum") / max(labels.numel(), 1)
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
        gt_boxes = [t["boxes"].to(proposals[0].dtype) for t in targets]
        gt_labels = [t["labels"] for t in targets]
        proposals = [torch.cat((p, g)) for p, g in zip(proposals, gt_boxes)]      # ground truth joins the pool
        labels, matched_idx = [], []
        for p, g, gl in zip(proposals, gt_boxes, gt_labels):
            if g.numel() == 0:
                matched_idx.append(torch.zeros((p.shape[0],), dtype=torch.int64, device=p.device))
                labels.append(torch.zeros((p.shape[0],), dtype=torch.int64, device=p.device))
                continue
            m = self.matcher(ops.box_iou(g, p))
            lab = gl[m.clamp(min=0)].to(torch.int64)
            lab[m == ops.Matcher.BELOW_LOW] = 0
            lab[m == ops.Matcher.BETWEEN] = -1
            matched_idx.append(m.clamp(min=0))
            labels.append(lab)
        picks = ops.sample_pos_neg(labels, self.batch_size_per_image, self.positive_fraction)
        out_p, out_l, out_t = [], [], []
        for (pos, neg), p, lab, mi, g in zip(picks, proposals, labels, matched_idx, gt_boxes):
            idx = torch.cat([pos, neg])
            p, lab = p[idx], lab[idx]
            g = g if g.numel() else torch.zeros((1, 4), dtype=p.dtype, device=p.device)
            out_p.append(p)
            out_l.append(lab)
            out_t.append(self.box_coder.encode(g[mi[idx]], p))
        return out_p, out_l, out_t

    def postprocess_detections(self, class_logits, box_regression, proposals, image_shapes):
        num_classes = class_logits.shape[-1]
        counts = [p.shape[0] for p in proposals]
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

    def forward(self, features, proposals, image_shapes, targets=None):
        if self.training:
            proposals, labels, reg_targets = self.select_training_samples(proposals, targets)
        box_features = self.box_head(self.box_roi_pool(features, proposals, image_shapes))
        class_logits, box_regression = self.box_predictor(box_features)
        if self.training:
            cls, box = fastrcnn_loss(class_logits, box_regression, labels, reg_targets)
            return [], {"loss_classifier": cls, "loss_box_reg": box}
        return self.postprocess_detections(class_logits, box_regression, proposals, image_shapes), {}
