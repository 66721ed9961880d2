"""Region proposal network of Faster R-CNN on FPN levels (Ren et al. 2015; Lin et al. 2017), with the
hyper-parameters the reference passes to torchvision's RegionProposalNetwork
(reference models/faster_rcnn.py:150-154, 185-203)."""
import torch
import torch.nn.functional as F
from torch import nn

from . import detector_ops as ops
from .backbone import bias_act, conv1x1

FUSE_HEAD = True     # RPN head: fused bias + ReLU epilogue, one convolution for both predictors (False: the plain module graph)


class AnchorGenerator(nn.Module):
    def __init__(self, sizes=((32,), (64,), (128,), (256,), (512,)), aspect_ratios=((0.5, 1.0, 2.0),) * 5):
        super().__init__()
        self.sizes, self.aspect_ratios = sizes, aspect_ratios

    def num_anchors_per_location(self):
        return [len(s) * len(a) for s, a in zip(self.sizes, self.aspect_ratios)]

    @staticmethod
    def _base(scales, ratios, dtype, device):
        scales = torch.as_tensor(scales, dtype=dtype, device=device)
        ratios = torch.as_tensor(ratios, dtype=dtype, device=device)
        h_r = torch.sqrt(ratios)
        w_r = 1 / h_r
        ws = (w_r[:, None] * scales[None, :]).view(-1)
        hs = (h_r[:, None] * scales[None, :]).view(-1)
        return (torch.stack([-ws, -hs, ws, hs], dim=1) / 2).round()

    def forward(self, image_list, feature_maps):
        img_h, img_w = image_list.tensors.shape[-2:]
        dtype, device = feature_maps[0].dtype, feature_maps[0].device
        # anchors depend only on the shapes: build them once per (image, pyramid) geometry.  Building
        # them every step costs ten blocking H2D copies (the size / ratio tuples) in the middle of the
        # forward pass, i.e. a full host stall behind the backbone.
        key = (int(img_h), int(img_w), tuple(tuple(f.shape[-2:]) for f in feature_maps), dtype, str(device))
        cached = getattr(self, "_cache", None)
        if cached is not None and cached[0] == key:
            return [cached[1] for _ in image_list.image_sizes]
        per_level = []
        for f, s, a in zip(feature_maps, self.sizes, self.aspect_ratios):
            gh, gw = f.shape[-2:]
            sh, sw = img_h // gh, img_w // gw
            xs = torch.arange(0, gw, dtype=torch.float32, device=device) * sw
            ys = torch.arange(0, gh, dtype=torch.float32, device=device) * sh
            yy, xx = torch.meshgrid(ys, xs, indexing="ij")
            shifts = torch.stack((xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)), dim=1)
            per_level.append((shifts[:, None, :] + self._base(s, a, dtype, device)[None]).reshape(-1, 4))
        anchors = torch.cat(per_level)
        self._cache = (key, anchors)
        return [anchors for _ in image_list.image_sizes]


class RPNHead(nn.Module):
    def __init__(self, in_channels, num_anchors):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, in_channels, 3, padding=1)
        self.cls_logits = nn.Conv2d(in_channels, num_anchors, 1)
        self.bbox_pred = nn.Conv2d(in_channels, num_anchors * 4, 1)
        for l in self.children():
            nn.init.normal_(l.weight, std=0.01)
            nn.init.constant_(l.bias, 0)

    def forward(self, feats):
        """Same values as conv -> relu -> (cls_logits, bbox_pred) per level, in fewer passes over the 256-channel hidden
        tensor (550 MB at the 200 x 336 level of a batch of 8): bias + ReLU as one in-place epilogue with the sign mask for
        the backward pass, and the two 1 x 1 predictors as ONE convolution with A + 4A (+ padding) output channels -- the hidden tensor
        is read once forward, and its gradient comes out of one data-gradient kernel instead of two plus autograd's add."""
        logits, deltas = [], []
        A = self.cls_logits.out_channels
        fused = FUSE_HEAD and feats and feats[0].is_cuda
        if fused:
            # padded to a multiple of 4 output channels: ATen's channels-last bias-gradient reduction takes 0.69 ms for 15
            # channels at 200 x 336 and 0.02 ms for 12 or 16 (scratch/t_bias_grad.py)
            pad = (-5 * A) % 4
            w = torch.cat([self.cls_logits.weight, self.bbox_pred.weight] + ([self.conv.weight.new_zeros((pad,) + tuple(self.cls_logits.weight.shape[1:]))] if pad else []))
            b = torch.cat([self.cls_logits.bias, self.bbox_pred.bias] + ([self.conv.bias.new_zeros(pad)] if pad else []))
        for f in feats:
            if not fused:
                t = F.relu(conv1x1(f, self.conv.weight, self.conv.bias, self.conv))
                logits.append(self.cls_logits(t))
                deltas.append(self.bbox_pred(t))
                continue
            t = bias_act(conv1x1(f, self.conv.weight, None, self.conv), self.conv.bias, relu=True)   # shape-based kernel choice
            if torch.is_grad_enabled():
                both = F.conv2d(t, w, b)
            else:
                # inference: the same contraction as a GEMM on the NHWC view.  MIOpen's kernel for this 1 x 1 convolution with 16
                # output channels accumulates with atomics at every level but the largest: two runs on the same input differ in
                # the last bits, and with them the proposals (scratch/t_pred_conv.py: 20 of 20 runs differ; the GEMM: 0, same time)
                both = F.linear(t.permute(0, 2, 3, 1), w.view(w.shape[0], -1), b).permute(0, 3, 1, 2)
            logits.append(both[:, :A])
            deltas.append(both[:, A:5 * A])
        return logits, deltas


def _flatten_levels(logits, deltas):
    """per level [N, A, H, W] / [N, 4A, H, W] -> [N * sum(HWA), 1] / [N * sum(HWA), 4], level-major per image"""
    ls, ds = [], []
    for lg, dl in zip(logits, deltas):
        N, A, H, W = lg.shape
        ls.append(lg.permute(0, 2, 3, 1).reshape(N, -1, 1))
        ds.append(dl.view(N, A, 4, H, W).permute(0, 3, 4, 1, 2).reshape(N, -1, 4))
    return torch.cat(ls, dim=1).flatten(0, -2), torch.cat(ds, dim=1).reshape(-1, 4)


class RegionProposalNetwork(nn.Module):
    def __init__(self, anchor_generator, head, fg_iou_thresh, bg_iou_thresh, batch_size_per_image, positive_fraction,
                 pre_nms_top_n, post_nms_top_n, nms_thresh):
        super().__init__()
        self.anchor_generator = anchor_generator
        self.head = head
        self.box_coder = ops.BoxCoder((1.0, 1.0, 1.0, 1.0))
        self.matcher = ops.Matcher(fg_iou_thresh, bg_iou_thresh, allow_low_quality_matches=True)
        self.batch_size_per_image, self.positive_fraction = batch_size_per_image, positive_fraction
        self._pre, self._post = pre_nms_top_n, post_nms_top_n
        self.nms_thresh = nms_thresh
        self.min_size = 1e-3

    def _n(self, d):
        return d["training"] if self.training else d["testing"]

    def filter_proposals(self, proposals, objectness, image_sizes, counts, padded=False):
        sizes = torch.tensor([[float(s[1]), float(s[0])] for s in image_sizes], dtype=proposals.dtype)   # (w, h) per image
        if proposals.is_cuda:
            sizes = sizes.pin_memory().to(proposals.device, non_blocking=True)
        boxes, scores, count = self._filter(proposals, objectness, sizes, counts)
        if padded:
            # training: fixed [N, post, 4] + validity mask, no host synchronisation at all (rows past
            # the kept count repeat the image's best box and are masked by every consumer)
            ok = torch.arange(boxes.shape[1], device=boxes.device)[None, :] < count[:, None]
            return (boxes, ok), scores
        return self.unpad(boxes, scores, count)

    @staticmethod
    def unpad(boxes, scores, count):
        counts_host = count.tolist()        # the one synchronisation point
        return [boxes[i, :c] for i, c in enumerate(counts_host)], [scores[i, :c] for i, c in enumerate(counts_host)]

    def _filter(self, proposals, objectness, sizes, counts):
        """Per-level top-k, clip, NMS per level, top post-NMS over the levels: fixed shapes, no host synchronisation (what a HIP
        graph can hold).  sizes: [N, 2] (w, h) on the proposals' device.  Returns boxes [N, post, 4], scores [N, post] and
        the number of rows that are real, per image.

        torchvision runs ONE batched NMS per image over all levels (boxes of different levels moved apart); boxes of different
        levels never suppress each other, so the same survivors come out of one NMS set per (image, level) -- N x L independent
        sets of <= pre-NMS-top-n boxes, already sorted by the per-level top-k, instead of N sets of their sum: the greedy pass,
        sequential in blocks of 64 boxes, is 4 x shorter (32 instead of 138 block steps at the training sizes) and the
        suppression mask has 4 x fewer words.  The survivors of all levels are then ranked by score (top-k), which is the
        order batched_nms returns."""
        N, L = proposals.shape[0], len(counts)
        objectness = objectness.detach().reshape(N, -1)
        K = max(min(self._n(self._pre), n) for n in counts)
        hip = ops.hip_boxes_ok(objectness, proposals, sizes) and K <= 2048 and L <= 16
        if hip:
            # per-level top-k + gather + clip + size test: one launch (csrc/dib_topk.hip) for what follows in the else branch
            lv_scores, boxes, valid = ops.topk_levels_split_hip(objectness, counts, [min(self._n(self._pre), n) for n in counts], K,
                                                                proposals, sizes, self.min_size)
        else:
            lv_scores, boxes, valid = self._select_levels(proposals, objectness, sizes, counts, K)
        keep, count = ops.nms_sets_sorted(boxes.reshape(N * L, K, 4), valid.reshape(N * L, K), self.nms_thresh)
        kept = torch.arange(K, device=keep.device)[None, :] < count[:, None]
        scores = torch.where(kept, lv_scores.reshape(N * L, K).gather(1, keep), lv_scores.new_full((), float("-inf"))).reshape(N, L * K)
        boxes = boxes.reshape(N * L, K, 4).gather(1, keep[..., None].expand(-1, -1, 4)).reshape(N, L * K, 4)
        post = min(self._n(self._post), L * K)
        if hip and post <= 2048:
            scores, _, boxes, _ = ops.topk_levels_hip(scores, [L * K], [post], post, boxes)
            return boxes.reshape(N, post, 4), scores.reshape(N, post), count.reshape(N, L).sum(1).clamp(max=post)
        scores, top = scores.topk(post, dim=1)                               # survivors of all levels in score order
        boxes = boxes.gather(1, top[..., None].expand(-1, -1, 4))
        return boxes, scores, count.reshape(N, L).sum(1).clamp(max=post)

    def _select_levels(self, proposals, objectness, sizes, counts, K):
        """Per-level top-k, clipping and the size test as tensor expressions (CPU tensors; the checker of the HIP launch)."""
        N, L = proposals.shape[0], len(counts)
        lv_scores = objectness.new_full((N, L, K), float("-inf"))
        lv_boxes = proposals.new_zeros((N, L, K, 4))
        off = 0
        for l, n in enumerate(counts):                                       # top-k per level before NMS (sorted by score)
            k = min(self._n(self._pre), n)
            s, i = objectness[:, off:off + n].topk(k, dim=1)
            lv_scores[:, l, :k] = s
            lv_boxes[:, l, :k] = proposals[:, off:off + n].gather(1, i[..., None].expand(-1, -1, 4))
            off += n
        real = lv_scores > float("-inf")
        # Every image and level goes through the same tensor ops at once, and nothing is compacted before the NMS: boxes
        # that torchvision would drop as too small are flagged invalid instead (they are never kept and never suppress).
        x = lv_boxes[..., 0::2].clamp(min=0).minimum(sizes[:, None, None, 0:1])
        y = lv_boxes[..., 1::2].clamp(min=0).minimum(sizes[:, None, None, 1:2])
        boxes = torch.stack((x[..., 0], y[..., 0], x[..., 1], y[..., 1]), dim=-1)                        # clip_boxes_to_image
        valid = real & ((boxes[..., 2] - boxes[..., 0]) >= self.min_size) & ((boxes[..., 3] - boxes[..., 1]) >= self.min_size)
        return lv_scores, boxes, valid

    def propose_static(self, feats, anchors, sizes):
        """head + decoding + `_filter` on a list of feature maps: the sync-free part of `forward` in inference."""
        logits, deltas = self.head(feats)
        counts = [l[0].numel() for l in logits]
        objectness, deltas = _flatten_levels(logits, deltas)
        N = feats[0].shape[0]
        if ops.hip_boxes_ok(deltas, anchors):
            proposals = ops.decode_boxes_hip(self.box_coder, deltas.detach(), anchors).view(N, -1, 4)
        else:
            proposals = self.box_coder.decode(deltas.detach(), torch.cat([anchors] * N)).view(N, -1, 4)
        return self._filter(proposals, objectness, sizes, counts)

    def assign_targets(self, anchors, targets):
        """labels (1 / 0 / -1) and matched ground-truth boxes per anchor and image.  All images at once (ground truth padded to
        the longest list): the per-image form -- kept below as `assign_targets_per_image`, the checker -- is ~70 small launches
        per image, 577 per step at b = 8, and the host issues them slower than the GPU runs them (profiles/r4_train_step_conv.txt)."""
        same = all(a is anchors[0] for a in anchors)
        if not same or not targets:
            return self.assign_targets_per_image(anchors, targets)
        a = anchors[0]
        if ops.hip_boxes_ok(a, *[t["boxes"] for t in targets]) and max(t["boxes"].shape[0] for t in targets) <= 256 and len(targets) <= 32:
            # CUDA: IoU + Matcher in two launches, labels in three, matched boxes in one (csrc/dib_detect.hip)
            gt_cat, offs = ops.cat_boxes([t["boxes"] for t in targets])
            m = ops.match_boxes_hip(self.matcher, gt_cat, offs, a, shared=True)
            lab = (m.clamp(max=0) + 1).to(torch.float32)                      # >= 0 -> 1, BELOW_LOW (-1) -> 0, BETWEEN (-2) -> -1
            return lab, ops.encode_matched_hip(self.box_coder, gt_cat, offs, m, a, True, want_targets=False, want_matched=True)[1]
        gt, valid = ops.pad_boxes([t["boxes"] for t in targets])
        m = ops.match_batched(self.matcher, ops.box_iou_batched(gt, a), valid)                  # [N, A]
        matched = gt.gather(1, m.clamp(min=0)[..., None].expand(-1, -1, 4))                   # images without boxes: zeros
        lab = (m >= 0).to(torch.float32)
        lab = torch.where(m == ops.Matcher.BELOW_LOW, lab.new_zeros(()), lab)
        lab = torch.where(m == ops.Matcher.BETWEEN, lab.new_full((), -1.0), lab)               # ignored by the sampler
        return lab, matched

    def assign_and_encode(self, anchors, targets):
        """(labels [N, A], regression targets [N, A, 4]) for the loss."""
        a = anchors[0]
        if (all(x is a for x in anchors) and targets and ops.hip_boxes_ok(a, *[t["boxes"] for t in targets])
                and max(t["boxes"].shape[0] for t in targets) <= 256 and len(targets) <= 32):
            gt_cat, offs = ops.cat_boxes([t["boxes"] for t in targets])
            m = ops.match_boxes_hip(self.matcher, gt_cat, offs, a, shared=True)
            lab = (m.clamp(max=0) + 1).to(torch.float32)
            return lab, ops.encode_matched_hip(self.box_coder, gt_cat, offs, m, a, True)[0]
        labels, matched = self.assign_targets(anchors, targets)
        return labels, self.box_coder.encode(matched.reshape(-1, 4), torch.cat(anchors)).view(len(anchors), -1, 4)

    def assign_targets_per_image(self, anchors, targets):
        labels, matched = [], []
        for a, t in zip(anchors, targets):
            gt = t["boxes"]
            if gt.numel() == 0:
                matched.append(torch.zeros_like(a))
                labels.append(torch.zeros((a.shape[0],), dtype=torch.float32, device=a.device))
                continue
            m = self.matcher(ops.box_iou(gt, a))
            matched.append(gt[m.clamp(min=0)])
            lab = (m >= 0).to(torch.float32)
            lab[m == ops.Matcher.BELOW_LOW] = 0.0
            lab[m == ops.Matcher.BETWEEN] = -1.0       # ignored by the sampler
            labels.append(lab)
        return torch.stack(labels), torch.stack(matched)

    def compute_loss(self, objectness, deltas, labels, regression_targets):
        """labels [N, A] (1 / 0 / -1), regression_targets [N, A, 4].  Fixed-size sampling + masked
        sums: the same losses as gathering a ragged index list (objectness: mean BCE over the sampled
        anchors; boxes: smooth-L1 sum over sampled positives / number sampled), with no host sync."""
        N, A = labels.shape
        pos_idx, pos_ok, neg_idx, neg_ok = ops.sample_pos_neg_fixed(labels, self.batch_size_per_image, self.positive_fraction)
        sel, ok = torch.cat([pos_idx, neg_idx], dim=1), torch.cat([pos_ok, neg_ok], dim=1)
        n_sel = ok.sum().clamp(min=1)
        obj = objectness.reshape(N, A).gather(1, sel)
        bce = F.binary_cross_entropy_with_logits(obj, labels.gather(1, sel), reduction="none")
        obj_loss = torch.where(ok, bce, bce.new_zeros(())).sum() / n_sel
        g = pos_idx[..., None].expand(-1, -1, 4)
        l1 = F.smooth_l1_loss(deltas.reshape(N, A, 4).gather(1, g), regression_targets.gather(1, g), beta=1 / 9, reduction="none")
        box_loss = torch.where(pos_ok[..., None], l1, l1.new_zeros(())).sum() / n_sel
        return obj_loss, box_loss

    def forward(self, images, features, targets=None):
        feats = list(features.values())
        logits, deltas = self.head(feats)
        anchors = self.anchor_generator(images, feats)
        counts = [l[0].numel() for l in logits]
        objectness, deltas = _flatten_levels(logits, deltas)
        N = len(anchors)
        if all(a is anchors[0] for a in anchors) and ops.hip_boxes_ok(deltas, anchors[0]):
            proposals = ops.decode_boxes_hip(self.box_coder, deltas.detach(), anchors[0]).view(N, -1, 4)      # one launch
        else:
            proposals = self.box_coder.decode(deltas.detach(), torch.cat(anchors)).view(N, -1, 4)
        boxes, _ = self.filter_proposals(proposals, objectness, images.image_sizes, counts, padded=self.training)
        losses = {}
        if self.training:
            assert targets is not None
            labels, reg_targets = self.assign_and_encode(anchors, targets)                   # [N, A], [N, A, 4]
            obj, box = self.compute_loss(objectness, deltas, labels, reg_targets)
            losses = {"loss_objectness": obj, "loss_rpn_box_reg": box}
        return boxes, losses
