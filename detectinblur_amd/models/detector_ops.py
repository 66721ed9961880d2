"""Box / RoI primitives of the detector (torchvision is not installed, so they live here).

On CUDA tensors `roi_align` and `nms` run the hand-written HIP kernels of csrc/dib_roi.hip through
the C ABI (and raise if libdib_hip.so is missing); on CPU tensors -- the gloo DDP tests and the
fp32 reference used by the numerics tests -- they run the plain-PyTorch restatements below.
Semantics: torchvision.ops.{roi_align(aligned=False), nms, batched_nms, box_iou, clip_boxes_to_image,
remove_small_boxes}, which the reference's RPN / RoIHeads call (reference models/faster_rcnn.py:7-16).
"""
import ctypes
import math

import torch

from .. import _lib

# ------------------------------------------------------------------------------------------------
# boxes
# ------------------------------------------------------------------------------------------------


def box_area(b):
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


def box_iou(a, b):
    """[N,4] x [M,4] -> [N,M]"""
    lt = torch.max(a[:, None, :2], b[None, :, :2])
    rb = torch.min(a[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (box_area(a)[:, None] + box_area(b)[None, :] - inter)


def pad_boxes(boxes_per_image, width=None):
    """Ragged per-image box lists -> ([N, G, 4] padded with zeros, valid [N, G] bool); G = the longest list (at least 1).
    Shapes are host knowledge: no device synchronisation."""
    G = max([b.shape[0] for b in boxes_per_image] + [1]) if width is None else width
    ref = boxes_per_image[0]
    out = ref.new_zeros((len(boxes_per_image), G, 4))
    for i, b in enumerate(boxes_per_image):
        if b.shape[0]:
            out[i, :b.shape[0]] = b
    counts = torch.tensor([b.shape[0] for b in boxes_per_image])
    valid = torch.arange(G)[None, :] < counts[:, None]
    if ref.is_cuda:
        valid = valid.pin_memory().to(ref.device, non_blocking=True)
    return out, valid


def box_iou_batched(a, b):
    """[N, G, 4] x ([N, M, 4] or [M, 4]) -> [N, G, M]: `box_iou` of every image at once, the same arithmetic per pair."""
    if b.dim() == 2:
        b = b[None]
    lt = torch.max(a[:, :, None, :2], b[:, None, :, :2])
    rb = torch.min(a[:, :, None, 2:], b[:, None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    area_a = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1])
    area_b = (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
    return inter / (area_a[:, :, None] + area_b[:, None, :] - inter)


def clip_boxes_to_image(boxes, size):
    h, w = size
    x = boxes[..., 0::2].clamp(min=0, max=w)
    y = boxes[..., 1::2].clamp(min=0, max=h)
    return torch.stack((x[..., 0], y[..., 0], x[..., 1], y[..., 1]), dim=-1)


def remove_small_boxes(boxes, min_size):
    ws, hs = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    return torch.where((ws >= min_size) & (hs >= min_size))[0]


class BoxCoder(object):
    """R-CNN box parameterisation (dx, dy, dw, dh) with per-coordinate weights."""

    def __init__(self, weights, clip=math.log(1000.0 / 16)):
        self.weights = weights
        self.clip = clip

    def encode(self, reference, proposals):
        wx, wy, ww, wh = self.weights
        pw = proposals[:, 2] - proposals[:, 0]
        ph = proposals[:, 3] - proposals[:, 1]
        px = proposals[:, 0] + 0.5 * pw
        py = proposals[:, 1] + 0.5 * ph
        gw = reference[:, 2] - reference[:, 0]
        gh = reference[:, 3] - reference[:, 1]
        gx = reference[:, 0] + 0.5 * gw
        gy = reference[:, 1] + 0.5 * gh
        return torch.stack((wx * (gx - px) / pw, wy * (gy - py) / ph, ww * torch.log(gw / pw), wh * torch.log(gh / ph)), dim=1)

    def decode(self, deltas, boxes):
        """deltas [N, 4*k], boxes [N, 4] -> [N, k, 4]"""
        boxes = boxes.to(deltas.dtype)
        wx, wy, ww, wh = self.weights
        w = boxes[:, 2] - boxes[:, 0]
        h = boxes[:, 3] - boxes[:, 1]
        cx = boxes[:, 0] + 0.5 * w
        cy = boxes[:, 1] + 0.5 * h
        dx = deltas[:, 0::4] / wx
        dy = deltas[:, 1::4] / wy
        dw = torch.clamp(deltas[:, 2::4] / ww, max=self.clip)
        dh = torch.clamp(deltas[:, 3::4] / wh, max=self.clip)
        pcx = dx * w[:, None] + cx[:, None]
        pcy = dy * h[:, None] + cy[:, None]
        pw = torch.exp(dw) * w[:, None]
        ph = torch.exp(dh) * h[:, None]
        return torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph), dim=2)


class Matcher(object):
    """Assigns each prediction the ground truth of highest IoU, -1 below `low`, -2 between."""
    BELOW_LOW = -1
    BETWEEN = -2

    def __init__(self, high, low, allow_low_quality_matches=False):
        self.high, self.low, self.allow_low = high, low, allow_low_quality_matches

    def __call__(self, quality):           # quality: [G, N]
        if quality.numel() == 0:
            raise ValueError("No ground-truth boxes or proposals available for one of the images during training")
        vals, matches = quality.max(dim=0)
        all_matches = matches.clone() if self.allow_low else None
        matches[vals < self.low] = self.BELOW_LOW
        matches[(vals >= self.low) & (vals < self.high)] = self.BETWEEN
        if self.allow_low:
            # every ground truth keeps the prediction(s) it overlaps best, however poorly
            best_per_gt = quality.max(dim=1)[0]
            restore = (quality == best_per_gt[:, None]).any(dim=0)       # a mask, not torch.where: no host sync
            matches = torch.where(restore, all_matches, matches)
        return matches


def match_batched(matcher, quality, valid):
    """`Matcher.__call__` for every image at once: quality [N, G, M] (IoU of padded ground truth x predictions), valid [N, G]
    (False = padding row).  Returns matches [N, M] with the same values the per-image call gives on the valid rows; an image
    without ground truth comes out all BELOW_LOW (its per-image call is skipped by the callers, who label such images 0)."""
    q = torch.where(valid[:, :, None], quality, quality.new_full((), -1.0))      # padding never wins a maximum
    vals, matches = q.max(dim=1)
    all_matches = matches.clone() if matcher.allow_low else None
    matches = torch.where(vals < matcher.low, matches.new_full((), Matcher.BELOW_LOW), matches)
    matches = torch.where((vals >= matcher.low) & (vals < matcher.high), matches.new_full((), Matcher.BETWEEN), matches)
    if matcher.allow_low:
        best_per_gt = q.max(dim=2)[0]
        restore = ((q == best_per_gt[:, :, None]) & valid[:, :, None]).any(dim=1)
        matches = torch.where(restore, all_matches, matches)
    return matches


# ---- the same bookkeeping as HIP launches (csrc/dib_detect.hip) for CUDA tensors --------------------------------------------
# Ragged ground truth travels as ONE concatenated tensor + host offsets (no padding, no per-image copies).
HIP_BOXES = True      # False: the tensor expressions above on the GPU as well (what the kernels are tested against)


def hip_boxes_ok(*tensors):
    return HIP_BOXES and all(t.is_cuda and t.dtype == torch.float32 for t in tensors)


def cat_boxes(boxes_per_image):
    """(gt_cat [T, 4] contiguous float32 -- None when T == 0 --, offsets: N + 1 host ints)"""
    offs, t = [0], 0
    for b in boxes_per_image:
        t += int(b.shape[0])
        offs.append(t)
    if t == 0:
        return None, offs
    live = [b for b in boxes_per_image if b.shape[0]]
    gt = (live[0] if len(live) == 1 else torch.cat(live)).contiguous()
    return (gt.clone() if gt.data_ptr() % 16 else gt), offs


def match_boxes_hip(matcher, gt_cat, offsets, cand, shared):
    """Matcher(box_iou(gt_i, cand_i)) for every image in one or two launches: matches [N, M] int64 (index into image i's own
    list, -1 / -2 as Matcher).  cand: [N, M, 4], or [M, 4] shared by every image."""
    N = len(offsets) - 1
    cand = cand.contiguous()
    M = cand.shape[-2]
    match = torch.empty((N, M), dtype=torch.int64, device=cand.device)
    best = torch.empty(max(offsets[-1], 1), dtype=torch.int32, device=cand.device) if matcher.allow_low else None
    _lib.check(_lib.lib().dib_box_match(gt_cat.data_ptr() if gt_cat is not None else None, _lib.int_array(offsets), N, cand.data_ptr(), M,
                                        int(bool(shared)), float(matcher.high), float(matcher.low), int(bool(matcher.allow_low)),
                                        best.data_ptr() if best is not None else None, match.data_ptr(),
                                        _lib.stream_of(cand)))
    return match


def encode_matched_hip(coder, gt_cat, offsets, match, cand, shared, want_targets=True, want_matched=False):
    """(BoxCoder.encode(gt[match.clamp(min=0)], cand) [N, M, 4] | None, the matched boxes [N, M, 4] | None) in one launch."""
    N = len(offsets) - 1
    cand, match = cand.contiguous(), match.contiguous()
    M = cand.shape[-2]
    tg = torch.empty((N, M, 4), dtype=torch.float32, device=cand.device) if want_targets else None
    mb = torch.empty((N, M, 4), dtype=torch.float32, device=cand.device) if want_matched else None
    wx, wy, ww, wh = coder.weights
    _lib.check(_lib.lib().dib_box_encode_matched(gt_cat.data_ptr() if gt_cat is not None else None, _lib.int_array(offsets), N, match.data_ptr(),
                                                 cand.data_ptr(), M, int(bool(shared)), float(wx), float(wy), float(ww), float(wh),
                                                 tg.data_ptr() if tg is not None else None, mb.data_ptr() if mb is not None else None,
                                                 _lib.stream_of(cand)))
    return tg, mb


def pool_boxes_hip(proposals, gt_cat, offsets, g_pad):
    """[N, P + g_pad, 4]: the proposals of every image followed by its ground truth and [0, 0, 1, 1] padding rows, in one launch."""
    proposals = proposals.contiguous()
    N, P = proposals.shape[:2]
    out = torch.empty((N, P + g_pad, 4), dtype=torch.float32, device=proposals.device)
    _lib.check(_lib.lib().dib_box_pool(proposals.data_ptr(), P, gt_cat.data_ptr() if gt_cat is not None else None, _lib.int_array(offsets), N, g_pad,
                                       out.data_ptr(), _lib.stream_of(proposals)))
    return out


def pool_labels_hip(match, gt_labels_cat, offsets, ok, P):
    """Class of every pool row [N, M] int64 (0 background, -1 ignored / padding) from the matches, in one launch."""
    match = match.contiguous()
    N, M = match.shape
    ok = ok.contiguous() if ok is not None else None
    out = torch.empty((N, M), dtype=torch.int64, device=match.device)
    _lib.check(_lib.lib().dib_box_labels(match.data_ptr(), gt_labels_cat.data_ptr() if gt_labels_cat is not None else None, _lib.int_array(offsets), N,
                                         ok.data_ptr() if ok is not None else None, P, M, out.data_ptr(), _lib.stream_of(match)))
    return out


def topk_levels_hip(values, counts, ks, K, boxes=None, clip_wh=None, min_size=0.0, want_index=False):
    """Sorted top-k per (row, level) in one launch (csrc/dib_topk.hip).  values [N, A] float32, level l = the next counts[l] columns,
    ks[l] <= K <= 2048 winners each.  Returns (scores [N, L, K] padded with -inf, index [N, L, K] | None, boxes [N, L, K, 4] | None --
    gathered from `boxes` [N, A, 4], clipped to clip_wh [N, 2] = (w, h) if given --, valid [N, L, K] bool | None)."""
    values = values.contiguous()
    N, A = values.shape
    L = len(counts)
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + int(c))
    dev = values.device
    scores = torch.empty((N, L, K), dtype=torch.float32, device=dev)
    index = torch.empty((N, L, K), dtype=torch.int64, device=dev) if want_index else None
    out_boxes = valid = None
    if boxes is not None:
        boxes = boxes.contiguous()
        out_boxes = torch.empty((N, L, K, 4), dtype=torch.float32, device=dev)
        valid = torch.empty((N, L, K), dtype=torch.bool, device=dev)
    if clip_wh is not None:
        clip_wh = clip_wh.contiguous()
    _lib.check(_lib.lib().dib_topk_levels(values.data_ptr(), A, N, _lib.int_array(offs), _lib.int_array([int(k) for k in ks]), L, K,
                                          boxes.data_ptr() if boxes is not None else None, clip_wh.data_ptr() if clip_wh is not None else None,
                                          float(min_size), scores.data_ptr(), index.data_ptr() if index is not None else None,
                                          out_boxes.data_ptr() if out_boxes is not None else None, valid.data_ptr() if valid is not None else None,
                                          _lib.stream_of(values)))
    return scores, index, out_boxes, valid


def det_candidates_hip(coder, class_logits, box_regression, rois, image_shape, score_thresh, min_size):
    """One image's detection candidates, class-major, in one launch (csrc/dib_detect.hip: softmax + decode + clip + tests).
    Returns (scores [C - 1, R] with -inf where dropped, boxes [C - 1, R, 4], stats int32 [2] = (candidates kept, bits of the largest
    coordinate among them))."""
    lg, dl, rois = class_logits.contiguous(), box_regression.contiguous(), rois.contiguous()
    R, C = lg.shape
    scores = torch.empty((C - 1, R), dtype=torch.float32, device=lg.device)
    boxes = torch.empty((C - 1, R, 4), dtype=torch.float32, device=lg.device)
    stats = torch.empty(2, dtype=torch.int32, device=lg.device)
    wx, wy, ww, wh = coder.weights
    _lib.check(_lib.lib().dib_det_candidates(lg.data_ptr(), dl.data_ptr(), rois.data_ptr(), R, C, float(image_shape[0]), float(image_shape[1]), float(wx),
                                             float(wy), float(ww), float(wh), float(coder.clip), float(score_thresh), float(min_size),
                                             scores.data_ptr(), boxes.data_ptr(), stats.data_ptr(), _lib.stream_of(lg)))
    return scores, boxes, stats


TOPK_SPLIT = 32768      # longest row dib_topk_levels is asked to select from in one piece (roi_heads: the detections' class-major score rows)
TOPK_PIECE = 16384      # the RPN filter's levels longer than this are selected in pieces (one workgroup sweeps a piece: ~0.8 us per 1000
                        # elements + ~25 us per piece whatever its length; 800 x 1333, b = 1: 166 us unsplit, 73 us at 32768, 66 us here)
TOPK_MAX_LEVELS = 32    # csrc/dib_topk.hip


def topk_levels_split_hip(values, counts, ks, K, boxes, clip_wh, min_size):
    """`topk_levels_hip(..)[0, 2, 3]` with long levels cut into consecutive pieces whose winners a second launch merges: the order
    (descending score, ascending index) is a total order, so the winners of a level are among the winners of its pieces, and the
    pieces' winners, laid out piece after piece, are still in ascending index order among equal scores -- the result is identical."""
    pieces, piece_ks, groups = [], [], []
    for c, k in zip(counts, ks):
        n = max(1, -(-int(c) // TOPK_PIECE))
        base, rem = divmod(int(c), n)
        sizes = [base + (1 if i < rem else 0) for i in range(n)]
        pieces += sizes
        piece_ks += [min(int(k), s) for s in sizes]
        groups.append(n)
    if len(pieces) == len(counts) or len(pieces) > TOPK_MAX_LEVELS:
        s, _, b, v = topk_levels_hip(values, counts, ks, K, boxes, clip_wh, min_size)
        return s, b, v
    s1, _, b1, _ = topk_levels_hip(values, pieces, piece_ks, K, boxes, clip_wh, min_size)
    N = values.shape[0]
    s, _, b, v = topk_levels_hip(s1.reshape(N, -1), [g * K for g in groups], ks, K, b1.reshape(N, -1, 4), clip_wh, min_size)
    return s, b, v


def decode_boxes_hip(coder, deltas, anchors):
    """BoxCoder.decode(deltas [R, 4], anchors [A, 4] repeated R / A times) -> [R, 4] in one launch (no repeated anchor tensor)."""
    deltas, anchors = deltas.contiguous(), anchors.contiguous()
    out = torch.empty_like(deltas)
    wx, wy, ww, wh = coder.weights
    _lib.check(_lib.lib().dib_box_decode(deltas.data_ptr(), anchors.data_ptr(), deltas.shape[0], anchors.shape[0], float(wx), float(wy), float(ww),
                                         float(wh), float(coder.clip), out.data_ptr(), _lib.stream_of(deltas)))
    return out


def sample_pos_neg(labels_per_image, batch_size, positive_fraction):
    """Random subset of at most `batch_size` entries per image with up to `positive_fraction`
    positives (label >= 1), rest negatives (label == 0).  Returns per-image index tensors."""
    out = []
    for labels in labels_per_image:
        pos = torch.where(labels >= 1)[0]
        neg = torch.where(labels == 0)[0]
        n_pos = min(pos.numel(), int(batch_size * positive_fraction))
        n_neg = min(neg.numel(), batch_size - n_pos)
        pos = pos[torch.randperm(pos.numel(), device=pos.device)[:n_pos]]
        neg = neg[torch.randperm(neg.numel(), device=neg.device)[:n_neg]]
        out.append((pos, neg))
    return out


def sample_pos_neg_fixed(labels, batch_size, positive_fraction):
    """The same sampling rule without data-dependent shapes (hence without host synchronisation):
    labels [N, A] (>= 1 positive, 0 negative, < 0 ignored).  Per row, a uniformly random subset of
    min(#pos, P) positives, P = int(batch_size * positive_fraction), and a uniformly random subset of
    min(#neg, batch_size - n_pos) negatives -- drawn as the smallest of i.i.d. uniform keys instead of
    a randperm prefix.  Returns (pos_idx [N, P'], pos_ok [N, P'], neg_idx [N, B'], neg_ok [N, B']);
    entries whose `ok` flag is False are padding and must be masked by the caller."""
    N, A = labels.shape
    P = min(int(batch_size * positive_fraction), A)
    Bn = min(batch_size, A)
    keys = torch.rand((2, N, A), device=labels.device)
    far = 2.0
    vp, pos_idx = torch.where(labels >= 1, keys[0], far).topk(P, dim=1, largest=False, sorted=True)
    pos_ok = vp < far
    n_pos = pos_ok.sum(dim=1, keepdim=True)
    vn, neg_idx = torch.where(labels == 0, keys[1], far).topk(Bn, dim=1, largest=False, sorted=True)
    neg_ok = (vn < far) & (torch.arange(Bn, device=labels.device)[None, :] < (batch_size - n_pos))
    return pos_idx, pos_ok, neg_idx, neg_ok


# ------------------------------------------------------------------------------------------------
# NMS
# ------------------------------------------------------------------------------------------------

def _nms_torch(boxes, scores, thr):
    order = scores.argsort(descending=True)
    b = boxes[order]
    n = b.shape[0]
    if n == 0:
        return order
    suppress = torch.triu(box_iou(b, b) > thr, diagonal=1)
    removed = torch.zeros(n, dtype=torch.bool, device=b.device)
    keep = []
    sup = suppress.cpu()
    rem = removed.cpu()
    for i in range(n):
        if not rem[i]:
            keep.append(i)
            rem |= sup[i]
    return order[torch.as_tensor(keep, dtype=torch.long, device=b.device)]


def nms(boxes, scores, iou_threshold):
    """Indices of the boxes that survive greedy NMS, by descending score."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    if not boxes.is_cuda:
        return _nms_torch(boxes.float(), scores.float(), iou_threshold)
    l = _lib.lib()
    order = scores.argsort(descending=True)
    b = boxes.float()[order].contiguous()
    n = b.shape[0]
    if n > 16384:
        order, b, n = order[:16384], b[:16384].contiguous(), 16384
    ws = torch.empty(l.dib_nms_workspace_bytes(n), dtype=torch.uint8, device=b.device)
    keep = torch.empty(n, dtype=torch.int64, device=b.device)
    count = torch.empty(1, dtype=torch.int32, device=b.device)
    _lib.check(l.dib_nms(b.data_ptr(), n, float(iou_threshold), ws.data_ptr(), keep.data_ptr(), count.data_ptr(),
                         _lib.stream_of(b)))
    return order[keep[:int(count.item())]]


def nms_sets_sorted(boxes, valid, iou_threshold):
    """Greedy NMS on B independent box sets at once.  boxes [B, n, 4], every set already sorted by
    descending score; valid [B, n] bool (False = the box takes no part) or None.
    Returns (keep [B, n] int64, count [B] int32): keep[b, :count[b]] are the surviving positions in
    score order, the rest of the row is 0.  No host synchronisation on the GPU path."""
    B, n = boxes.shape[0], boxes.shape[1]
    keep = torch.zeros((B, n), dtype=torch.int64, device=boxes.device)
    count = torch.zeros((B,), dtype=torch.int32, device=boxes.device)
    if B == 0 or n == 0:
        return keep, count
    if not boxes.is_cuda:
        fake = torch.arange(n, 0, -1, dtype=torch.float32)
        for b in range(B):
            idx = torch.arange(n) if valid is None else torch.where(valid[b])[0]
            k = idx[_nms_torch(boxes[b, idx].float(), fake[idx], iou_threshold)]
            keep[b, :k.numel()] = k
            count[b] = k.numel()
        return keep, count
    if n > 16384:
        raise ValueError("nms_sets_sorted: at most 16384 boxes per set")
    l = _lib.lib()
    b = boxes.float().contiguous()
    v = None if valid is None else valid.to(torch.uint8).contiguous()
    ws = torch.empty(B * l.dib_nms_workspace_bytes(n), dtype=torch.uint8, device=b.device)
    _lib.check(l.dib_nms_batched(b.data_ptr(), v.data_ptr() if v is not None else None, B, n, float(iou_threshold),
                                 ws.data_ptr(), keep.data_ptr(), count.data_ptr(), _lib.stream_of(b)))
    return keep, count


def batched_nms(boxes, scores, groups, iou_threshold):
    """NMS within each group (FPN level / class): boxes of different groups are moved apart."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    offsets = groups.to(boxes) * (boxes.max() + 1)
    return nms(boxes + offsets[:, None], scores, iou_threshold)


def coco_box_iou(dt, gt, iscrowd=None):
    """pycocotools' bbIou on the GPU (include/dib.h: dib_coco_box_iou).  dt [m, 4], gt [n, 4]: (x, y, w, h);
    iscrowd [n] bool / uint8 or None.  Returns float64 [m, n] laid out as maskUtils.iou returns it
    (detections along the rows)."""
    if not dt.is_cuda:
        raise RuntimeError("coco_box_iou runs on the GPU (the CPU routine is pycocotools' own)")
    d = dt.to(torch.float64).contiguous()
    g = gt.to(device=d.device, dtype=torch.float64).contiguous()
    m, n = d.shape[0], g.shape[0]
    out = torch.zeros((n, m), dtype=torch.float64, device=d.device)
    c = None if iscrowd is None else iscrowd.to(device=d.device, dtype=torch.uint8).contiguous()
    _lib.check(_lib.lib().dib_coco_box_iou(d.data_ptr(), g.data_ptr(), c.data_ptr() if c is not None else None, m, n,
                                           out.data_ptr(), _lib.stream_of(d)))
    return out.t()


# ------------------------------------------------------------------------------------------------
# RoIAlign
# ------------------------------------------------------------------------------------------------

def _bilinear_torch(feat, b, y, x):
    """feat [N,C,H,W]; b [K] batch index; y, x [K, S] sample coordinates -> [K, C, S]"""
    H, W = feat.shape[-2:]
    outside = (y < -1) | (y > H) | (x < -1) | (x > W)
    y = y.clamp(min=0)
    x = x.clamp(min=0)
    y0 = y.floor().long().clamp(max=H - 1)
    x0 = x.floor().long().clamp(max=W - 1)
    y1 = (y0 + 1).clamp(max=H - 1)
    x1 = (x0 + 1).clamp(max=W - 1)
    y = torch.where(y0 >= H - 1, y0.to(y), y)
    x = torch.where(x0 >= W - 1, x0.to(x), x)
    ly, lx = y - y0, x - x0
    hy, hx = 1 - ly, 1 - lx
    fb = feat[b]                                           # [K, C, H, W]
    K, C = fb.shape[:2]
    flat = fb.reshape(K, C, H * W)

    def g(yy, xx):
        return flat.gather(2, (yy * W + xx)[:, None, :].expand(K, C, -1))
    v = (hy * hx)[:, None] * g(y0, x0) + (hy * lx)[:, None] * g(y0, x1) + (ly * hx)[:, None] * g(y1, x0) + (ly * lx)[:, None] * g(y1, x1)
    return v * (~outside)[:, None].to(v)


def roi_align_torch(feat, rois, spatial_scale, pooled, sampling_ratio, aligned=False):
    """Plain-PyTorch fp32 RoIAlign (differentiable): the CPU path and the numerics reference."""
    K = rois.shape[0]
    C = feat.shape[1]
    if K == 0:
        return feat.new_zeros((0, C, pooled, pooled))
    b = rois[:, 0].long()
    off = 0.5 if aligned else 0.0
    x1, y1, x2, y2 = (rois[:, i] * spatial_scale - off for i in (1, 2, 3, 4))
    rw, rh = x2 - x1, y2 - y1
    if not aligned:
        rw, rh = rw.clamp(min=1.0), rh.clamp(min=1.0)
    bw, bh = rw / pooled, rh / pooled
    assert sampling_ratio > 0, "adaptive sampling is not restated in the torch path"
    g = sampling_ratio
    p = torch.arange(pooled, device=feat.device, dtype=feat.dtype)
    s = (torch.arange(g, device=feat.device, dtype=feat.dtype) + 0.5) / g
    ys = y1[:, None, None] + (p[None, :, None] + s[None, None, :]) * bh[:, None, None]       # [K, P, g]
    xs = x1[:, None, None] + (p[None, :, None] + s[None, None, :]) * bw[:, None, None]
    yy = ys[:, :, None, :, None].expand(K, pooled, pooled, g, g).reshape(K, -1)
    xx = xs[:, None, :, None, :].expand(K, pooled, pooled, g, g).reshape(K, -1)
    v = _bilinear_torch(feat, b, yy, xx).reshape(K, C, pooled, pooled, g * g)
    return v.mean(dim=-1)


class _RoIAlignHIP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, rois, spatial_scale, pooled, sampling_ratio, aligned):
        feat = feat.contiguous()
        rois = rois.contiguous().float()
        N, C, H, W = feat.shape
        K = rois.shape[0]
        out = torch.empty((K, C, pooled, pooled), dtype=torch.float32, device=feat.device)
        _lib.check(_lib.lib().dib_roi_align_forward(feat.data_ptr(), rois.data_ptr(), K, C, H, W, float(spatial_scale), pooled,
                                                    sampling_ratio, int(aligned), out.data_ptr(),
                                                    _lib.stream_of(out)))
        ctx.save_for_backward(rois)
        ctx.meta = (tuple(feat.shape), float(spatial_scale), pooled, sampling_ratio, int(aligned))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (rois,) = ctx.saved_tensors
        shape, scale, pooled, sr, aligned = ctx.meta
        N, C, H, W = shape
        grad = torch.zeros(shape, dtype=torch.float32, device=grad_out.device)
        g = grad_out.contiguous().float()
        _lib.check(_lib.lib().dib_roi_align_backward(g.data_ptr(), rois.data_ptr(), rois.shape[0], C, H, W, scale, pooled, sr,
                                                     aligned, grad.data_ptr(), _lib.stream_of(grad)))
        return grad, None, None, None, None, None


def _is_nhwc(f):
    return (f.is_cuda and f.dtype == torch.float32 and f.dim() == 4 and f.shape[1] > 1
            and f.is_contiguous(memory_format=torch.channels_last))


class _RoIAlignNHWC(torch.autograd.Function):
    """Channels-last RoIAlign over 1..4 pyramid levels in one launch (include/dib.h:
    dib_roi_align_nhwc_forward / _backward); `level` is a device int32 [K] tensor."""

    @staticmethod
    def forward(ctx, rois, level, scales, pooled, sampling_ratio, aligned, *feats):
        rois = rois.contiguous().float()
        K, C = rois.shape[0], feats[0].shape[1]
        hs, ws = _lib.int_array([f.shape[2] for f in feats]), _lib.int_array([f.shape[3] for f in feats])
        sc = (ctypes.c_float * len(feats))(*[float(s) for s in scales])
        out = torch.empty((K, C, pooled, pooled), dtype=torch.float32, device=rois.device)
        _lib.check(_lib.lib().dib_roi_align_nhwc_forward(
            _lib.ptr_array([f.data_ptr() for f in feats]), hs, ws, sc, len(feats), rois.data_ptr(),
            level.data_ptr() if level is not None else None, K, C, pooled, sampling_ratio, int(aligned), out.data_ptr(),
            _lib.stream_of(out)))
        ctx.save_for_backward(rois, level) if level is not None else ctx.save_for_backward(rois)
        ctx.meta = ([tuple(f.shape) for f in feats], [float(s) for s in scales], pooled, sampling_ratio, int(aligned))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        saved = ctx.saved_tensors
        rois, level = saved[0], (saved[1] if len(saved) > 1 else None)
        shapes, scales, pooled, sr, aligned = ctx.meta
        grads = [torch.empty(sh, dtype=torch.float32, device=grad_out.device, memory_format=torch.channels_last).zero_()
                 for sh in shapes]
        g = grad_out.contiguous().float()
        hs, ws = _lib.int_array([sh[2] for sh in shapes]), _lib.int_array([sh[3] for sh in shapes])
        sc = (ctypes.c_float * len(shapes))(*scales)
        _lib.check(_lib.lib().dib_roi_align_nhwc_backward(
            g.data_ptr(), hs, ws, sc, len(shapes), rois.data_ptr(), level.data_ptr() if level is not None else None,
            rois.shape[0], shapes[0][1], pooled, sr, aligned, _lib.ptr_array([x.data_ptr() for x in grads]),
            _lib.stream_of(g)))
        return (None, None, None, None, None, None) + tuple(grads)


def roi_align(feat, rois, spatial_scale, pooled, sampling_ratio, aligned=False):
    """feat [N,C,H,W] fp32, rois [K,5] (batch index, x1, y1, x2, y2) -> [K,C,pooled,pooled]."""
    if _is_nhwc(feat) and pooled <= 7:
        return _RoIAlignNHWC.apply(rois, None, [spatial_scale], pooled, sampling_ratio, aligned, feat)
    if feat.is_cuda:
        if feat.dtype != torch.float32:
            return _RoIAlignHIP.apply(feat.float(), rois, spatial_scale, pooled, sampling_ratio, aligned).to(feat.dtype)
        return _RoIAlignHIP.apply(feat, rois, spatial_scale, pooled, sampling_ratio, aligned)
    return roi_align_torch(feat, rois.to(feat.dtype), spatial_scale, pooled, sampling_ratio, aligned)


class MultiScaleRoIAlign(torch.nn.Module):
    """FPN RoI pooling: each RoI is pooled from the level k = floor(4 + log2(sqrt(area) / 224))
    clamped to the available levels (Lin et al., FPN, eq. 1), output `output_size`^2, `sampling_ratio`
    samples per bin side."""

    def __init__(self, featmap_names, output_size, sampling_ratio):
        super().__init__()
        self.featmap_names = featmap_names
        self.output_size = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)
        self.sampling_ratio = sampling_ratio

    def forward(self, features, boxes, image_shapes):
        feats = [features[k] for k in self.featmap_names if k in features]
        ids = torch.cat([torch.full((b.shape[0], 1), i, dtype=b.dtype, device=b.device) for i, b in enumerate(boxes)], dim=0)
        rois = torch.cat([ids, torch.cat(boxes, dim=0)], dim=1)
        max_h = max(s[0] for s in image_shapes)
        max_w = max(s[1] for s in image_shapes)
        scales = []
        for f in feats:
            # spatial scale = 2^round(log2(feature / image)) as in torchvision's infer_scale
            s = 2.0 ** round(math.log2(f.shape[-2] / float(max_h)))
            s2 = 2.0 ** round(math.log2(f.shape[-1] / float(max_w)))
            assert s == s2
            scales.append(s)
        P = self.output_size[0]
        if len(feats) == 1:
            return roi_align(feats[0], rois, scales[0], P, self.sampling_ratio)
        k_min, k_max = -math.log2(scales[0]), -math.log2(scales[-1])
        area = box_area(rois[:, 1:])
        lvl = torch.floor(4 + torch.log2(torch.sqrt(area) / 224) + 1e-6).clamp(min=k_min, max=k_max).long() - int(k_min)
        if len(feats) <= 4 and P <= 7 and all(_is_nhwc(f) for f in feats):
            # channels-last pyramid: every level in one launch, no per-level index_select / scatter / host sync
            # nan_to_num + clamp: boxes of a diverged step (NaN / inf area) must still name a real level
            lvl = torch.nan_to_num(torch.floor(4 + torch.log2(torch.sqrt(area) / 224) + 1e-6), nan=k_min, posinf=k_max, neginf=k_min)
            lvl = (lvl.clamp(min=k_min, max=k_max) - k_min).to(torch.int32)
            return _RoIAlignNHWC.apply(rois, lvl, scales, P, self.sampling_ratio, False, *feats)
        out = torch.zeros((rois.shape[0], feats[0].shape[1], P, P), dtype=feats[0].dtype, device=rois.device)
        for i, (f, s) in enumerate(zip(feats, scales)):
            idx = torch.where(lvl == i)[0]
            if idx.numel():
                out[idx] = roi_align(f, rois[idx], s, P, self.sampling_ratio).to(out.dtype)
        return out
