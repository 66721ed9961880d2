"""ResNet-50 + FPN trunk of `fasterrcnn_resnet50_fpn` (reference models/faster_rcnn.py:367 builds it
with torchvision's `resnet_fpn_backbone('resnet50', ...)`).  The convolutions and GEMMs are stock PyTorch-ROCm
(MIOpen / hipBLASLt, MFMA); what this file adds around them, each switchable for A/B runs and each checked against the
plain autograd graph (tests/test_detector_ops.py): the frozen batch-norm folded into the weights with a fused
bias + residual + ReLU epilogue, the ReLU backward from a sign mask, conv1 + skip (or + downsample) of a bottleneck as one
autograd node, and three shape-based detours where MIOpen's channels-last kernel is not the fast one (small-M 1x1 as
GEMM, layer4's 3x3 through the planar kernels, wide-output 1x1 data gradients as GEMM).

Layout matches torchvision's so that published checkpoints load unchanged:
`body.conv1 / body.bn1 / body.layer{1..4}.{i}.{conv,bn}{1,2,3} / downsample.{0,1}` and
`fpn.inner_blocks.{0..3} / fpn.layer_blocks.{0..3}`; batch-norm layers are frozen affine maps.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F
from torch import nn


class FrozenBatchNorm2d(nn.Module):
    """BatchNorm with fixed statistics and affine parameters: y = x * scale + shift."""

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        state_dict.pop(prefix + "num_batches_tracked", None)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def affine(self):
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        return scale, self.bias - self.running_mean * scale

    def forward(self, x):
        scale, shift = self.affine()
        return x * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)


# A frozen batch-norm behind a convolution is a per-output-channel affine map, so it folds into the
# convolution: conv(x, w) * s + t == conv(x, w * s) + t.  The fold is recomputed from the live weight
# every call (a few M elements), so gradients reach `conv.weight` unchanged in value up to fp32
# rounding, and two full-activation elementwise passes per layer (forward) and one (backward)
# disappear from the step.  Set to False to run the unfused module-by-module form.
FOLD_FROZEN_BN = True


class _BiasAct(torch.autograd.Function):
    """x = act(x + bias[c] (+ residual)) in place on a channels-last fp32 CUDA tensor, one HIP launch
    (include/dib.h: dib_bias_act_nhwc / dib_bias_act_mask_nhwc).  With ReLU the forward also writes the output's sign
    pattern (one byte per 4 elements) and the backward is one HIP pass over the gradient and that mask
    (dib_relu_mask_backward: 8.25 bytes per element where torch's threshold_backward on the saved output moves 12);
    the same gradient flows to x and to the residual; the bias gradient is a channel sum when asked for."""

    @staticmethod
    def forward(ctx, x, bias, residual, relu, state=None):
        from .. import _lib
        N, C, H, W = x.shape
        stream = _lib.stream_of(x)
        res = residual.data_ptr() if residual is not None else None
        ctx.relu, ctx.has_res, ctx.masked, ctx.state = bool(relu), residual is not None, False, state
        if (relu and RELU_MASK and any(ctx.needs_input_grad[:3]) and C % 4 == 0
                and not ((x.data_ptr() | bias.data_ptr() | (res or 0)) & 15)):
            mask = torch.empty(x.numel() // 4, dtype=torch.uint8, device=x.device)
            _lib.check(_lib.lib().dib_bias_act_mask_nhwc(x.data_ptr(), bias.data_ptr(), res, x.numel(), C, mask.data_ptr(), stream))
            ctx.masked = True
            ctx.save_for_backward(mask)
            if state is not None:
                state["mask"] = mask
        else:
            _lib.check(_lib.lib().dib_bias_act_nhwc(x.data_ptr(), bias.data_ptr(), res, x.numel(), C, int(relu), stream))
            if relu:
                ctx.save_for_backward(x)
        ctx.mark_dirty(x)
        return x

    @staticmethod
    def backward(ctx, grad):
        handed = ctx.state.get("masked_grad") if (ctx.masked and ctx.state is not None) else None
        if handed is not None and handed == (grad.data_ptr(), grad._version, tuple(grad.shape)):
            # this gradient IS the tensor the output's _BlockEntry returned, untouched since (autograd's in-place accumulation
            # of a second consumer's gradient would have bumped its version): the mask was applied while it was accumulated
            pass
        elif ctx.masked:
            from .. import _lib
            (mask,) = ctx.saved_tensors
            if not grad.is_contiguous(memory_format=torch.channels_last):
                grad = grad.contiguous(memory_format=torch.channels_last)      # the mask is in NHWC element order
            if grad.data_ptr() & 15:
                grad = grad.clone(memory_format=torch.channels_last)
            out = torch.empty_like(grad)     # not in place: autograd may hand the same gradient tensor to another node
            _lib.check(_lib.lib().dib_relu_mask_backward(grad.data_ptr(), mask.data_ptr(), out.data_ptr(), grad.numel(),
                                                         _lib.stream_of(grad)))
            grad = out
        elif ctx.relu:
            (y,) = ctx.saved_tensors
            grad = torch.ops.aten.threshold_backward(grad, y, 0)
        gb = grad.sum(dim=(0, 2, 3)) if ctx.needs_input_grad[1] else None
        return grad, gb, (grad if ctx.has_res else None), None, None


def bias_act(x, bias, residual=None, relu=True):
    """act(x + bias[:, None, None] (+ residual)); fused and in place for channels-last fp32 CUDA tensors
    fresh out of a convolution, plain torch ops otherwise."""
    fast = (FUSE_EPILOGUE and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] > 1
            and x.is_contiguous(memory_format=torch.channels_last) and bias.dtype == torch.float32
            and (residual is None or (residual.shape == x.shape and residual.dtype == torch.float32
                                      and residual.is_contiguous(memory_format=torch.channels_last))))
    if fast:
        state = {} if relu else None
        y = _BiasAct.apply(x, bias.contiguous(), residual, relu, state)
        if state:                       # the sign mask exists: a _BlockEntry consuming y may take over the ReLU backward
            y._dib_relu_state = state
        return y
    y = x + bias.reshape(1, -1, 1, 1)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


FUSE_EPILOGUE = True
RELU_MASK = True          # ReLU backward from the 1-byte-per-4 sign mask (False: torch's threshold_backward on the saved output)

# A 1x1 stride-1 convolution on a channels-last tensor IS a GEMM [N*H*W, Cin] x [Cin, Cout] on the same memory.  MIOpen's
# fp32 implicit-GEMM kernels win for the large-M shapes of the trunk; for the small-M, wide-channel ones (ResNet layer4 and
# the top FPN lateral at 25 x 42: M = 8,400 at b = 8) hipBLASLt's SGEMM is 1.5-1.7x faster forward + backward
# (scratch/t_conv1x1.py: 512<->2048 0.81 / 0.77 -> 0.47 / 0.46 ms, 2048->256 0.44 -> 0.29 ms; ~1.8 ms of a 118 ms step).
LINEAR_1X1 = True


def _as_gemm(x, conv, forward_only=False):
    """forward_only: the caller computes the gradients itself (the entry nodes below), so the rule may follow the forward
    direction alone: deep contractions (Cin >= 1024) also win there at M = 33,600 (1024 -> 256 at 50 x 84: 0.193 -> 0.145 ms),
    while their data gradient is faster in MIOpen (scratch/t_conv1x1_dirs.py)."""
    if not (LINEAR_1X1 and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.groups == 1
            and x.is_cuda and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)):
        return False
    M = x.shape[0] * x.shape[2] * x.shape[3]
    if M <= 16384 and conv.in_channels * conv.out_channels >= 512 * 1024:
        return True
    # forward alone (the entry nodes' forward pass; inference): the GEMM also wins at M <= 67,200 unless both channel
    # counts are small -- 1024 -> 256 at M = 33,600: 0.193 -> 0.145 ms (b = 8); at b = 1 every 1x1 of the trunk but
    # 64 -> 64: 2.28 -> 1.60 ms per image (scratch/t_b1_shapes.py)
    return (forward_only or not torch.is_grad_enabled()) and M <= 67200 and max(conv.in_channels, conv.out_channels) >= 128


# Inference only: the strided 1x1 "downsample" convolution of a stage's first block as a GEMM on the gathered pixels (what
# _DownEntry does in training).  MIOpen's kernels for these shapes at small M accumulate with atomics: two runs on the same
# input differ in the last bits (scratch/t_nondet.py, profiles/r4_nondeterminism.txt), which a replayed HIP graph must not.
STRIDED_1X1_GEMM = True


def _as_strided_gemm(x, conv):
    return (STRIDED_1X1_GEMM and not torch.is_grad_enabled() and conv.kernel_size == (1, 1) and conv.stride[0] == conv.stride[1]
            and conv.stride[0] > 1 and conv.padding == (0, 0) and conv.groups == 1 and x.is_cuda and x.dim() == 4
            and x.is_contiguous(memory_format=torch.channels_last) and max(conv.in_channels, conv.out_channels) >= 128)


# MIOpen's channels-last fp32 kernels are the fast ones for the large activations of this network; for the small-M, wide 3x3
# convolutions of ResNet layer4 (512 -> 512 at 25 x 42: M = 8,400 at b = 8) its planar kernels are 2.2x faster forward and 1.4x
# backward (scratch/t_conv3x3_layout.py: 0.91 / 0.57 ms -> 0.41 / 0.41 ms), far more than the two 17 MB layout changes cost.
NCHW_SMALL_3X3 = True


def _as_planar(x, conv):
    if not (NCHW_SMALL_3X3 and conv.kernel_size == (3, 3) and conv.groups == 1 and x.is_cuda and x.dim() == 4
            and x.is_contiguous(memory_format=torch.channels_last)):
        return False
    ho = (x.shape[2] + 2 * conv.padding[0] - 3) // conv.stride[0] + 1
    wo = (x.shape[3] + 2 * conv.padding[1] - 3) // conv.stride[1] + 1
    if conv.in_channels >= 512 and x.shape[0] * ho * wo <= 16384:
        return True
    # inference: at batch 1 the planar kernels win from 128 channels on wherever M <= 16,800 (256 -> 256 at 50 x 84:
    # 147 -> 88 us; 4.28 -> 3.33 ms per image over the trunk's 3x3 convolutions, scratch/t_b1_shapes.py)
    return not torch.is_grad_enabled() and conv.in_channels >= 128 and x.shape[0] * ho * wo <= 16800


class _Gemm1x1(torch.autograd.Function):
    """A 1x1 stride-1 convolution of a channels-last tensor as GEMMs on the same memory (hipBLASLt), returning a tensor of its
    OWN -- not the permuted view `F.linear(x.permute(...)).permute(...)` gives.  The difference matters to the in-place
    epilogue behind it: modifying a view in place makes autograd rebuild the gradient through CopySlices / AsStridedBackward,
    four full-size copies per convolution in the backward pass (12 x 69 MB per train step for layer4's three 512 -> 2048
    convolutions: 1 ms)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        N, Ci, H, W = x.shape
        Co = weight.shape[0]
        out = torch.empty((N, Co, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        xf, wf = x.permute(0, 2, 3, 1).reshape(-1, Ci), weight.reshape(Co, Ci)
        of = out.permute(0, 2, 3, 1).view(-1, Co)
        if bias is None:
            torch.mm(xf, wf.t(), out=of)
        else:
            torch.addmm(bias, xf, wf.t(), out=of)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        N, Ci, H, W = x.shape
        Co = weight.shape[0]
        if not g.is_contiguous(memory_format=torch.channels_last):
            g = g.contiguous(memory_format=torch.channels_last)
        gf = g.permute(0, 2, 3, 1).reshape(-1, Co)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.mm(gf, weight.reshape(Co, Ci)).view(N, H, W, Ci).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            dw = torch.mm(gf.t(), x.permute(0, 2, 3, 1).reshape(-1, Ci)).view(weight.shape)
            if dw.stride() != weight.stride():
                # a 1x1 kernel's [Co, Ci, 1, 1] memory is the same contiguous or channels-last: hand the gradient over with the
                # parameter's own strides, or DDP's bucket views (gradient_as_bucket_view) copy it instead of aliasing it
                dw = dw.as_strided(weight.shape, weight.stride())
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = gf.sum(0)
        return dx, dw, db


def conv1x1(x, weight, bias, conv):
    """F.conv2d for every convolution, with two shape-based detours: the same contraction through F.linear for 1x1 / stride 1 /
    small M / wide channels, and planar (NCHW) tensors around small-M wide 3x3 convolutions.  Output always channels-last
    when the input is."""
    if _as_gemm(x, conv):
        if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad)):
            return _Gemm1x1.apply(x, weight, bias)
        return F.linear(x.permute(0, 2, 3, 1), weight.reshape(conv.out_channels, conv.in_channels), bias).permute(0, 3, 1, 2)
    if _as_strided_gemm(x, conv):
        s = conv.stride[0]
        xs = x[:, :, ::s, ::s].contiguous(memory_format=torch.channels_last)
        return F.linear(xs.permute(0, 2, 3, 1), weight.reshape(conv.out_channels, conv.in_channels), bias).permute(0, 3, 1, 2)
    if _as_planar(x, conv):
        y = F.conv2d(x.contiguous(), weight.contiguous(), bias, conv.stride, conv.padding, conv.dilation, conv.groups)
        return y.contiguous(memory_format=torch.channels_last)
    return F.conv2d(x, weight, bias, conv.stride, conv.padding, conv.dilation, conv.groups)


def _conv1x1_base(x, weight, conv):
    """conv1x1 without autograd whose result is a tensor of its own (not a view of the GEMM's output): what a custom
    autograd node may hand to an in-place epilogue."""
    if _as_gemm(x, conv, forward_only=True):
        N, _, H, W = x.shape
        out = torch.empty((N, conv.out_channels, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        torch.mm(x.permute(0, 2, 3, 1).reshape(-1, conv.in_channels), weight.reshape(conv.out_channels, conv.in_channels).t(),
                 out=out.permute(0, 2, 3, 1).view(-1, conv.out_channels))
        return out
    return F.conv2d(x, weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)


# The input of an identity-skip bottleneck feeds its first convolution AND its skip connection: autograd would add the two
# gradients (a 12 B / element pass) and the ReLU behind that input would mask the sum in another pass (8.25 B).  _BlockEntry is
# that first 1x1 convolution plus the skip as ONE autograd node: its backward accumulates and masks in place on the fresh data
# gradient (dib_add_relu_mask, 12.25 B), and tells the producing _BiasAct WHICH gradient tensor (pointer + version) carries the
# mask already; the producer skips its own mask pass only for exactly that tensor, so a second consumer of the output (a hook,
# an auxiliary loss) costs a pass, never a wrong gradient.  Values identical to the unfused graph (the mask is idempotent and linear).
BLOCK_ENTRY = True


class _BlockEntry(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, state, conv):
        ctx.conv = conv
        mask = state.get("mask") if state is not None else None
        ctx.has_mask, ctx.state = mask is not None, (state if mask is not None else None)
        ctx.save_for_backward(x, weight, *([mask] if mask is not None else []))
        return _conv1x1_base(x, weight, conv), x.view_as(x)

    @staticmethod
    def backward(ctx, g_out, g_skip):
        from .. import _lib
        x, weight = ctx.saved_tensors[:2]
        mask = ctx.saved_tensors[2] if ctx.has_mask else None
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if not g_out.is_contiguous(memory_format=torch.channels_last):
            g_out = g_out.contiguous(memory_format=torch.channels_last)
        dx, dw, _ = torch.ops.aten.convolution_backward(g_out, x, weight, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1,
                                                        [need_x, need_w, False])
        if need_x:
            if not dx.is_contiguous(memory_format=torch.channels_last) or dx.data_ptr() & 15:
                dx = dx.clone(memory_format=torch.channels_last)
            if g_skip is not None:
                if not g_skip.is_contiguous(memory_format=torch.channels_last) or g_skip.data_ptr() & 15:
                    g_skip = g_skip.clone(memory_format=torch.channels_last)
                _lib.check(_lib.lib().dib_add_relu_mask(dx.data_ptr(), g_skip.data_ptr(), mask.data_ptr() if mask is not None else None,
                                                        dx.numel(), _lib.stream_of(dx)))
            elif mask is not None:
                _lib.check(_lib.lib().dib_relu_mask_backward(dx.data_ptr(), mask.data_ptr(), dx.data_ptr(), dx.numel(),
                                                             _lib.stream_of(dx)))
            if mask is not None:
                # tell the producing _BiasAct which gradient tensor already carries its ReLU mask; it re-applies the mask (idempotent
                # and linear, so always correct) to anything else -- e.g. the sum autograd forms when the output has a second consumer
                ctx.state["masked_grad"] = (dx.data_ptr(), dx._version, tuple(dx.shape))
        return (dx if need_x else None), (dw if need_w else None), None, None


class _DownEntry(torch.autograd.Function):
    """The first block of a stage: its input feeds conv1 (1x1, stride 1) and the downsample convolution (1x1, stride s).  One
    node: the strided convolution runs densely on the gathered pixels x[:, :, ::s, ::s]; in the backward pass its data
    gradient is added in place into conv1's (dib_scatter_add_nhwc) -- no zero-filled full-size gradient, no full-size add."""

    @staticmethod
    def forward(ctx, x, w1, wd, conv1, convd):
        s = convd.stride[0]
        xs = x if s == 1 else x[:, :, ::s, ::s].contiguous(memory_format=torch.channels_last)
        ctx.conv1, ctx.convd, ctx.s = conv1, convd, s
        ctx.save_for_backward(x, w1, wd, *([xs] if s != 1 else []))
        import types
        dense = types.SimpleNamespace(kernel_size=(1, 1), stride=(1, 1), padding=(0, 0), dilation=(1, 1), groups=1,
                                      in_channels=convd.in_channels, out_channels=convd.out_channels)
        return _conv1x1_base(x, w1, conv1), _conv1x1_base(xs, wd, dense)

    @staticmethod
    def backward(ctx, g_a, g_d):
        from .. import _lib
        x, w1, wd = ctx.saved_tensors[:3]
        xs = ctx.saved_tensors[3] if ctx.s != 1 else x
        need_x, need_w1, need_wd = ctx.needs_input_grad[:3]
        cl = lambda t: t if t.is_contiguous(memory_format=torch.channels_last) and not (t.data_ptr() & 15) else t.clone(memory_format=torch.channels_last)   # noqa: E731
        args = ([1, 1], [0, 0], [1, 1], False, [0, 0], 1)
        dx, dw1, _ = torch.ops.aten.convolution_backward(cl(g_a), x, w1, None, *args, [need_x, need_w1, False])
        dxs, dwd, _ = torch.ops.aten.convolution_backward(cl(g_d), xs, wd, None, *args, [need_x, need_wd, False])
        if need_x:
            dx, dxs = cl(dx), cl(dxs)
            stream = _lib.stream_of(dx)
            if ctx.s == 1:
                _lib.check(_lib.lib().dib_add_relu_mask(dx.data_ptr(), dxs.data_ptr(), None, dx.numel(), stream))
            else:
                N, C, H, W = dx.shape
                _lib.check(_lib.lib().dib_scatter_add_nhwc(dx.data_ptr(), dxs.data_ptr(), N, H, W, dxs.shape[2], dxs.shape[3], C, ctx.s, stream))
        return (dx if need_x else None), (dw1 if need_w1 else None), (dwd if need_wd else None), None, None


# Training re-folds every trunk convolution's frozen batch-norm each step (the weights move).  Per convolution that is five
# tiny launches forward and one backward, 53 times: ~320 launches of ~5 us each that the GPU sits through.  FOLD_ALL folds
# every trainable pair of the body in ONE launch per 32 at the start of its forward pass (dib_fold_bn_multi; backward:
# dib_scale_rows_multi), bit-identical to the per-convolution expressions (the same operations in the same order).
FOLD_ALL = True


class _FoldAll(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pairs, *weights):
        from .. import _lib
        n = len(weights)
        dev = weights[0].device
        cos = [int(w.shape[0]) for w in weights]
        inner = [int(w.numel() // w.shape[0]) for w in weights]
        wfs = [torch.empty_like(w) for w in weights]                     # preserve_format: the fold keeps the weight's layout
        flat = torch.empty(2 * sum(cos), dtype=torch.float32, device=dev)
        scales, shifts, off = [], [], 0
        for c in cos:
            scales.append(flat[off:off + c]); shifts.append(flat[off + c:off + 2 * c]); off += 2 * c
        P = _lib.ptr_array
        bns = [bn for _, bn in pairs]
        _lib.check(_lib.lib().dib_fold_bn_multi(
            P([w.data_ptr() for w in weights]), P([b.weight.data_ptr() for b in bns]), P([b.bias.data_ptr() for b in bns]),
            P([b.running_mean.data_ptr() for b in bns]), P([b.running_var.data_ptr() for b in bns]), _lib.int_array(cos),
            _lib.int_array(inner), n, float(bns[0].eps), P([t.data_ptr() for t in wfs]), P([t.data_ptr() for t in scales]),
            P([t.data_ptr() for t in shifts]), _lib.stream_of(wfs[0])))
        ctx.save_for_backward(flat)
        ctx.meta = (cos, inner)
        ctx.mark_non_differentiable(*shifts)
        return tuple(wfs) + tuple(shifts)

    @staticmethod
    def backward(ctx, *grads):
        from .. import _lib
        (flat,) = ctx.saved_tensors
        cos, inner = ctx.meta
        n = len(cos)
        gs, ks, off, offs = [], [], 0, []
        for k, c in enumerate(cos):
            offs.append(off)
            off += 2 * c
        dws = [None] * n
        for k in range(n):
            g = grads[k]
            if g is None or not ctx.needs_input_grad[1 + k]:
                continue
            # dense, output channel outermost (contiguous or channels-last), as the forward pass saw the weight
            if not (g.is_contiguous() or g.is_contiguous(memory_format=torch.channels_last)):
                g = g.contiguous()
            gs.append(g); ks.append(k)
        if ks:
            outs = [torch.empty_like(g) for g in gs]
            P = _lib.ptr_array
            _lib.check(_lib.lib().dib_scale_rows_multi(
                P([g.data_ptr() for g in gs]), P([flat.data_ptr() + 4 * offs[k] for k in ks]), _lib.int_array([cos[k] for k in ks]),
                _lib.int_array([inner[k] for k in ks]), len(ks), P([o.data_ptr() for o in outs]), _lib.stream_of(outs[0])))
            for k, o in zip(ks, outs):
                dws[k] = o
        return (None,) + tuple(dws)


def _begin_step_folds(body, x):
    """Fold every trainable (convolution, frozen batch-norm) pair of `body` at once and park the results on the convolutions
    for the duration of this forward pass; returns the list to hand to _end_step_folds (None: nothing parked)."""
    if not (FOLD_ALL and FOLD_FROZEN_BN and torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32):
        return None
    pairs = []
    for conv, bn in _fold_pairs(body):
        w = conv.weight
        if (w.requires_grad and isinstance(bn, FrozenBatchNorm2d) and conv.bias is None and w.is_cuda and w.dtype == torch.float32
                and (w.is_contiguous() or w.is_contiguous(memory_format=torch.channels_last)) and bn.weight.is_cuda):
            pairs.append((conv, bn))
    if not pairs or len({float(bn.eps) for _, bn in pairs}) != 1:
        return None
    outs = _FoldAll.apply(pairs, *[conv.weight for conv, _ in pairs])
    n = len(pairs)
    for k, (conv, _) in enumerate(pairs):
        conv.__dict__["_dib_step_fold"] = (outs[k], outs[n + k])
    return pairs


def _end_step_folds(pairs):
    for conv, _ in pairs or ():
        conv.__dict__.pop("_dib_step_fold", None)


def _train_fold(conv, bn):
    """(weight * scale, shift) of a trainable convolution + frozen batch-norm pair for THIS forward pass: the parked result of
    _begin_step_folds when there is one, the per-convolution expressions otherwise (same values)."""
    hit = conv.__dict__.get("_dib_step_fold")
    if hit is not None:
        return hit
    scale, shift = bn.affine()
    return conv.weight * scale.reshape(-1, 1, 1, 1), shift


def down_entry(x, conv1, bn1, convd, bnd):
    """(relu(bn1(conv1(x))), bnd(convd(x))) for the first block of a stage, or None where the fused node does not apply."""
    if not (BLOCK_ENTRY and FUSE_EPILOGUE and FOLD_FROZEN_BN and isinstance(bn1, FrozenBatchNorm2d) and isinstance(bnd, FrozenBatchNorm2d)
            and conv1.bias is None and convd.bias is None and conv1.kernel_size == (1, 1) and conv1.stride == (1, 1)
            and convd.kernel_size == (1, 1) and convd.stride[0] == convd.stride[1] and convd.stride[0] in (1, 2)
            and conv1.padding == (0, 0) and convd.padding == (0, 0) and conv1.groups == 1 and convd.groups == 1
            and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] % 4 == 0
            and x.is_contiguous(memory_format=torch.channels_last) and not (x.data_ptr() & 15) and torch.is_grad_enabled()
            and x.requires_grad):
        return None
    w1, t1 = _train_fold(conv1, bn1)
    wd, td = _train_fold(convd, bnd)
    a, d = _DownEntry.apply(x, w1, wd, conv1, convd)
    return bias_act(a, t1, None, True), bias_act(d, td, None, False)


def block_entry(x, conv, bn):
    """(relu(bn(conv(x))), x) for the first convolution of an identity-skip bottleneck, or None where the fused node does not
    apply (CPU, planar tensors, odd channel counts, a batch-norm that is not frozen)."""
    if not (BLOCK_ENTRY and FUSE_EPILOGUE and FOLD_FROZEN_BN and isinstance(bn, FrozenBatchNorm2d) and conv.bias is None
            and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.groups == 1
            and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] % 4 == 0
            and x.is_contiguous(memory_format=torch.channels_last) and not (x.data_ptr() & 15) and torch.is_grad_enabled()
            and (x.requires_grad or conv.weight.requires_grad)):
        return None
    weight, shift = _train_fold(conv, bn)
    out, skip = _BlockEntry.apply(x, weight, getattr(x, "_dib_relu_state", None), conv)
    return bias_act(out, shift, None, True), skip


def _fold_key(conv, bn):
    return (conv.weight.data_ptr(), conv.weight._version, bn.weight._version, bn.bias._version, bn.running_mean._version,
            bn.running_var._version, bn.weight.data_ptr())


def _folded(conv, bn):
    """(weight * scale, shift) of a convolution + frozen batch-norm pair for inference, cached while neither changes (the fold
    is 5 tiny launches per convolution, ~1 ms of GPU time per image at batch 1, for values that never move).  The two result
    tensors are allocated ONCE per pair and from then on REWRITTEN IN PLACE when a weight or statistic changed: a HIP graph
    captured over them (graphs.py) keeps valid pointers and reads current values once `refresh_folded` has run."""
    key = _fold_key(conv, bn)
    hit = conv.__dict__.get("_dib_fold")
    if hit is None or hit[0] != key:
        with torch.no_grad():
            scale, shift = bn.affine()
            w = conv.weight * scale.reshape(-1, 1, 1, 1)
            if conv.weight.is_contiguous(memory_format=torch.channels_last):
                w = w.contiguous(memory_format=torch.channels_last)
            if (hit is not None and hit[1].shape == w.shape and hit[1].stride() == w.stride() and hit[1].device == w.device
                    and hit[1].dtype == w.dtype):
                hit[1].copy_(w)
                hit[2].copy_(shift)
                hit = (key, hit[1], hit[2])
            else:
                hit = (key, w, shift.contiguous())
        conv.__dict__["_dib_fold"] = hit
    return hit[1], hit[2]


def _folded_planar(conv, bn):
    """`_folded` with the weight also as a planar (contiguous) tensor, for the convolutions inference runs through MIOpen's planar
    kernels: cached beside the fold and rewritten in place with it (a captured graph keeps reading the same memory)."""
    weight, shift = _folded(conv, bn)
    key = conv.__dict__["_dib_fold"][0]
    hit = conv.__dict__.get("_dib_fold_planar")
    if hit is None or hit[0] != key:
        with torch.no_grad():
            if hit is not None and hit[1].shape == weight.shape and hit[1].device == weight.device:
                hit[1].copy_(weight)
                hit = (key, hit[1])
            else:
                hit = (key, weight.contiguous().clone() if weight.is_contiguous() else weight.contiguous())
        conv.__dict__["_dib_fold_planar"] = hit
    return hit[1], shift


def refresh_folded(module):
    """Bring every cached fold under `module` up to date, in place (see _folded).  Returns the number of folds rewritten.
    Cheap when nothing changed: seven attribute reads per convolution."""
    # fast path (every replay of a captured trunk calls this): tensor versions only grow, so an unchanged sum over the folds'
    # inputs means no in-place update since the last call; replaced storage is the caller's pointer check
    # (GeneralizedRCNN._sync_graphs_with_weights) or shows up as a missing cache entry below
    pairs = _fold_pairs(module)
    watch = module.__dict__.get("_dib_fold_watch")
    if watch is None:
        watch = module.__dict__["_dib_fold_watch"] = [t for conv, bn in pairs
                                                       for t in (conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var)]
    state = (sum(t._version for t in watch), sum(1 for conv, _ in pairs if "_dib_fold" in conv.__dict__))
    if module.__dict__.get("_dib_fold_state") == state:
        return 0
    n = 0
    for conv, bn in pairs:
        hit = conv.__dict__.get("_dib_fold")
        if hit is not None and hit[0] != _fold_key(conv, bn):
            _folded(conv, bn)
            if "_dib_fold_planar" in conv.__dict__:
                _folded_planar(conv, bn)
            n += 1
    module.__dict__["_dib_fold_state"] = (sum(t._version for t in watch), sum(1 for conv, _ in pairs if "_dib_fold" in conv.__dict__))
    return n


def _fold_pairs(module):
    pairs = module.__dict__.get("_dib_fold_pairs")
    if pairs is None:
        pairs = []
        for m in module.modules():
            if isinstance(m, Bottleneck):
                pairs += [(m.conv1, m.bn1), (m.conv2, m.bn2), (m.conv3, m.bn3)]
                if m.downsample is not None:
                    pairs.append((m.downsample[0], m.downsample[1]))
            elif isinstance(m, ResNet50Body):
                pairs.append((m.conv1, m.bn1))
        module.__dict__["_dib_fold_pairs"] = pairs
    return pairs


class _WideOut1x1(torch.autograd.Function):
    """A 1x1 stride-1 convolution with few input and many output channels (ResNet conv3: 256 -> 1024 at 50 x 84): forward and
    weight gradient through MIOpen, the DATA gradient -- a contraction over the 1024 output channels -- as a hipBLASLt GEMM
    on the NHWC view (0.187 -> 0.143 ms; scratch/t_conv1x1_dirs.py)."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return F.conv2d(x, weight)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        if not g.is_contiguous(memory_format=torch.channels_last):
            g = g.contiguous(memory_format=torch.channels_last)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            N, Co, H, W = g.shape
            dx = torch.mm(g.permute(0, 2, 3, 1).reshape(-1, Co), weight.reshape(Co, -1)).view(N, H, W, -1).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            dw = torch.ops.aten.convolution_backward(g, x, weight, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        return dx, dw


def _wide_out(x, conv):
    return (LINEAR_1X1 and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.groups == 1
            and conv.out_channels >= 1024 and conv.in_channels <= 256 and x.is_cuda and x.dim() == 4 and torch.is_grad_enabled()
            and x.requires_grad and x.is_contiguous(memory_format=torch.channels_last)
            and 16384 < x.shape[0] * x.shape[2] * x.shape[3] <= 65536)


def conv_bn(x, conv, bn, relu=False, residual=None):
    """conv -> frozen batch-norm (-> + residual) (-> ReLU).  With FOLD_FROZEN_BN the norm's scale goes
    into the weights and its shift into the fused epilogue."""
    if FOLD_FROZEN_BN and isinstance(bn, FrozenBatchNorm2d) and conv.bias is None:
        if torch.is_grad_enabled() and conv.weight.requires_grad:
            weight, shift = _train_fold(conv, bn)
        else:
            weight, shift = _folded(conv, bn)
        y = _WideOut1x1.apply(x, weight) if _wide_out(x, conv) else conv1x1(x, weight, None, conv)
        return bias_act(y, shift, residual, relu)
    y = bn(conv(x))
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


PLANAR_FUSED = True      # inference: the layout changes around planar 3x3 convolutions ride on the neighbouring epilogues


def bias_act_transpose(x, bias, relu, to_planar):
    """act(x + bias) of a channels-last tensor written planar (to_planar) or of a planar tensor written channels-last: one pass
    (csrc/dib_eltwise.hip) for what bias_act + .contiguous(...) do in two."""
    from .. import _lib
    N, C, H, W = x.shape
    out = torch.empty((N, C, H, W), dtype=x.dtype, device=x.device,
                      memory_format=torch.contiguous_format if to_planar else torch.channels_last)
    _lib.check(_lib.lib().dib_bias_act_transpose(x.data_ptr(), bias.data_ptr(), out.data_ptr(), N, C, H * W, int(to_planar), int(relu),
                                                 _lib.stream_of(x)))
    return out


def _planar_middle(y1, t1, conv2, bn2):
    """relu(bn2(conv2(relu(y1 + t1)))) with conv2 through MIOpen's planar kernels and both layout changes fused into the epilogues;
    None where that does not apply (training, CPU, shapes the channels-last kernels win)."""
    if not (PLANAR_FUSED and FUSE_EPILOGUE and FOLD_FROZEN_BN and not torch.is_grad_enabled() and isinstance(bn2, FrozenBatchNorm2d)
            and conv2.bias is None and y1.dtype == torch.float32 and _as_planar(y1, conv2) and conv2.dilation == (1, 1)):
        return None
    w2, t2 = _folded_planar(conv2, bn2)
    p = bias_act_transpose(y1, t1, True, True)
    y2 = F.conv2d(p, w2, None, conv2.stride, conv2.padding, conv2.dilation, conv2.groups)
    return bias_act_transpose(y2.contiguous(), t2, True, False)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, norm_layer=FrozenBatchNorm2d):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)   # stride on the 3x3 (ResNet v1.5)
        self.bn2 = norm_layer(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = norm_layer(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        entry = (block_entry(x, self.conv1, self.bn1) if self.downsample is None
                 else down_entry(x, self.conv1, self.bn1, self.downsample[0], self.downsample[1]))
        if entry is not None:
            out, idt = entry
        else:
            idt = x if self.downsample is None else conv_bn(x, self.downsample[0], self.downsample[1])
            mid = None
            if (PLANAR_FUSED and not torch.is_grad_enabled() and FOLD_FROZEN_BN and isinstance(self.bn1, FrozenBatchNorm2d)
                    and self.conv1.bias is None and x.is_cuda):
                w1, t1 = _folded(self.conv1, self.bn1)
                y1 = conv1x1(x, w1, None, self.conv1)
                mid = _planar_middle(y1, t1, self.conv2, self.bn2)
                out = mid if mid is not None else conv_bn(bias_act(y1, t1, None, True), self.conv2, self.bn2, relu=True)
                return conv_bn(out, self.conv3, self.bn3, relu=True, residual=idt)
            out = conv_bn(x, self.conv1, self.bn1, relu=True)
        out = conv_bn(out, self.conv2, self.bn2, relu=True)
        return conv_bn(out, self.conv3, self.bn3, relu=True, residual=idt)


FUSE_STEM_POOL = True     # stem: folded conv1 -> (bias + ReLU + 3x3/2 max-pool) as one pass


class _StemPool(torch.autograd.Function):
    """max_pool2d(relu(x + bias), 3, stride 2, padding 1) of a channels-last convolution output in one pass
    (dib_stem_pool_forward); the backward pass rebuilds the dense gradient of x from the pooled gradient and 4 bits per pooled
    element (dib_stem_pool_backward) -- ATen's max-pool backward and the ReLU backward in one."""

    @staticmethod
    def forward(ctx, x, bias):
        from .. import _lib
        N, C, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        out = torch.empty((N, C, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        arg = torch.empty((N * Ho * Wo * (C // 4),), dtype=torch.int16, device=x.device)
        _lib.check(_lib.lib().dib_stem_pool_forward(x.data_ptr(), bias.data_ptr(), out.data_ptr(), arg.data_ptr(), N, H, W, C,
                                                    _lib.stream_of(x)))
        ctx.shape = (N, C, H, W)
        ctx.save_for_backward(arg)
        return out

    @staticmethod
    def backward(ctx, g):
        from .. import _lib
        (arg,) = ctx.saved_tensors
        N, C, H, W = ctx.shape
        if not g.is_contiguous(memory_format=torch.channels_last) or (g.data_ptr() & 15):
            g = g.clone(memory_format=torch.channels_last)
        gx = torch.empty((N, C, H, W), dtype=g.dtype, device=g.device, memory_format=torch.channels_last)
        _lib.check(_lib.lib().dib_stem_pool_backward(g.data_ptr(), arg.data_ptr(), gx.data_ptr(), N, H, W, C,
                                                     _lib.stream_of(g)))
        return gx, (gx.sum(dim=(0, 2, 3)) if ctx.needs_input_grad[1] else None)


def stem(x, conv, bn):
    """max_pool2d(relu(bn(conv(x))), 3, stride=2, padding=1): the ResNet stem."""
    if (FUSE_STEM_POOL and FUSE_EPILOGUE and FOLD_FROZEN_BN and isinstance(bn, FrozenBatchNorm2d) and conv.bias is None and x.is_cuda
            and x.dtype == torch.float32 and conv.out_channels % 4 == 0):
        if torch.is_grad_enabled() and conv.weight.requires_grad:
            weight, shift = _train_fold(conv, bn)
        else:
            weight, shift = _folded(conv, bn)
        y = conv1x1(x, weight, None, conv)
        if y.is_contiguous(memory_format=torch.channels_last) and not (y.data_ptr() & 15):
            return _StemPool.apply(y, shift.contiguous())
        return F.max_pool2d(bias_act(y, shift, None, True), 3, stride=2, padding=1)
    return F.max_pool2d(conv_bn(x, conv, bn, relu=True), 3, stride=2, padding=1)


class ResNet50Body(nn.Module):
    def __init__(self, norm_layer=FrozenBatchNorm2d):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(64)
        self.inplanes = 64
        self.layer1 = self._stage(64, 3, 1, norm_layer)
        self.layer2 = self._stage(128, 4, 2, norm_layer)
        self.layer3 = self._stage(256, 6, 2, norm_layer)
        self.layer4 = self._stage(512, 3, 2, norm_layer)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _stage(self, planes, blocks, stride, norm_layer):
        down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False), norm_layer(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, down, norm_layer)]
        self.inplanes = planes * 4
        layers += [Bottleneck(self.inplanes, planes, norm_layer=norm_layer) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        parked = _begin_step_folds(self, x)
        try:
            x = stem(x, self.conv1, self.bn1)
            c2 = self.layer1(x)
            c3 = self.layer2(c2)
            c4 = self.layer3(c3)
            c5 = self.layer4(c4)
        finally:
            _end_step_folds(parked)
        return [c2, c3, c4, c5]


FUSE_TOPDOWN = True       # FPN: lateral bias + nearest upsample of the level above + add as one in-place pass


class _TopDownMerge(torch.autograd.Function):
    """inner = (lateral conv output + bias) + F.interpolate(top, size=lateral.shape[-2:], mode="nearest"), in place on the
    convolution's output (dib_fpn_topdown_merge_nhwc).  Backward: the lateral's gradient IS the incoming one, the bias
    gradient its sum over pixels, the upper level's ATen's own nearest-upsample backward -- what autograd computes for the
    unfused graph."""

    @staticmethod
    def forward(ctx, x, bias, top):
        from .. import _lib
        N, C, H, W = x.shape
        ctx.top_shape = tuple(top.shape)
        ctx.mark_dirty(x)
        _lib.check(_lib.lib().dib_fpn_topdown_merge_nhwc(x.data_ptr(), bias.data_ptr(), top.data_ptr(), N, H, W, top.shape[2], top.shape[3], C,
                                                         _lib.stream_of(x)))
        return x

    @staticmethod
    def backward(ctx, g):
        need_x, need_b, need_t = ctx.needs_input_grad
        g_b = g.sum((0, 2, 3)) if need_b else None
        g_t = None
        if need_t:
            g_t = torch.ops.aten.upsample_nearest2d_backward(g, list(g.shape[-2:]), list(ctx.top_shape), None, None)
        return (g if need_x else None), g_b, g_t


def topdown_merge(lateral, bias, top):
    """lateral (a convolution output WITHOUT its bias) + bias + nearest-upsampled `top`."""
    cl = lambda t: t.is_contiguous(memory_format=torch.channels_last) and not (t.data_ptr() & 15)   # noqa: E731
    if (FUSE_TOPDOWN and FUSE_EPILOGUE and lateral.is_cuda and lateral.dtype == torch.float32 and top.dtype == torch.float32
            and lateral.dim() == 4 and lateral.shape[1] % 4 == 0 and lateral.shape[:2] == top.shape[:2] and cl(lateral) and cl(top)
            and bias.dtype == torch.float32):
        return _TopDownMerge.apply(lateral, bias.contiguous(), top)
    return lateral + bias.reshape(1, -1, 1, 1) + F.interpolate(top, size=lateral.shape[-2:], mode="nearest")


class FeaturePyramidNetwork(nn.Module):
    """Top-down pathway with lateral 1x1 and output 3x3 convolutions, plus a stride-2 max-pool level."""

    def __init__(self, in_channels_list, out_channels):
        super().__init__()
        self.inner_blocks = nn.ModuleList([nn.Conv2d(c, out_channels, 1) for c in in_channels_list])
        self.layer_blocks = nn.ModuleList([nn.Conv2d(out_channels, out_channels, 3, padding=1) for _ in in_channels_list])
        for m in self.children():
            for conv in m:
                nn.init.kaiming_uniform_(conv.weight, a=1)
                nn.init.constant_(conv.bias, 0)

    def forward(self, feats):
        lat = lambda i: conv1x1(feats[i], self.inner_blocks[i].weight, self.inner_blocks[i].bias, self.inner_blocks[i])  # noqa: E731
        last = lat(-1)
        out = lambda i, t: conv1x1(t, self.layer_blocks[i].weight, self.layer_blocks[i].bias, self.layer_blocks[i])  # noqa: E731
        outs = [out(-1, last)]
        for i in range(len(feats) - 2, -1, -1):
            blk = self.inner_blocks[i]
            last = topdown_merge(conv1x1(feats[i], blk.weight, None, blk), blk.bias, last)
            outs.insert(0, out(i, last))
        outs.append(F.max_pool2d(outs[-1], 1, 2, 0))
        return OrderedDict(zip(["0", "1", "2", "3", "pool"], outs))


class BackboneWithFPN(nn.Module):
    def __init__(self, trainable_layers=5, out_channels=256):
        super().__init__()
        self.body = ResNet50Body()
        # freeze everything outside the last `trainable_layers` of (layer4, layer3, layer2, layer1, conv1)
        keep = ["layer4", "layer3", "layer2", "layer1", "conv1"][:trainable_layers]
        for name, p in self.body.named_parameters():
            if not any(name.startswith(k) for k in keep):
                p.requires_grad_(False)
        self.fpn = FeaturePyramidNetwork([256, 512, 1024, 2048], out_channels)
        self.out_channels = out_channels

    def forward(self, x):
        return self.fpn(self.body(x))


def resnet_fpn_backbone(backbone_name="resnet50", pretrained=False, trainable_layers=3):
    if backbone_name != "resnet50":
        raise ValueError("only resnet50 is built here (SURVEY.md section 2: other trunks are out of scope)")
    if pretrained:
        raise RuntimeError("no network access: load ImageNet weights with model.backbone.body.load_state_dict(...)")
    return BackboneWithFPN(trainable_layers)
