"""The fused blur epilogue (csrc/dib_epilogue.hip: float conversion + per-image normalisation + zero-padded batch in
one launch) against the module-by-module path it replaces -- which is pinned to the reference by
tests/test_net_transforms.py.  Bit-identical for fp16 and fp32 images, planar and channels-last, ragged sizes,
custom per-image statistics; the training-mode generator draw is consumed exactly once per image either way."""
import numpy as np
import pytest
import torch

from detectinblur_amd import blur_ops
from detectinblur_amd.models.net_transforms import GeneralizedRCNNTransform

pytestmark = pytest.mark.gpu
MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def _batch(dtype, sizes, seed=0):
    g = torch.Generator().manual_seed(seed)
    return [torch.rand(3, h, w, generator=g).to(dtype).cuda() for h, w in sizes]


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("channels_last", [False, True])
def test_fused_equals_unfused_bit_for_bit(dtype, channels_last):
    sizes = [(64, 100), (64, 100), (64, 100)]        # min side 64, max side 100: scale factor exactly 1
    imgs = _batch(dtype, sizes)
    means = np.array([MEAN, MEAN, [0.5, 0.4, 0.3]])
    stds = np.array([STD, [0.11716, 0.11548, 0.11734], [0.2, 0.25, 0.3]])
    tg = [{"boxes": torch.tensor([[1.0, 2.0, 30.0, 40.0]]).cuda()} for _ in imgs]
    out = {}
    for fused in (True, False):
        t = GeneralizedRCNNTransform(64, 100, MEAN, STD, training=True)
        t.fused, t.channels_last = fused, channels_last
        torch.manual_seed(3)
        il, res = t([i.clone() for i in imgs], [dict(d) for d in tg], newMeans=means, newSTDs=stds)
        out[fused] = (il.tensors, il.image_sizes, [d["boxes"] for d in res], torch.rand(1).item())
    a, b = out[True], out[False]
    assert a[0].dtype == torch.float32 and tuple(a[0].shape) == (3, 3, 64, 128)
    assert a[0].is_contiguous(memory_format=torch.channels_last if channels_last else torch.contiguous_format)
    assert torch.equal(a[0], b[0]) and a[1] == b[1] and a[3] == b[3]          # values, sizes, generator state
    assert all(torch.equal(x, y) for x, y in zip(a[2], b[2]))
    assert float(a[0][:, :, :, 100:].abs().max()) == 0.0                       # the padding is zero


def test_ragged_batch_and_default_statistics():
    imgs = [torch.rand(3, 40, 70).half().cuda(), torch.rand(3, 64, 33).half().cuda(), torch.rand(3, 1, 1).half().cuda()]
    got = blur_ops.normalize_pad(imgs, [MEAN] * 3, [STD] * 3, 64, 96, channels_last=True)
    m = torch.tensor(MEAN).cuda()[:, None, None]
    s = torch.tensor(STD).cuda()[:, None, None]
    want = torch.zeros(3, 3, 64, 96, device="cuda")
    for k, i in enumerate(imgs):
        want[k, :, :i.shape[1], :i.shape[2]] = (i.float() - m) / s
    assert torch.equal(got, want)
    with pytest.raises(Exception, match="larger than the batch"):
        blur_ops.normalize_pad(imgs, [MEAN] * 3, [STD] * 3, 32, 96)


COCO_SIZES = [(480, 640), (427, 640), (640, 480), (375, 500), (333, 500), (640, 428), (500, 375), (612, 612)]


def _contraction(on):
    import ctypes
    from detectinblur_amd import _lib
    l = _lib.lib()
    l.dib_debug_set_resize_contraction.argtypes = [ctypes.c_int]
    l.dib_debug_set_resize_contraction.restype = None
    l.dib_debug_set_resize_contraction(int(on))


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("channels_last", [False, True])
@pytest.mark.parametrize("training", [False, True])
def test_fused_resize_equals_unfused_bit_for_bit(dtype, channels_last, training):
    """Native-size COCO batches (the reference blurs before the model resizes, engine.py:101): float + normalise + bilinear
    resize (min side 800 / max side 1333, recompute_scale_factor=True, align_corners=False) + zero-padded batch in ONE launch
    (dib_normalize_resize_pad) against the module-by-module path -- which IS torch's own interpolate on the GPU: equal bit
    for bit, sizes, boxes and generator state included; a multi-scale training transform draws one size per image."""
    imgs = _batch(dtype, COCO_SIZES, seed=4)
    n = len(imgs)
    means = np.array([MEAN] * (n - 1) + [[0.5, 0.4, 0.3]])
    stds = np.array([STD] * (n - 2) + [[0.11716, 0.11548, 0.11734], [0.2, 0.25, 0.3]])
    tg = [{"boxes": torch.tensor([[1.0, 2.0, 300.0, 240.0], [50.5, 60.25, 200.0, 333.0]]).cuda()} for _ in imgs]
    out = {}
    for fused in (True, False):
        t = GeneralizedRCNNTransform((640, 704, 800) if training else 800, 1333, MEAN, STD, training=training)
        t.fused, t.channels_last = fused, channels_last
        torch.manual_seed(3)
        il, res = t([i.clone() for i in imgs], [dict(d) for d in tg], newMeans=means, newSTDs=stds)
        out[fused] = (il.tensors, il.image_sizes, [d["boxes"] for d in res], torch.rand(1).item())
    a, b = out[True], out[False]
    assert a[0].dtype == torch.float32 and a[0].shape == b[0].shape and a[0].shape[2] % 32 == 0 and a[0].shape[3] % 32 == 0
    assert a[0].is_contiguous(memory_format=torch.channels_last if channels_last else torch.contiguous_format)
    assert a[1] == b[1] and a[3] == b[3]                                         # sizes, generator state
    if not training:
        assert a[1][0] == (800, 1066) and a[1][2] == (1066, 800) and a[1][7] == (800, 800)
    assert torch.equal(a[0], b[0])
    assert all(torch.equal(x, y) for x, y in zip(a[2], b[2]))
    for k, (h, w) in enumerate(a[1]):                                            # the padding is zero
        assert float(a[0][k, :, h:, :].abs().max() if h < a[0].shape[2] else 0.0) == 0.0
        assert float(a[0][k, :, :, w:].abs().max() if w < a[0].shape[3] else 0.0) == 0.0


def test_fused_resize_downscale_mixed_with_unit_scale_and_tiny_images():
    """max-side-limited downscale (1500 x 2600 -> 769 x 1333), an image already at scale 1 in the same batch (not
    interpolated), 1-pixel-wide and 2 x 2 images (the h1p / w1p edge clamps)."""
    imgs = _batch(torch.float16, [(1500, 2600), (800, 1333), (3, 1), (2, 2), (1, 5)], seed=9)
    out = {}
    for fused in (True, False):
        t = GeneralizedRCNNTransform(800, 1333, MEAN, STD, training=False)
        t.fused, t.channels_last = fused, True
        il, _ = t([i.clone() for i in imgs], None)
        out[fused] = il
    assert out[True].image_sizes == out[False].image_sizes and out[True].image_sizes[0] == (769, 1333) and out[True].image_sizes[1] == (800, 1333)
    assert torch.equal(out[True].tensors, out[False].tensors)


def test_the_uncontracted_variant_is_not_what_aten_computes():
    """The kernel restates ATen's bilinear arithmetic WITH hipcc's default contraction (a * b + c * d = fma(a, b, c * d)); the
    plain expression (what the CPU computes) differs from the eager GPU path: the source index `scale * (x + 0.5) - 0.5` is
    rounded once instead of twice, which moves the interpolation weight by an ulp of the INDEX (6e-5 at x ~ 1000), i.e. the
    result by ~1e-4 of the local contrast at full size -- evidence that the contraction pattern matters and that the shipped
    one is ATen's.  (The reference-generated goldens of tests/test_net_transforms.py are 64..100 pixels wide: 2e-6 there.)"""
    imgs = _batch(torch.float32, COCO_SIZES[:3], seed=4)
    t = GeneralizedRCNNTransform(800, 1333, MEAN, STD, training=False)
    t.fused = False
    want = t([i.clone() for i in imgs], None)[0].tensors
    t.fused = True
    try:
        _contraction(False)
        plain = t([i.clone() for i in imgs], None)[0].tensors
    finally:
        _contraction(True)
    fused = t([i.clone() for i in imgs], None)[0].tensors
    assert torch.equal(fused, want)
    assert not torch.equal(plain, want) and float((plain - want).abs().max()) <= 1e-3


def test_train_step_is_identical_with_and_without_the_fused_epilogue():
    """engine._to_float hands Half images to the model when its transform fuses: the losses must not change."""
    from detectinblur_amd import engine
    from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
    torch.manual_seed(0)
    m = fasterrcnn_resnet50_fpn(pretrained=False, pretrained_backbone=False, num_classes=91, min_size=64, max_size=96,
                                rpn_pre_nms_top_n_train=200, rpn_post_nms_top_n_train=100, box_batch_size_per_image=32).cuda()
    m.train()
    imgs = _batch(torch.float16, [(64, 96), (64, 96)], seed=5)
    tg = [{"boxes": torch.tensor([[5.0, 6.0, 50.0, 40.0]]).cuda(), "labels": torch.tensor([3]).cuda()} for _ in imgs]
    means, stds = np.array([MEAN, MEAN]), np.array([STD, STD])
    losses = {}
    for fused in (True, False):
        m.transform.fused = fused
        x = engine._to_float(list(imgs), m, torch.device("cuda"))
        assert (x[0].dtype == torch.float16) == fused
        torch.manual_seed(11)
        ld = m(x, [dict(t) for t in tg], newMeans=means, newSTDs=stds)
        losses[fused] = torch.stack([ld[k] for k in sorted(ld)])
    # the transform outputs are bit-identical (tests above); the convolutions behind them are not run-to-run
    # deterministic on MIOpen, so the losses are compared to fp32 noise level
    assert torch.allclose(losses[True], losses[False], rtol=1e-4, atol=1e-6)
