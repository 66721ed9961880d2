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


def test_resize_needed_falls_back_to_the_unfused_path():
    imgs = _batch(torch.float16, [(50, 70), (60, 45)])
    t = GeneralizedRCNNTransform(64, 100, MEAN, STD, training=False)
    il, _ = t(imgs, None)
    u = GeneralizedRCNNTransform(64, 100, MEAN, STD, training=False)
    u.fused = False
    il2, _ = u([i.float() for i in imgs], None)
    assert torch.equal(il.tensors, il2.tensors) and il.image_sizes == il2.image_sizes


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
