"""Inference-side layout handling around the planar 3x3 convolutions of ResNet layer2-4 (models/backbone.py: _planar_middle;
csrc/dib_eltwise.hip: dib_bias_act_transpose): the fused passes give the values of epilogue + copy, and the detector's trunk
gives bit-identical features with them on and off."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(1, 128, 100, 168), (2, 512, 25, 42), (1, 4, 3, 5), (3, 68, 7, 9), (1, 256, 50, 84)])
def test_bias_act_transpose_equals_epilogue_plus_copy(shape):
    from detectinblur_amd.models import backbone as B
    g = torch.Generator().manual_seed(sum(shape))
    bias = torch.randn(shape[1], generator=g).cuda()
    x = torch.randn(shape, generator=g).cuda()
    x[0, 0, 0, 0] = float("nan")
    for relu in (True, False):
        nhwc = x.contiguous(memory_format=torch.channels_last)
        want = B.bias_act(nhwc.clone(memory_format=torch.channels_last), bias, None, relu)
        got = B.bias_act_transpose(nhwc, bias, relu, True)
        assert got.is_contiguous() and torch.equal(got.nan_to_num(7.0), want.contiguous().nan_to_num(7.0))
        back = B.bias_act_transpose(x.contiguous(), bias, relu, False)
        assert back.is_contiguous(memory_format=torch.channels_last) and torch.equal(back.nan_to_num(7.0), want.nan_to_num(7.0))


@pytest.mark.parametrize("size", [(800, 1333), (800, 1088)])       # sizes the shipped find-db pins the kernels of
def test_trunk_features_are_identical_with_the_fused_layout_changes(size):
    from detectinblur_amd.models import backbone as B
    from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
    torch.manual_seed(0)
    m = fasterrcnn_resnet50_fpn(num_classes=91, pretrained=False, pretrained_backbone=False).cuda().eval()
    x = torch.rand(1, 3, *size, device="cuda").contiguous(memory_format=torch.channels_last)
    out = {}
    with torch.no_grad():
        for flag in (True, False):
            B.PLANAR_FUSED = flag
            try:
                out[flag] = m.backbone(x)
            finally:
                B.PLANAR_FUSED = True
    assert set(out[True]) == set(out[False])
    for k in out[True]:
        assert out[True][k].shape == out[False][k].shape and torch.equal(out[True][k], out[False][k]), k
    # the fused path was taken: layer3's middle convolutions cache a planar weight
    assert "_dib_fold_planar" in m.backbone.body.layer3[1].conv2.__dict__


def test_planar_weight_follows_a_weight_update():
    from detectinblur_amd.models import backbone as B
    from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
    torch.manual_seed(0)
    m = fasterrcnn_resnet50_fpn(num_classes=91, pretrained=False, pretrained_backbone=False).cuda().eval()
    x = torch.rand(1, 3, 800, 1088, device="cuda").contiguous(memory_format=torch.channels_last)
    conv = m.backbone.body.layer4[1].conv2
    with torch.no_grad():
        a = m.backbone(x)["3"].clone()
        ptr = conv.__dict__["_dib_fold_planar"][1].data_ptr()
        conv.weight.mul_(1.5)
        assert B.refresh_folded(m.backbone) >= 1
        b = m.backbone(x)["3"]
        assert conv.__dict__["_dib_fold_planar"][1].data_ptr() == ptr          # rewritten in place: a captured graph stays valid
        assert not torch.equal(a, b)
        B.PLANAR_FUSED = False
        try:
            c = m.backbone(x)["3"]
        finally:
            B.PLANAR_FUSED = True
        assert torch.equal(b, c)
