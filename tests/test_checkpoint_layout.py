"""The checkpoint contract: reference evaluate.py:181-205 / train.py:230-236 load `checkpoint['model']` of detectors the
reference trained with torchvision's `fasterrcnn_resnet50_fpn` (0.7 / 0.8 era: FrozenBatchNorm2d with four buffers, no
`num_batches_tracked`; `backbone.body.*`, `backbone.fpn.{inner,layer}_blocks.N.*`, `rpn.head.{conv,cls_logits,bbox_pred}.*`,
`roi_heads.box_head.fc{6,7}.*`, `roi_heads.box_predictor.{cls_score,bbox_pred}.*`).  The list of keys and shapes below is
written out from torchvision's published architecture -- NOT read from this repo's model -- and the model must carry exactly
it, in order; checkpoints in that layout (also with the `num_batches_tracked` entries newer BatchNorm checkpoints carry) must
load strictly, and the RPN head -- which runs its two predictors as ONE concatenated, padded convolution on the GPU -- must
compute with the loaded tensors."""
import numpy as np
import pytest
import torch


def torchvision_fasterrcnn_resnet50_fpn_layout(num_classes=91):
    """[(key, shape)] of torchvision.models.detection.fasterrcnn_resnet50_fpn(num_classes).state_dict(), torchvision 0.7 / 0.8."""
    out = []

    def bn(prefix, c):
        for k in ("weight", "bias", "running_mean", "running_var"):
            out.append(("%s.%s" % (prefix, k), (c,)))

    out.append(("backbone.body.conv1.weight", (64, 3, 7, 7)))
    bn("backbone.body.bn1", 64)
    inplanes = 64
    for li, (planes, blocks) in enumerate(((64, 3), (128, 4), (256, 6), (512, 3)), start=1):
        for b in range(blocks):
            p = "backbone.body.layer%d.%d" % (li, b)
            out.append((p + ".conv1.weight", (planes, inplanes, 1, 1))); bn(p + ".bn1", planes)
            out.append((p + ".conv2.weight", (planes, planes, 3, 3))); bn(p + ".bn2", planes)
            out.append((p + ".conv3.weight", (planes * 4, planes, 1, 1))); bn(p + ".bn3", planes * 4)
            if b == 0:
                out.append((p + ".downsample.0.weight", (planes * 4, inplanes, 1, 1))); bn(p + ".downsample.1", planes * 4)
            inplanes = planes * 4
    for i, c in enumerate((256, 512, 1024, 2048)):
        out.append(("backbone.fpn.inner_blocks.%d.weight" % i, (256, c, 1, 1)))
        out.append(("backbone.fpn.inner_blocks.%d.bias" % i, (256,)))
    for i in range(4):
        out.append(("backbone.fpn.layer_blocks.%d.weight" % i, (256, 256, 3, 3)))
        out.append(("backbone.fpn.layer_blocks.%d.bias" % i, (256,)))
    out += [("rpn.head.conv.weight", (256, 256, 3, 3)), ("rpn.head.conv.bias", (256,)),
            ("rpn.head.cls_logits.weight", (3, 256, 1, 1)), ("rpn.head.cls_logits.bias", (3,)),
            ("rpn.head.bbox_pred.weight", (12, 256, 1, 1)), ("rpn.head.bbox_pred.bias", (12,)),
            ("roi_heads.box_head.fc6.weight", (1024, 256 * 7 * 7)), ("roi_heads.box_head.fc6.bias", (1024,)),
            ("roi_heads.box_head.fc7.weight", (1024, 1024)), ("roi_heads.box_head.fc7.bias", (1024,)),
            ("roi_heads.box_predictor.cls_score.weight", (num_classes, 1024)), ("roi_heads.box_predictor.cls_score.bias", (num_classes,)),
            ("roi_heads.box_predictor.bbox_pred.weight", (num_classes * 4, 1024)), ("roi_heads.box_predictor.bbox_pred.bias", (num_classes * 4,))]
    return out


def _model(**kw):
    from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
    torch.manual_seed(0)
    return fasterrcnn_resnet50_fpn(pretrained=False, pretrained_backbone=False, **kw)


def test_state_dict_is_torchvisions_layout_key_for_key():
    want = torchvision_fasterrcnn_resnet50_fpn_layout(91)
    assert len(want) == 295
    sd = _model(num_classes=91).state_dict()
    assert list(sd.keys()) == [k for k, _ in want]
    for k, shape in want:
        assert tuple(sd[k].shape) == shape, k
        assert sd[k].dtype == torch.float32, k
    assert [tuple(v.shape) for k, v in _model(num_classes=2).state_dict().items() if "box_predictor" in k] == [(2, 1024), (2,), (8, 1024), (8,)]


def test_reference_style_checkpoint_loads_strictly_and_is_used():
    g = torch.Generator().manual_seed(5)
    ck = {k: (torch.rand(s, generator=g) + 0.5 if k.endswith("running_var") else torch.randn(s, generator=g) * 0.05)
          for k, s in torchvision_fasterrcnn_resnet50_fpn_layout(91)}
    with_tracked = dict(ck)
    for k in list(ck):
        if k.endswith("running_var"):           # what a torch.nn.BatchNorm2d checkpoint carries next to it
            with_tracked[k.replace("running_var", "num_batches_tracked")] = torch.tensor(7)
    for state in (ck, with_tracked):
        m = _model(num_classes=91)
        res = m.load_state_dict(dict(state), strict=True)      # evaluate.py:49 / train.py:230-236: model.load_state_dict(checkpoint['model'])
        assert not res.missing_keys and not res.unexpected_keys
        for k, v in m.state_dict().items():
            assert torch.equal(v, ck[k]), k
    # the RPN head computes with the loaded tensors (CPU: the plain module graph)
    m.eval()
    f = torch.randn(1, 256, 9, 11, generator=g)
    with torch.no_grad():
        logits, deltas = m.rpn.head([f])
        t = torch.relu(torch.nn.functional.conv2d(f, ck["rpn.head.conv.weight"], ck["rpn.head.conv.bias"], padding=1))
        assert torch.allclose(logits[0], torch.nn.functional.conv2d(t, ck["rpn.head.cls_logits.weight"], ck["rpn.head.cls_logits.bias"]), rtol=1e-4, atol=1e-5)
        assert torch.allclose(deltas[0], torch.nn.functional.conv2d(t, ck["rpn.head.bbox_pred.weight"], ck["rpn.head.bbox_pred.bias"]), rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_concatenated_rpn_head_uses_the_loaded_predictors_on_the_gpu():
    """On the GPU the two predictors run as one convolution / GEMM over the concatenated, zero-padded weight (rpn.py): after
    load_state_dict -- in place, as train.py / evaluate.py do it -- training and inference both see the new tensors."""
    g = torch.Generator().manual_seed(6)
    m = _model(num_classes=91).cuda()
    ck = {k: (torch.rand(s, generator=g) + 0.5 if k.endswith("running_var") else torch.randn(s, generator=g) * 0.05)
          for k, s in torchvision_fasterrcnn_resnet50_fpn_layout(91)}
    f = torch.randn(2, 256, 25, 42, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        m.rpn.head([f])                                         # a first call with the initial weights
    m.load_state_dict(ck, strict=True)
    F = torch.nn.functional
    t = torch.relu(F.conv2d(f, ck["rpn.head.conv.weight"].cuda(), ck["rpn.head.conv.bias"].cuda(), padding=1))
    want_l = F.conv2d(t, ck["rpn.head.cls_logits.weight"].cuda(), ck["rpn.head.cls_logits.bias"].cuda())
    want_d = F.conv2d(t, ck["rpn.head.bbox_pred.weight"].cuda(), ck["rpn.head.bbox_pred.bias"].cuda())
    for grad in (False, True):
        with torch.set_grad_enabled(grad):
            logits, deltas = m.rpn.head([f])
        assert logits[0].shape == want_l.shape and deltas[0].shape == want_d.shape
        assert torch.allclose(logits[0], want_l, atol=2e-5 * float(want_l.abs().max()) + 1e-6)
        assert torch.allclose(deltas[0], want_d, atol=2e-5 * float(want_d.abs().max()) + 1e-6)
