"""`FasterRCNN` / `fasterrcnn_resnet50_fpn` with the reference's constructor surface and defaults
(reference models/faster_rcnn.py:144-243, 301-373).  No weights are downloaded here (no network):
`pretrained` / `pretrained_backbone` load the files the reference would have downloaded when they already
sit in a local cache (`find_pretrained`), and raise otherwise; random initialisation trains every backbone
layer, as in the reference when neither is set (:361-363).
"""
import os

import torch
from .backbone import resnet_fpn_backbone
from .detector_ops import MultiScaleRoIAlign
from .generalized_rcnn import GeneralizedRCNN
from .net_transforms import GeneralizedRCNNTransform
from .roi_heads import FastRCNNPredictor, RoIHeads, TwoMLPHead
from .rpn import AnchorGenerator, RegionProposalNetwork, RPNHead

__all__ = ["FasterRCNN", "fasterrcnn_resnet50_fpn", "TwoMLPHead", "FastRCNNPredictor"]


class FasterRCNN(GeneralizedRCNN):
    def __init__(self, backbone, num_classes=None,
                 min_size=800, max_size=1333, image_mean=None, image_std=None,
                 rpn_anchor_generator=None, rpn_head=None,
                 rpn_pre_nms_top_n_train=2000, rpn_pre_nms_top_n_test=1000,
                 rpn_post_nms_top_n_train=2000, rpn_post_nms_top_n_test=1000,
                 rpn_nms_thresh=0.7, rpn_fg_iou_thresh=0.7, rpn_bg_iou_thresh=0.3,
                 rpn_batch_size_per_image=256, rpn_positive_fraction=0.5,
                 box_roi_pool=None, box_head=None, box_predictor=None,
                 box_score_thresh=0.05, box_nms_thresh=0.5, box_detections_per_img=100,
                 box_fg_iou_thresh=0.5, box_bg_iou_thresh=0.5,
                 box_batch_size_per_image=512, box_positive_fraction=0.25,
                 bbox_reg_weights=None, warp_internally=False):
        if not hasattr(backbone, "out_channels"):
            raise ValueError("backbone should contain an attribute out_channels specifying the number of output channels "
                             "(assumed to be the same for all the levels)")
        if num_classes is not None and box_predictor is not None:
            raise ValueError("num_classes should be None when box_predictor is specified")
        if num_classes is None and box_predictor is None:
            raise ValueError("num_classes should not be None when box_predictor is not specified")
        out_channels = backbone.out_channels
        if rpn_anchor_generator is None:
            rpn_anchor_generator = AnchorGenerator(((32,), (64,), (128,), (256,), (512,)), ((0.5, 1.0, 2.0),) * 5)
        if rpn_head is None:
            rpn_head = RPNHead(out_channels, rpn_anchor_generator.num_anchors_per_location()[0])
        rpn = RegionProposalNetwork(rpn_anchor_generator, rpn_head, rpn_fg_iou_thresh, rpn_bg_iou_thresh,
                                    rpn_batch_size_per_image, rpn_positive_fraction,
                                    dict(training=rpn_pre_nms_top_n_train, testing=rpn_pre_nms_top_n_test),
                                    dict(training=rpn_post_nms_top_n_train, testing=rpn_post_nms_top_n_test), rpn_nms_thresh)
        if box_roi_pool is None:
            box_roi_pool = MultiScaleRoIAlign(featmap_names=["0", "1", "2", "3"], output_size=7, sampling_ratio=2)
        if box_head is None:
            box_head = TwoMLPHead(out_channels * box_roi_pool.output_size[0] ** 2, 1024)
        if box_predictor is None:
            box_predictor = FastRCNNPredictor(1024, num_classes)
        roi_heads = RoIHeads(box_roi_pool, box_head, box_predictor, box_fg_iou_thresh, box_bg_iou_thresh,
                             box_batch_size_per_image, box_positive_fraction, bbox_reg_weights,
                             box_score_thresh, box_nms_thresh, box_detections_per_img)
        transform = GeneralizedRCNNTransform(min_size, max_size, image_mean or [0.485, 0.456, 0.406],
                                             image_std or [0.229, 0.224, 0.225])
        super().__init__(backbone, rpn, roi_heads, transform, warp_internally)


# what the reference downloads (models/faster_rcnn.py:295-298; torchvision.models.resnet.model_urls['resnet50'])
PRETRAINED_FILES = {"fasterrcnn_resnet50_fpn_coco": ("fasterrcnn_resnet50_fpn_coco-258fb6c6.pth",),
                    "resnet50": ("resnet50-19c8e357.pth", "resnet50-0676ba61.pth")}


def find_pretrained(kind):
    """Path of a locally cached checkpoint of `kind`, or None.  Searched: $DIB_WEIGHTS_DIR,
    $TORCH_HOME/hub/checkpoints, ~/.cache/torch/hub/checkpoints, ./weights."""
    dirs = [os.environ.get("DIB_WEIGHTS_DIR"),
            os.path.join(os.environ.get("TORCH_HOME", os.path.join(os.path.expanduser("~"), ".cache", "torch")), "hub", "checkpoints"),
            "weights"]
    for d in dirs:
        for name in PRETRAINED_FILES[kind]:
            if d and os.path.isfile(os.path.join(d, name)):
                return os.path.join(d, name)
    return None


def fasterrcnn_resnet50_fpn(pretrained=False, progress=True, num_classes=91, pretrained_backbone=True,
                            trainable_backbone_layers=3, channels_last=True, **kwargs):
    """reference models/faster_rcnn.py:301-373.  `pretrained_backbone="auto"`: ImageNet trunk if a cached
    file exists, random initialisation (with a note) otherwise -- what `train.py` asks for, since the
    reference's default (True) means "download"."""
    assert 0 <= trainable_backbone_layers <= 5
    full = find_pretrained("fasterrcnn_resnet50_fpn_coco") if pretrained else None
    if pretrained and full is None:
        raise RuntimeError("pretrained=True: %s is not in $DIB_WEIGHTS_DIR / the torch hub cache / ./weights and "
                           "cannot be downloaded here (no network)" % PRETRAINED_FILES["fasterrcnn_resnet50_fpn_coco"][0])
    trunk = None
    if pretrained:
        pretrained_backbone = False      # no need for the trunk file when the full model is loaded (reference :364-366)
    elif pretrained_backbone:
        trunk = find_pretrained("resnet50")
        if trunk is None:
            if pretrained_backbone != "auto":
                raise RuntimeError("pretrained_backbone=True: none of %s is cached locally and there is no network; pass "
                                   "pretrained_backbone=False (random init)" % (PRETRAINED_FILES["resnet50"],))
            print("No cached ImageNet ResNet-50 found: the trunk starts from random weights, every layer trainable.")
            pretrained_backbone = False
    if not (pretrained or pretrained_backbone):
        trainable_backbone_layers = 5      # nothing is frozen without pretrained weights (reference :361-363)
    backbone = resnet_fpn_backbone("resnet50", False, trainable_layers=trainable_backbone_layers)
    if trunk is not None:
        sd = torch.load(trunk, map_location="cpu", weights_only=True)
        # strict=False only to tolerate what is known to differ (the classifier head is dropped above; frozen batch-norms
        # carry no num_batches_tracked): anything else missing or unexpected means the file has another key layout and
        # would leave the trunk at random weights with frozen layers
        bad = backbone.body.load_state_dict({k: v for k, v in sd.items() if not k.startswith("fc.")}, strict=False)
        missing = [k for k in bad.missing_keys if not k.endswith("num_batches_tracked")]
        unexpected = [k for k in bad.unexpected_keys if not k.endswith("num_batches_tracked")]
        if missing or unexpected:
            raise RuntimeError("%s does not hold a torchvision ResNet-50 state dict: missing %s, unexpected %s"
                               % (trunk, missing[:5], unexpected[:5]))
    model = FasterRCNN(backbone, num_classes, **kwargs)
    if full is not None:
        model.load_state_dict(torch.load(full, map_location="cpu", weights_only=True))
    if channels_last:
        # MI355X: MIOpen's fp32 implicit-GEMM convolutions are NHWC kernels (planar tensors pay a
        # transpose around each), and the NHWC RoIAlign issues one atomic per 64 contiguous channels.
        # Values, state_dict keys and shapes are unchanged; only the strides differ.
        model = model.to(memory_format=torch.channels_last)
        model.transform.channels_last = True
    return model
