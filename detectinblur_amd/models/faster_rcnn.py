"""`FasterRCNN` / `fasterrcnn_resnet50_fpn` with the reference's constructor surface and defaults
(reference models/faster_rcnn.py:144-243, 301-373).  No weights are downloaded here (no network):
`pretrained` / `pretrained_backbone` raise; random initialisation trains every backbone layer, as in
the reference when neither is set (:361-363).
"""
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


def fasterrcnn_resnet50_fpn(pretrained=False, progress=True, num_classes=91, pretrained_backbone=True,
                            trainable_backbone_layers=3, channels_last=True, **kwargs):
    assert 0 <= trainable_backbone_layers <= 5
    if pretrained or pretrained_backbone:
        raise RuntimeError("pretrained weights cannot be downloaded here (no network); build with pretrained=False, "
                           "pretrained_backbone=False and load a state_dict (torchvision key layout is kept)")
    trainable_backbone_layers = 5      # nothing is frozen without pretrained weights (reference :361-363)
    backbone = resnet_fpn_backbone("resnet50", False, trainable_layers=trainable_backbone_layers)
    model = FasterRCNN(backbone, num_classes, **kwargs)
    if channels_last:
        # MI355X: MIOpen's fp32 implicit-GEMM convolutions are NHWC kernels (planar tensors pay a
        # transpose around each), and the NHWC RoIAlign issues one atomic per 64 contiguous channels.
        # Values, state_dict keys and shapes are unchanged; only the strides differ.
        model = model.to(memory_format=torch.channels_last)
        model.transform.channels_last = True
    return model
