"""Import harness for the *actual* reference (read-only at /root/reference).

TEST INFRASTRUCTURE ONLY.  This module exists so that `oracle/gen_goldens.py`
and the `-m "not gpu"` tests that run inside the build container can execute the
reference's own Python for the hot path and pin the oracle restatement
(`oracle/dib_oracle.py`) against it.  `/root/reference` does not exist on the
GPU box, so nothing under `-m gpu`, `smoke()` or `bench.py` imports this file.

The reference's modules import packages that are not installed in this image
(cv2, torchvision, pycocotools, tensorboard).  None of them is *used* by the
functions on the hot path except `cv2.normalize` inside
the CPU FFT blur, for which a functional numpy shim is installed; everything
else is a MagicMock.  `np.math` (removed in numpy 2) is aliased to `math` because
`motion_blur/generate_PSF.py:59` calls `np.math.floor`.  The reference's files
are never copied or modified.
"""
import importlib
import math
import os
import sys
import types
from unittest import mock

import numpy as np

REFERENCE_ROOT = os.environ.get("DIB_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "models", "blur_functions.py"))


def _cv2_shim():
    cv2 = types.ModuleType("cv2")
    cv2.NORM_MINMAX = 32
    cv2.CV_32F = 5
    cv2.INTER_LANCZOS4 = 4
    cv2.COLOR_RGB2BGR = 4

    def normalize(src, dst=None, alpha=0, beta=1, norm_type=32, dtype=5):
        # cv2.normalize(NORM_MINMAX): global min/max over all channels,
        # dst = (src - min) * (beta - alpha) / (max - min) + alpha, as float32.
        s = np.asarray(src, dtype=np.float64)
        lo, hi = float(s.min()), float(s.max())
        scale = (beta - alpha) / (hi - lo) if hi > lo else 0.0
        out = ((s - lo) * scale + alpha).astype(np.float32)
        if dst is not None and isinstance(dst, np.ndarray) and dst.dtype == np.float32 \
                and dst.shape == out.shape:
            dst[...] = out
        return out

    cv2.normalize = normalize
    return cv2


_loaded = {}


def load():
    """Returns a namespace with the reference modules for the hot path."""
    if _loaded:
        return _loaded["ns"]
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    if not hasattr(np, "math"):
        np.math = math  # generate_PSF.py:59
    stubs = ["torchvision", "torchvision.transforms", "torchvision.transforms.functional",
             "torchvision.ops", "torchvision.ops.misc", "torchvision.models",
             "torchvision.models.detection", "torchvision.models.detection.mask_rcnn",
             "torchvision.models.utils", "torchvision.models.detection.rpn",
             "torchvision.models.detection.roi_heads", "torchvision.models.detection.image_list",
             "torchvision.models.detection.anchor_utils", "torchvision.models.detection.backbone_utils",
             "pycocotools", "pycocotools.mask", "pycocotools.coco", "pycocotools.cocoeval",
             "torch.utils.tensorboard", "skimage", "skimage.io"]
    for name in stubs:
        if name not in sys.modules:
            sys.modules[name] = mock.MagicMock(name=name)
    sys.modules["cv2"] = _cv2_shim()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # The reference's top-level module names (utils, transforms, models, engine)
    # are generic; import them under their own names but keep handles here.
    ns = types.SimpleNamespace()
    ns.generate_trajectory = importlib.import_module("motion_blur.generate_trajectory")
    ns.generate_PSF = importlib.import_module("motion_blur.generate_PSF")
    ns.blur_image = importlib.import_module("motion_blur.blur_image")
    ns.utils = importlib.import_module("utils")
    ns.transforms = importlib.import_module("transforms")
    ns.blur_functions = importlib.import_module("models.blur_functions")
    _loaded["ns"] = ns
    return ns


def load_net_transforms():
    """The reference's models/net_transforms.py (GeneralizedRCNNTransform, resize_boxes).  Its arithmetic is plain
    torch; torchvision only contributes `_is_tracing()` (False outside ONNX export) and the two-field `ImageList`
    container, both supplied here."""
    load()
    import torchvision                                  # the MagicMock installed by load()
    torchvision._is_tracing = lambda: False

    class ImageList(object):
        def __init__(self, tensors, image_sizes):
            self.tensors, self.image_sizes = tensors, image_sizes

    sys.modules["torchvision.models.detection.image_list"].ImageList = ImageList
    return importlib.import_module("models.net_transforms")


_EXTRA_STUBS = ["torchvision.models.detection.transform", "torchvision.models.detection.faster_rcnn"]


def load_detector():
    """The reference's models/faster_rcnn.py and models/generalized_rcnn.py (rows A13 / A14).  What these files
    define themselves -- constructor defaults, the argument lists handed to the torchvision classes, the control
    flow of `GeneralizedRCNN.forward` -- runs as is; the torchvision classes they import are MagicMocks that
    oracle/gen_detector_pins.py replaces by recording fakes (oracle/pin_inputs.py)."""
    load()
    for name in _EXTRA_STUBS:
        if name not in sys.modules:
            sys.modules[name] = mock.MagicMock(name=name)
    return importlib.import_module("models.faster_rcnn"), importlib.import_module("models.generalized_rcnn")


def load_engine():
    """The reference's engine.py (`train_one_epoch`, `evaluate`, the three ensemble routers; rows A15 / A17).
    `coco_eval` / `coco_utils` (pycocotools, `torch._six`) are stubbed: the generator installs a recording
    evaluator in their place.  engine.py:277 calls `torch.cuda.synchronize()` unconditionally; on this GPU-less
    container the caller patches it to a no-op for the duration of the call."""
    load()
    for name in _EXTRA_STUBS + ["coco_eval", "coco_utils"]:
        if name not in sys.modules:
            sys.modules[name] = mock.MagicMock(name=name)
    return importlib.import_module("engine")


def load_estimator_engine():
    """The reference's engine_blur_estimator.py (`train_one_epoch`, `evaluate`, its private `manual_blur` with the optional
    800-pixel round trip, the label functions; row f4).  Same stubs as `load_engine`; the caller patches
    `torch.cuda.synchronize` (engine_blur_estimator.py:257, :261, :347) to a no-op on this GPU-less container and selects the
    `.to(device)` branches with `distributed_mode=True`."""
    load_net_transforms()
    for name in _EXTRA_STUBS + ["coco_eval", "coco_utils"]:
        if name not in sys.modules:
            sys.modules[name] = mock.MagicMock(name=name)
    return importlib.import_module("engine_blur_estimator")
