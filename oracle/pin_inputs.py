"""Recording fakes, toy models and seeded batches for the detector-row pins.

TEST INFRASTRUCTURE ONLY -- shared by `oracle/gen_detector_pins.py`, which drives the REAL reference files
(`models/faster_rcnn.py`, `models/generalized_rcnn.py`, `engine.py`) with these objects in this container and
stores what they did under tests/golden/, and by `tests/test_detector_pins.py`, which drives this repo's classes
with the very same objects and compares.  Nothing here restates reference logic: the fakes only RECORD how they
are called, the toy models are arbitrary small networks that make every input of a step (blurred pixels,
expanded boxes, normalisation rows, theta / lambda scalars) visible in a loss or a detection.
"""
import collections

import numpy as np
import torch
from torch import nn

import golden_inputs as GI


# ---- turning call arguments into JSON ---------------------------------------------------------------------------

def describe(x):
    if isinstance(x, Recorder):
        return "<%s>" % x._name
    if isinstance(x, torch.Tensor):
        return {"shape": list(x.shape), "dtype": str(x.dtype).replace("torch.", ""),
                "sum": round(float(x.detach().double().sum()), 5)}
    if isinstance(x, np.ndarray):
        return {"shape": list(x.shape), "dtype": str(x.dtype), "sum": round(float(x.astype(np.float64).sum()), 5)}
    if isinstance(x, nn.Module):
        return {"module": type(x).__name__, "params": {k: list(v.shape) for k, v in x.state_dict().items()}}
    if hasattr(x, "tensors") and hasattr(x, "image_sizes"):
        return {"tensors": describe(x.tensors), "image_sizes": [list(s) for s in x.image_sizes]}
    if isinstance(x, dict):
        return collections.OrderedDict((str(k), describe(v)) for k, v in x.items())
    if isinstance(x, (list, tuple, torch.Size)):
        return [describe(v) for v in x]
    if isinstance(x, (bool, int, float, str, type(None))):
        return x
    if isinstance(x, (np.integer, np.floating)):
        return x.item()
    return "<%s>" % type(x).__name__


# ---- A14: constructor recorders ---------------------------------------------------------------------------------

class Recorder(object):
    """Base of the fake torchvision classes: the constructor call is all that is kept."""
    _log = None
    _name = "Recorder"

    def __init__(self, *args, **kwargs):
        self._log.append([self._name, describe(args), describe(kwargs)])
        self._init(*args, **kwargs)

    def _init(self, *args, **kwargs):
        pass


def detector_fakes(log):
    """{name: fake} for every name models/faster_rcnn.py takes from torchvision (reference :7-16) plus its own
    input transform (:241).  The fakes answer the three questions the constructor asks of them."""
    def cls(name, **body):
        return type(name, (Recorder,), dict(_log=log, _name=name, **body))

    def roi_init(self, featmap_names=None, output_size=None, sampling_ratio=None):
        self.output_size = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)

    class Body(object):
        def load_state_dict(self, sd, strict=True):
            log.append(["body.load_state_dict", sorted(sd.keys()), {"strict": strict}])
            return collections.namedtuple("Keys", "missing_keys unexpected_keys")([], [])

    class Backbone(Recorder):            # described as "<backbone>" when handed on
        out_channels = 256
        _name = "backbone"

        def __init__(self):
            self.body = Body()

    def resnet_fpn_backbone(*args, **kwargs):
        log.append(["resnet_fpn_backbone", describe(args), describe(kwargs)])
        return Backbone()

    return {"AnchorGenerator": cls("AnchorGenerator", num_anchors_per_location=lambda self: [3, 3, 3, 3, 3]),
            "RPNHead": cls("RPNHead"),
            "RegionProposalNetwork": cls("RegionProposalNetwork"),
            "MultiScaleRoIAlign": cls("MultiScaleRoIAlign", _init=roi_init),
            "RoIHeads": cls("RoIHeads"),
            "GeneralizedRCNNTransform": cls("GeneralizedRCNNTransform"),
            "resnet_fpn_backbone": resnet_fpn_backbone}


def detector_ctor_cases():
    """kwargs of `fasterrcnn_resnet50_fpn(...)` to record (reference models/faster_rcnn.py:301-373, 144-243)."""
    return collections.OrderedDict([
        ("random_init_default", dict(num_classes=91, pretrained=False, pretrained_backbone=False)),
        ("random_init_trainable0", dict(num_classes=91, pretrained=False, pretrained_backbone=False, trainable_backbone_layers=0)),
        ("imagenet_trunk_trainable3", dict(num_classes=91, pretrained=False, pretrained_backbone=True, trainable_backbone_layers=3)),
        ("imagenet_trunk_trainable1", dict(num_classes=91, pretrained=False, pretrained_backbone=True, trainable_backbone_layers=1)),
        ("overrides", dict(num_classes=5, pretrained=False, pretrained_backbone=False, min_size=512, max_size=640,
                           image_mean=[0.1, 0.2, 0.3], image_std=[0.4, 0.5, 0.6],
                           rpn_pre_nms_top_n_train=11, rpn_pre_nms_top_n_test=12, rpn_post_nms_top_n_train=13,
                           rpn_post_nms_top_n_test=14, rpn_nms_thresh=0.6, rpn_fg_iou_thresh=0.65, rpn_bg_iou_thresh=0.25,
                           rpn_batch_size_per_image=64, rpn_positive_fraction=0.4, box_score_thresh=0.2, box_nms_thresh=0.45,
                           box_detections_per_img=7, box_fg_iou_thresh=0.55, box_bg_iou_thresh=0.45,
                           box_batch_size_per_image=128, box_positive_fraction=0.3, bbox_reg_weights=(1.0, 2.0, 3.0, 4.0),
                           warp_internally=True)),
    ])


# ---- A13: recording sub-modules for GeneralizedRCNN.forward --------------------------------------------------------

class FakeImageList(object):
    def __init__(self, tensors, image_sizes):
        self.tensors, self.image_sizes = tensors, image_sizes


class ForwardFakes(object):
    """transform / backbone / rpn / roi_heads / warper that log every call (name + described arguments) and return
    small deterministic values, so that the data flow through `GeneralizedRCNN.forward` is visible in the log."""

    def __init__(self, backbone_returns_tensor=False):
        self.log = []
        self.backbone_returns_tensor = backbone_returns_tensor
        fk = self

        class Transform(object):
            def __call__(self, images, targets=None, newMeans=None, newSTDs=None):
                fk.log.append(["transform", describe(images), describe(targets), describe(newMeans), describe(newSTDs)])
                out_t = None
                if targets is not None:
                    out_t = [dict(t, boxes=t["boxes"] * 2.0) for t in targets]
                return FakeImageList(torch.stack([im * 0.5 for im in images]), [tuple(im.shape[-2:]) for im in images]), out_t

            def postprocess(self, detections, image_sizes, original_sizes):
                fk.log.append(["postprocess", describe(detections), describe(image_sizes), describe(original_sizes)])
                return [dict(d, post=torch.tensor([float(k)])) for k, d in enumerate(detections)]

        def backbone(x):
            fk.log.append(["backbone", describe(x)])
            if fk.backbone_returns_tensor:
                return x * 2.0
            return collections.OrderedDict([("0", x * 2.0), ("1", x[..., ::2, ::2] * 3.0)])

        def rpn(images, features, targets=None):
            fk.log.append(["rpn", describe(images), describe(features), describe(targets)])
            n = images.tensors.shape[0]
            return ([torch.full((2, 4), float(k)) for k in range(n)],
                    collections.OrderedDict([("loss_objectness", torch.tensor(0.25)), ("loss_rpn_box_reg", torch.tensor(0.5))]))

        def roi_heads(features, proposals, image_sizes, targets=None):
            fk.log.append(["roi_heads", describe(features), describe(proposals), describe(image_sizes), describe(targets)])
            dets = [{"boxes": p + 1.0, "labels": torch.ones(2, dtype=torch.int64), "scores": torch.full((2,), 0.5)} for p in proposals]
            return dets, collections.OrderedDict([("loss_classifier", torch.tensor(1.0)), ("loss_box_reg", torch.tensor(2.0))])

        def warper(x, thetas, lambda1s, lambda2s):
            fk.log.append(["warper", describe(x), describe(thetas), describe(lambda1s), describe(lambda2s)])
            return x + 1.0

        self.transform, self.backbone, self.rpn, self.roi_heads, self.warper = Transform(), backbone, rpn, roi_heads, warper


def forward_cases():
    """name -> dict(training, warp, kill, tensor_backbone, targets: None | 'ok' | 'shape' | 'type' | 'degenerate')."""
    c = collections.OrderedDict()
    c["train_ok"] = dict(training=True, targets="ok")
    c["eval_ok"] = dict(training=False, targets=None)
    c["eval_with_targets"] = dict(training=False, targets="ok")
    c["train_no_targets"] = dict(training=True, targets=None)
    c["train_bad_shape"] = dict(training=True, targets="shape")
    c["train_bad_type"] = dict(training=True, targets="type")
    c["train_degenerate"] = dict(training=True, targets="degenerate")
    c["eval_degenerate"] = dict(training=False, targets="degenerate")
    c["warp_train"] = dict(training=True, targets="ok", warp=True)
    c["warp_eval_kill"] = dict(training=False, targets=None, warp=True, kill=True)
    c["warp_eval"] = dict(training=False, targets=None, warp=True)
    c["tensor_backbone_eval"] = dict(training=False, targets=None, tensor_backbone=True)
    return c


def forward_inputs(case):
    rs = np.random.RandomState(99)
    images = [torch.from_numpy(rs.random_sample((3, 8, 10)).astype(np.float32)) for _ in range(2)]
    kind = case.get("targets")
    targets = None
    if kind is not None:
        boxes = [torch.tensor([[1.0, 1.0, 4.0, 5.0], [2.0, 0.0, 7.0, 3.0]]), torch.tensor([[0.0, 2.0, 5.0, 6.0]])]
        if kind == "shape":
            boxes[1] = torch.zeros(3, 5)
        elif kind == "type":
            boxes[0] = [[1.0, 1.0, 4.0, 5.0]]
        elif kind == "degenerate":
            boxes[1] = torch.tensor([[0.0, 2.0, 5.0, 6.0], [3.0, 4.0, 3.0, 9.0]])
        targets = [{"boxes": b, "labels": torch.ones(len(b), dtype=torch.int64)} for b in boxes]
    kw = dict(thetas=torch.tensor([0.1, 0.2]), lambda1s=torch.tensor([0.5, 0.8]), lambda2s=torch.tensor([1.0, 0.25]),
              newMeans=np.array([[0.4, 0.5, 0.6], [0.1, 0.2, 0.3]]), newSTDs=np.array([[0.2, 0.2, 0.2], [0.3, 0.3, 0.3]]))
    if case.get("kill"):
        kw["killWarp"] = True
    return images, targets, kw


def run_forward_case(model_cls, case):
    """Builds `model_cls(backbone, rpn, roi_heads, transform, warp_internally)` on fresh fakes, calls forward, and
    returns {"log": [...], "result": described | None, "error": [type, text] | None}."""
    fk = ForwardFakes(backbone_returns_tensor=case.get("tensor_backbone", False))
    model = model_cls(fk.backbone, fk.rpn, fk.roi_heads, fk.transform, bool(case.get("warp")))
    if case.get("warp"):
        model._modules.pop("warper", None)      # the class built its own Warper; the recorder takes its place
        model.__dict__["warper"] = fk.warper
    model.train(case["training"])
    images, targets, kw = forward_inputs(case)
    out = {"log": fk.log, "result": None, "error": None}
    try:
        res = model(images, targets, **kw)
        out["result"] = describe(res)
    except Exception as e:          # noqa: BLE001 -- the exception IS the datum
        out["error"] = [type(e).__name__, str(e)]
    return out


# ---- A17: router grids ----------------------------------------------------------------------------------------------

def router_dicts():
    d = [{"blurring": False, "param_index": None, "fraction_index": None},
         {"blurring": True, "param_index": None, "fraction_index": None}]
    for p in (-1, 0, 1, 2, 3):
        for f in (-1, 0, 3):
            d.append({"blurring": True, "param_index": p, "fraction_index": f})
    d.append({"blurring": False, "param_index": 1, "fraction_index": 2})
    return d


def router_oracle_batches():
    """Every batch of 1 and of 2 dicts, plus the batches of 3 that start with two fall-through dicts."""
    d = router_dicts()
    out = [[a] for a in d] + [[a, b] for a in d for b in d]
    fall = [x for x in d if x["blurring"] and x["param_index"] in (-1, 3) and x["fraction_index"] != -1]
    out += [[a, b, c] for a in fall[:2] for b in fall[:2] for c in d]
    return out


def router_estimations():
    """(logits, n_models) pairs: one-hot rows for every class of the 16-way and 4-way estimators, one class
    beyond either, near-ties, and a batch of two rows (argmax then runs over the flattened tensor)."""
    out = []
    for n in (4, 5, 16, 17):
        for k in range(n):
            v = torch.zeros(1, n)
            v[0, k] = 1.0
            out.append(v)
    out.append(torch.tensor([[0.2, 0.9, 0.9, 0.1]]))
    out.append(torch.tensor([[0.0, 0.0, 0.0, 0.0]]))
    out.append(torch.tensor([[0.1, 0.2, 0.3, 0.4], [0.9, 0.0, 0.0, 0.0]]))
    out.append(torch.tensor([[0.1] * 16, [0.0] * 5 + [1.0] + [0.0] * 10]))
    return out


# ---- A15 / A17: toy detector, toy estimator, seeded batches -------------------------------------------------------------

def _seeded(module, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in module.parameters():
            p.copy_((torch.rand(p.shape, generator=g) - 0.5) * 0.6)
    return module


class ToyDetector(nn.Module):
    """Stands where `fasterrcnn_resnet50_fpn` stands in a step: same call signature (reference
    models/generalized_rcnn.py:78), four losses when training, `{boxes, labels, scores}` per image otherwise.
    Every argument the engines compute reaches an output: pixels (after blur) and the normalisation rows through the
    convolution, target boxes (after expansion) through the losses, theta / lambda through a loss and the scores."""

    def __init__(self, seed=0):
        super().__init__()
        self.conv = nn.Conv2d(3, 4, 3, padding=1)
        self.fc = nn.Linear(4, 4)
        _seeded(self, 1000 + seed)
        self.calls = []            # one record per forward: kwargs seen + the LR of `lr_probe` at that moment
        self.lr_probe = None

    tick = [0]                     # global call counter: the order in which an ensemble's members were called

    @staticmethod
    def _scal(v, device):
        if v is None:
            return None
        if isinstance(v, (list, tuple)):
            v = torch.stack([torch.as_tensor(e).reshape(-1)[0] for e in v])
        return v.to(device).float().reshape(-1)

    def forward(self, images, targets=None, thetas=None, lambda1s=None, lambda2s=None, killWarp=False, newMeans=None, newSTDs=None):
        dev = images[0].device
        feats = []
        for i, img in enumerate(images):
            m = torch.as_tensor(np.asarray(newMeans[i]), dtype=torch.float32, device=dev)[:, None, None]
            s = torch.as_tensor(np.asarray(newSTDs[i]), dtype=torch.float32, device=dev)[:, None, None]
            feats.append(self.conv(((img - m) / s)[None]).mean(dim=(2, 3))[0])
        F = torch.stack(feats)
        out = self.fc(F)
        th, l1, l2 = self._scal(thetas, dev), self._scal(lambda1s, dev), self._scal(lambda2s, dev)
        ToyDetector.tick[0] += 1
        self.calls.append({"order": ToyDetector.tick[0], "training": self.training, "killWarp": bool(killWarp), "n": len(images),
                           "dtypes": sorted({str(i.dtype) for i in images}), "thetas": None if th is None else [round(float(v), 6) for v in th],
                           "lambda1s": None if l1 is None else [round(float(v), 6) for v in l1],
                           "lambda2s": None if l2 is None else [round(float(v), 6) for v in l2],
                           "lr": None if self.lr_probe is None else self.lr_probe.param_groups[0]["lr"]})
        extra = out.sum() * 0.0
        if th is not None:
            extra = extra + (th * l1 / l2).sum() * 1e-3
        if self.training:
            tb = torch.stack([t["boxes"].float().mean(0) for t in targets]) / 100.0
            return {"loss_classifier": ((out - tb) ** 2).mean(), "loss_box_reg": out.abs().mean() * 0.1,
                    "loss_objectness": (F ** 2).mean(), "loss_rpn_box_reg": (out[:, 0] * tb[:, 1]).mean() * 0.01 + extra}
        dets = []
        for i, img in enumerate(images):
            H, W = float(img.shape[-2]), float(img.shape[-1])
            o = out[i].abs()
            x1, y1 = o[0] * W * 0.3, o[1] * H * 0.3
            boxes = torch.stack([torch.stack([x1, y1, x1 + 4 + o[2] * W * 0.3, y1 + 4 + o[3] * H * 0.3]),
                                 torch.stack([y1 * 0.5, x1 * 0.5, y1 * 0.5 + 6, x1 * 0.5 + 9])])
            score = torch.sigmoid(out[i, :2] + extra)
            dets.append({"boxes": boxes, "labels": torch.tensor([1, 2], dtype=torch.int64, device=dev), "scores": score})
        return dets


class ToyEstimator(nn.Module):
    """Stands where the ResNet-18 blur estimator stands (reference evaluate.py:186-205): batched image -> logits."""

    def __init__(self, n_classes, seed=0):
        super().__init__()
        self.fc = nn.Linear(3, n_classes)
        _seeded(self, 2000 + seed)
        self.calls = []

    def forward(self, x):
        self.calls.append(describe(x))
        return self.fc((x[:, :, 5, 7] - x[:, :, 20, 30]) * 4.0)      # two pixels: varies from image to image


class ListLoader(list):
    """A data loader that is a list of ready batches (`len`, iteration, `.dataset`)."""
    dataset = None


class RecordingWriter(object):
    def __init__(self):
        self.scalars = []

    def add_scalar(self, tag, value, step):
        self.scalars.append([tag, float(value.detach()) if isinstance(value, torch.Tensor) else float(value), int(step)])


def _blur_dict(rs, param_i, frac_i, blurring, psf_kind="crop"):
    if not blurring:      # what BlurImage writes for an image it leaves sharp (reference transforms.py:455-461)
        return {"blurring": False, "psf": [0], "theta_rad": 0, "scale_factor_lambda1": 1, "scale_factor_lambda2": 1,
                "param_index": None, "fraction_index": None}
    psf = GI.golden_psf(GI.PARAMS[param_i], frac_i, psf_kind)
    return {"blurring": True, "psf": psf, "theta_rad": float(rs.uniform(-1.5, 1.5)),
            "scale_factor_lambda1": float(rs.uniform(0.7, 1.0)), "scale_factor_lambda2": float(rs.uniform(0.7, 1.0)),
            "param_index": param_i, "fraction_index": frac_i if frac_i < 5 else -1}


def _target(rs, H, W, n, image_id):
    x1 = rs.uniform(0, W - 12, n)
    y1 = rs.uniform(0, H - 12, n)
    b = np.stack([x1, y1, x1 + rs.uniform(4, W / 2, n), y1 + rs.uniform(4, H / 2, n)], 1)
    b[:, 2] = np.minimum(b[:, 2], W)
    b[:, 3] = np.minimum(b[:, 3], H)
    return {"boxes": torch.from_numpy(b.astype(np.float32)), "labels": torch.from_numpy(rs.randint(1, 91, n).astype(np.int64)),
            "image_id": torch.tensor([image_id])}


def train_batches(blur, poison_at=None):
    """5 batches x 2 images, ragged sizes >= 65 (reflect branch); with `blur`, image 0 of every batch and image 1 of
    the even ones carry golden PSFs of all three types and of exposures 1/18 ... 1 and 1/25.  `poison_at`: one pixel of that
    batch's first image is +inf (the toy detector's losses of that step are then not finite)."""
    rs = np.random.RandomState(4711)
    loader = ListLoader()
    for k in range(5):
        shapes = [(3, 70 + 2 * k, 90), (3, 80, 75 + k)]
        images = tuple(torch.from_numpy(rs.random_sample(s).astype(np.float32)) for s in shapes)
        if poison_at == k:
            images[0][1, 7, 9] = float("inf")
        targets = tuple(_target(rs, s[1], s[2], 3 + (k + j) % 2, 10 * k + j) for j, s in enumerate(shapes))
        if blur:
            dicts = (_blur_dict(rs, k % 3, (k * 2) % 6, True), _blur_dict(rs, (k + 1) % 3, (k + 3) % 6, k % 2 == 0))
        else:
            dicts = tuple({"blurring": False} for _ in shapes)
        loader.append((images, targets, dicts))
    return loader


class FakeCoco(object):
    def __init__(self, loader, extra_ann_for=()):
        self.imgToAnns = collections.OrderedDict()
        anns, k = [], 0
        for images, targets, _ in loader:
            for t in targets:
                iid = int(t["image_id"].item())
                self.imgToAnns[iid] = []
                boxes = t["boxes"].tolist()
                if iid in extra_ann_for:         # a ground-truth annotation the target lost (crowd / degenerate)
                    boxes = boxes + [[1.0, 2.0, 3.0, 4.0], [5.0, 6.0, 7.0, 8.0]]
                for b in boxes:
                    a = {"id": k, "image_id": iid, "bbox": [b[0], b[1], b[2] - b[0], b[3] - b[1]], "category_id": 1}
                    self.imgToAnns[iid].append(a)
                    anns.append(a)
                    k += 1
        self.dataset = {"annotations": anns}


class FakeBboxEval(object):
    stats = [0.0] * 12

    def summarize(self):
        return self.stats


class FakeCocoEvaluator(object):
    """Stands where `coco_eval.CocoEvaluator` stands inside `engine.evaluate`: keeps what it is given."""
    last = None

    def __init__(self, coco_gt, iou_types, **kw):
        self.coco_gt, self.iou_types = coco_gt, list(iou_types)
        self.coco_eval = {"bbox": FakeBboxEval()}
        self.updates, self.calls = [], []
        self.img_ids = []
        FakeCocoEvaluator.last = self

    def update(self, res):
        self.calls.append("update")
        self.updates.append({int(k): {n: v.detach().cpu().numpy().copy() for n, v in o.items()} for k, o in res.items()})
        self.img_ids.extend(int(k) for k in res)

    def synchronize_between_processes(self):
        self.calls.append("synchronize_between_processes")

    def accumulate(self):
        self.calls.append("accumulate")

    def summarize(self):
        self.calls.append("summarize")
        return FakeBboxEval.stats


def eval_batches(blur, param_cycle=(0, 1, 2, 0), frac_cycle=(1, 3, 5, 4)):
    """4 batches of one image (the reference evaluates with batch size 1, evaluate.py:335-339)."""
    rs = np.random.RandomState(815)
    loader = ListLoader()
    for k in range(4):
        s = (3, 72 + 3 * k, 96 - 2 * k)
        images = (torch.from_numpy(rs.random_sample(s).astype(np.float32)),)
        targets = (_target(rs, s[1], s[2], 2 + k % 3, 100 + k),)
        dicts = (_blur_dict(rs, param_cycle[k], frac_cycle[k], True),) if blur else ({"blurring": False},)
        loader.append((images, targets, dicts))
    loader.dataset = object()
    return loader


def eval_cases():
    """name -> kwargs of `engine.evaluate` + how to build the models (reference engine.py:220-416)."""
    c = collections.OrderedDict()
    blur = dict(blurring_images=True, gpu_blur=True, expand_target_boxes=True)
    c["single_blur"] = dict(kw=dict(blur, use_custom_image_norm=True), ensemble=0, estimator=0, blur=True)
    c["single_blur_early_stop"] = dict(kw=dict(blur, early_stop=1), ensemble=0, estimator=0, blur=True)
    c["single_vanilla"] = dict(kw=dict(vanilla_eval=True), ensemble=0, estimator=0, blur=False)
    c["ensemble_oracle"] = dict(kw=dict(blur, use_ensemble=True), ensemble=4, estimator=0, blur=True)
    c["ensemble_estimator_lehe"] = dict(kw=dict(blur, use_ensemble=True, LEHE=True), ensemble=4, estimator=4, blur=True)
    c["ensemble_estimator_16"] = dict(kw=dict(blur, use_ensemble=True, use_custom_image_norm=True), ensemble=4, estimator=16, blur=True)
    return c


def routes_of(models):
    """Index of the model that served each call, in call order."""
    return [k for _, k in sorted((c["order"], k) for k, m in enumerate(models) for c in m.calls)]


def build_eval_models(case, device="cpu"):
    model = ToyDetector(7).to(device) if not case["ensemble"] else None
    ens = [ToyDetector(20 + k).to(device) for k in range(case["ensemble"])] or None
    est = ToyEstimator(case["estimator"], 3).to(device) if case["estimator"] else None
    return model, ens, est


# ---- f4: the blur estimator's own loops (reference engine_blur_estimator.py:132-298, :301-492) -------------------------------

class ToyClassifier(nn.Module):
    """Stands where the ResNet-18 blur estimator stands in `engine_blur_estimator.train_one_epoch` / `evaluate`: the batched,
    normalised, resized image in, logits out.  Every pixel (after blur, quantisation, block artefacts, normalisation, resize and
    padding) reaches the logits through a strided convolution and a spatial mean."""

    def __init__(self, n_classes, seed=0):
        super().__init__()
        self.conv = nn.Conv2d(3, 6, 5, stride=4, padding=2)
        self.fc = nn.Linear(6, n_classes)
        _seeded(self, 3000 + seed)
        self.calls = []            # per forward: batch shape, dtype, the LR of `lr_probe`, training flag
        self.lr_probe = None

    def forward(self, x):
        self.calls.append({"shape": list(x.shape), "dtype": str(x.dtype), "training": self.training,
                           "lr": None if self.lr_probe is None else self.lr_probe.param_groups[0]["lr"]})
        return self.fc(torch.tanh(self.conv(x)).mean(dim=(2, 3)))


def est_train_cases():
    """name -> (loader kind, kwargs of engine_blur_estimator.train_one_epoch, classes)."""
    c = collections.OrderedDict()
    blur = dict(blur_train=True, gpu_blur=True)
    c["plain16"] = dict(kind="plain", kw=dict(), classes=16)
    c["blur16"] = dict(kind="blur", kw=dict(blur), classes=16)
    c["blur_lehe"] = dict(kind="blur_label", kw=dict(blur, LEHE_blur_seg=True), classes=4)
    c["blur_lehe_epoch1_crop"] = dict(kind="blur", kw=dict(blur, LEHE_blur_seg=True, epoch=1, crop_images=True), classes=4)
    c["blur_resize_quant"] = dict(kind="blur_portrait", kw=dict(blur, resize_images=True, quantize_image=True), classes=16)
    c["blur_block_early_stop"] = dict(kind="blur", kw=dict(blur, add_block=True, early_stop=2), classes=16)
    c["blur_train_without_gpu_blur"] = dict(kind="blur", kw=dict(blur_train=True, gpu_blur=False), classes=16)
    return c


def est_eval_cases():
    c = collections.OrderedDict()
    blur = dict(blurring_images=True, gpu_blur=True)
    c["plain16"] = dict(kind="plain", kw=dict(), classes=16)
    c["blur16"] = dict(kind="blur", kw=dict(blur), classes=16)
    c["blur_lehe_send_back"] = dict(kind="blur_label", kw=dict(blur, LEHE_blur_seg=True, send_back_preds_targets=True), classes=4)
    c["blur_resize_quant"] = dict(kind="blur_portrait", kw=dict(blur, resize_images=True, quantize_image=True), classes=16)
    c["blur_block_early_stop"] = dict(kind="blur", kw=dict(blur, add_block=True, early_stop=1, LEHE_blur_seg=True), classes=4)
    return c


def est_batches(kind, train):
    """Training: 5 batches x 2 images; evaluation: 5 batches x 1 image (reference train_blur_estimator.py:206).  Sizes >= 65 (the
    reflect branch); `blur_portrait` makes image 0 higher than wide (the reference's resize round trip transposes those and
    crops with the ORIGINAL height and width, engine_blur_estimator.py:33-44, :64-72); `blur_label` gives one dict per batch an
    explicit `blur_est_label`."""
    rs = np.random.RandomState(9021 if train else 9022)
    loader = ListLoader()
    for k in range(5):
        if kind == "blur_portrait":
            shapes = [(3, 96 + 2 * k, 70), (3, 68, 88 + k)]
        else:
            shapes = [(3, 70 + 2 * k, 90), (3, 80, 75 + k)]
        if not train:
            shapes = shapes[k % 2:k % 2 + 1]
        images = tuple(torch.from_numpy(rs.random_sample(s).astype(np.float32)) for s in shapes)
        targets = tuple(_target(rs, s[1], s[2], 2 + (k + j) % 2, 10 * k + j) for j, s in enumerate(shapes))
        if kind == "plain":
            dicts = tuple({"blurring": False, "psf": [0], "param_index": None, "fraction_index": None} for _ in shapes)
        else:
            dicts = [_blur_dict(rs, (k + j) % 3, (2 * k + 3 * j) % 6, not (j == 1 and k % 2 == 1)) for j in range(len(shapes))]
            if kind == "blur_label" and k % 2 == 0:
                dicts[0]["blur_est_label"] = (k // 2 + 1) % 4
            dicts = tuple(dicts)
        loader.append((images, targets, dicts))
    loader.dataset = object()
    return loader
