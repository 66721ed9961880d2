#!/usr/bin/env python3
"""Pins for the reference-owned parts of the detector rows.

Runs, in THIS container only, the reference's own files through oracle/ref_harness.py:
  * `models/faster_rcnn.py`     `fasterrcnn_resnet50_fpn(...)` with recording stand-ins for the torchvision classes
                                -> every constructor call it makes, argument by argument          (A14)
  * `models/generalized_rcnn.py` `GeneralizedRCNN.forward` with recording sub-modules -> call order, arguments,
                                returned dicts, error texts                                        (A13)
  * `engine.py`                 the three ensemble routers over an exhaustive small grid          (A17)
                                `train_one_epoch` on CPU (toy detector, 5 steps, with and without the blur path)
                                -> weights, losses, LR trajectory, TensorBoard scalars             (A15)
                                `evaluate` on CPU (toy detectors / estimator, recording evaluator)
                                -> detections handed to the evaluator, expanded ground truth, routes (A17)
  * `engine_blur_estimator.py`  `train_one_epoch` / `evaluate` on CPU (toy classifier; 16-way and LEHE labels, blur on / off,
                                the 800-pixel round trip, quantisation, block artefacts, crop batcher, early stop)  (f4)
and writes tests/golden/detector_pins.json + detector_pins.npz.  Inputs come from oracle/pin_inputs.py, which the
tests re-use; outputs are data.  Usage:  python oracle/gen_detector_pins.py
"""
import contextlib
import io
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import pin_inputs as PI  # noqa: E402
import ref_harness  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def gen_ctor(pins):
    frcnn, _ = ref_harness.load_detector()
    import models.net_transforms as ref_nt          # the reference's module object (sys.path holds the reference root)
    out = {}
    for name, kw in PI.detector_ctor_cases().items():
        log = []
        fakes = PI.detector_fakes(log)
        saved = {k: getattr(frcnn, k) for k in fakes if k != "GeneralizedRCNNTransform"}
        saved_nt = ref_nt.GeneralizedRCNNTransform
        try:
            for k, v in fakes.items():
                if k == "GeneralizedRCNNTransform":
                    ref_nt.GeneralizedRCNNTransform = v       # reference models/faster_rcnn.py:241 reads it from the module
                else:
                    setattr(frcnn, k, v)
            model = frcnn.fasterrcnn_resnet50_fpn(**kw)
        finally:
            for k, v in saved.items():
                setattr(frcnn, k, v)
            ref_nt.GeneralizedRCNNTransform = saved_nt
        out[name] = {"calls": log,
                     "model": {"backbone": PI.describe(model.backbone), "rpn": PI.describe(model.rpn),
                               "roi_heads": PI.describe(model.roi_heads), "transform": PI.describe(model.transform),
                               "warp_internally": bool(model.warp_internally), "has_warper": hasattr(model, "warper")}}
    pins["ctor"] = out


def gen_forward(pins):
    _, grcnn = ref_harness.load_detector()
    pins["forward"] = {name: PI.run_forward_case(grcnn.GeneralizedRCNN, case) for name, case in PI.forward_cases().items()}


def gen_routers(pins):
    eng = ref_harness.load_engine()
    idx4 = [0, 1, 2, 3]
    labels = ["n0", "n1", "n2", "n3"]                # distinguishable from list positions
    pins["router_oracle"] = [eng.get_network_index_to_use_oracle(b, labels) for b in PI.router_oracle_batches()]
    pins["router_estimator"] = [eng.get_network_index_to_use_blur_estimator(e, idx4) for e in PI.router_estimations()]
    pins["router_estimator_lehe"] = [eng.get_network_index_to_use_blur_estimator_LEHE(e, idx4) for e in PI.router_estimations()]


@contextlib.contextmanager
def _cpu_engine(eng, group=True):
    """What the reference's engine needs to run on a GPU-less host: a one-rank gloo group (its rank-0 test calls
    torch.distributed.get_rank() in distributed mode, engine.py:137) and a no-op `torch.cuda.synchronize`
    (engine.py:277).  `distributed_mode=True` selects its `.to(device)` branch instead of `.cuda()`."""
    import torch.distributed as dist
    made = False
    if group and not dist.is_initialized():
        f = tempfile.NamedTemporaryFile(delete=False)
        f.close()
        dist.init_process_group("gloo", init_method="file://" + f.name, rank=0, world_size=1)
        made = True
    saved = torch.cuda.synchronize
    torch.cuda.synchronize = lambda *a, **k: None
    try:
        yield
    finally:
        torch.cuda.synchronize = saved
        if made:
            dist.destroy_process_group()


def _run_train(eng, blur, poison_at=None, out_text=None, **kw):
    torch.manual_seed(0)
    np.random.seed(0)
    model = PI.ToyDetector(1)
    opt = torch.optim.SGD(model.parameters(), lr=0.04, momentum=0.9, weight_decay=1e-4)
    model.lr_probe = opt
    writer = PI.RecordingWriter()
    loader = PI.train_batches(blur, poison_at)
    losses = []
    hook = model.register_forward_hook(lambda m, i, o: losses.append({k: float(v.detach()) for k, v in o.items()}))
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            eng.train_one_epoch(model, opt, loader, torch.device("cpu"), epoch=kw.pop("epoch", 0), print_freq=2, writer=writer,
                                distributed_mode=True, blur_train=blur, gpu_blur=blur, expand_target_boxes=blur,
                                use_custom_image_norm=blur, **kw)
    finally:
        hook.remove()
        if out_text is not None:
            out_text.append(buf.getvalue())
            out_text.append(len(model.calls))
    return model, opt, writer, losses


def gen_train(pins, store):
    eng = ref_harness.load_engine()
    out = {}
    with _cpu_engine(eng):
        for name, blur, kw in (("plain", False, dict(early_stop=None)), ("blur", True, dict(early_stop=None)),
                               ("blur_epoch1", True, dict(early_stop=None, epoch=1)),
                               ("default_early_stop", False, dict()), ("early_stop_2", False, dict(early_stop=2))):
            model, opt, writer, losses = _run_train(eng, blur, **kw)
            for k, v in model.state_dict().items():
                store["train_%s_%s" % (name, k)] = v.numpy().copy()
            out[name] = {"steps": len(model.calls), "lr_seen_by_forward": [c["lr"] for c in model.calls],
                         "final_lr": opt.param_groups[0]["lr"], "losses": losses, "scalars": writer.scalars,
                         "calls": [{k: c[k] for k in ("thetas", "lambda1s", "lambda2s", "dtypes", "killWarp")} for c in model.calls]}
        # a step whose loss is not finite: the reference prints and leaves the process IN FRONT of that step's update (engine.py:145-148)
        torch.manual_seed(0)
        np.random.seed(0)
        text, code = [], None
        probe = {}
        saved_step = torch.optim.SGD.step

        def spy(self, *a, **k):          # the optimiser the run used, for its state at the exit
            probe["opt"] = self
            return saved_step(self, *a, **k)
        torch.optim.SGD.step = spy
        try:
            _run_train(eng, False, poison_at=2, out_text=text, early_stop=None)
        except SystemExit as e:
            code = e.code
        finally:
            torch.optim.SGD.step = saved_step
        opt = probe["opt"]
        params = [p for g in opt.param_groups for p in g["params"]]
        for i, p in enumerate(params):
            store["train_nonfinite_param%d" % i] = p.detach().numpy().copy()
            store["train_nonfinite_momentum%d" % i] = opt.state[p]["momentum_buffer"].numpy().copy()
        out["nonfinite"] = {"steps": text[1], "exit_code": code, "lines": [ln for ln in text[0].splitlines() if ln.startswith("Loss is")],
                            "final_lr": opt.param_groups[0]["lr"]}
    pins["train"] = out


def gen_eval(pins, store):
    eng = ref_harness.load_engine()
    import utils as ref_utils
    out = {}
    saved = (eng.CocoEvaluator, eng.get_coco_api_from_dataset, ref_utils.get_iou_types)
    eng.CocoEvaluator = PI.FakeCocoEvaluator
    ref_utils.get_iou_types = lambda model: ["bbox"]
    try:
        with _cpu_engine(eng, group=False):      # with a group its meters synchronise through device='cuda' (utils.py:498)
            for name, case in PI.eval_cases().items():
                loader = PI.eval_batches(case["blur"])
                coco = PI.FakeCoco(loader, extra_ann_for=(101,))
                eng.get_coco_api_from_dataset = lambda ds, coco=coco: coco
                model, ens, est = PI.build_eval_models(case)
                buf = io.StringIO()
                with contextlib.redirect_stdout(buf):
                    ret = eng.evaluate(model, loader, torch.device("cpu"), distributed_mode=True, ensemble_models=ens,
                                       blur_estimator=est, **case["kw"])
                ev = PI.FakeCocoEvaluator.last
                assert ret is ev
                for b, upd in enumerate(ev.updates):
                    for iid, o in upd.items():
                        for n, v in o.items():
                            store["eval_%s_%d_%d_%s" % (name, b, iid, n)] = v
                gt = {str(iid): [a["bbox"] for a in anns] for iid, anns in coco.imgToAnns.items()}
                routes = PI.routes_of(ens) if ens else None
                faulty = [ln for ln in buf.getvalue().splitlines() if ln.startswith("Number of Faulty boxes")]
                out[name] = {"evaluator_calls": ev.calls, "image_ids": ev.img_ids, "gt_bbox": gt,
                             "routes": routes,
                             "model_calls": [[{k: c[k] for k in ("thetas", "lambda1s", "lambda2s", "dtypes", "killWarp", "training")}
                                              for c in m.calls] for m in (ens or [model])],
                             "estimator_calls": est.calls if est is not None else None, "faulty_line": faulty}
    finally:
        eng.CocoEvaluator, eng.get_coco_api_from_dataset, ref_utils.get_iou_types = saved
    pins["eval"] = out


# ---- f4: the blur estimator's own loops --------------------------------------------------------------------------------

def _run_est_train(est, case):
    torch.manual_seed(0)
    np.random.seed(0)
    model = PI.ToyClassifier(case["classes"], 1)
    opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
    model.lr_probe = opt
    writer = PI.RecordingWriter()
    loader = PI.est_batches(case["kind"], train=True)
    losses = []
    criterion = torch.nn.CrossEntropyLoss()

    def crit(output, target):
        loss = criterion(output, target)
        losses.append({"loss": float(loss.detach()), "target": target.tolist(), "logits0": [float(v) for v in output[0].detach()]})
        return loss
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        est.train_one_epoch(model, opt, crit, loader, torch.device("cpu"), print_freq=2, writer=writer, distributed_mode=True,
                            **case["kw"])
    return model, opt, writer, losses


def gen_estimator(pins, store):
    """The reference's engine_blur_estimator.train_one_epoch / evaluate on CPU (toy classifier, seeded batches and blur dicts):
    weights, every loss and label vector, the LR trajectory, TensorBoard scalars; accuracies, predictions, targets and the
    printed summary of evaluate.  Its blur is its own copy of the roll loop, with the 800-pixel round trip of `resize_images`."""
    est = ref_harness.load_estimator_engine()
    eng = ref_harness.load_engine()
    out = {"train": {}, "eval": {}}
    with _cpu_engine(eng):
        for name, case in PI.est_train_cases().items():
            model, opt, writer, losses = _run_est_train(est, case)
            for k, v in model.state_dict().items():
                store["est_train_%s_%s" % (name, k)] = v.numpy().copy()
            out["train"][name] = {"steps": len(model.calls), "calls": model.calls, "final_lr": opt.param_groups[0]["lr"],
                                  "losses": losses, "scalars": writer.scalars}
    with _cpu_engine(eng, group=False):       # with a group its meters synchronise through device='cuda' (utils.py:498)
        for name, case in PI.est_eval_cases().items():
            torch.manual_seed(0)
            np.random.seed(0)
            model = PI.ToyClassifier(case["classes"], 2)
            loader = PI.est_batches(case["kind"], train=False)
            logits = []
            hook = model.register_forward_hook(lambda m, i, o: logits.append([float(v) for v in o[0].detach()]))
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                ret = est.evaluate(model, loader, torch.device("cpu"), distributed_mode=True, **case["kw"])
            hook.remove()
            rec = {"batches": len(model.calls), "calls": model.calls, "logits": logits,
                   "printed": [ln for ln in buf.getvalue().splitlines() if ln.startswith("Top ")]}
            if case["kw"].get("send_back_preds_targets"):
                acc, tg, pr = ret
                rec.update(accuracies=[float(a) for a in acc], targets=[int(t) for t in tg], preds=[int(p) for p in pr])
            else:
                rec.update(accuracies=[float(a) for a in ret])
            out["eval"][name] = rec
    pins["estimator"] = out


def main():
    pins, store = {}, {}
    # first: the reference's models/net_transforms.py binds `ImageList` at import time, and the estimator's crop
    # batcher inside engine.evaluate (engine.py:262-264) is that file's own class
    ref_harness.load_net_transforms()
    gen_ctor(pins)
    gen_forward(pins)
    gen_routers(pins)
    gen_train(pins, store)
    gen_eval(pins, store)
    gen_estimator(pins, store)
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "detector_pins.json"), "w") as f:
        json.dump(pins, f, indent=1)
    np.savez_compressed(os.path.join(OUT, "detector_pins.npz"), **store)
    print("ctor cases:", list(pins["ctor"]))
    print("forward cases:", {k: (v["error"][0] if v["error"] else "ok") for k, v in pins["forward"].items()})
    print("router grid:", len(pins["router_oracle"]), "oracle batches,", len(pins["router_estimator"]), "estimations")
    print("train:", {k: v["steps"] for k, v in pins["train"].items()})
    print("eval:", {k: (v["routes"], v["faulty_line"]) for k, v in pins["eval"].items()})
    print("estimator train:", {k: (v["steps"], round(v["losses"][-1]["loss"], 4)) for k, v in pins["estimator"]["train"].items()})
    print("estimator eval:", {k: (v["batches"], v["accuracies"]) for k, v in pins["estimator"]["eval"].items()})
    print("wrote", OUT)


if __name__ == "__main__":
    main()
