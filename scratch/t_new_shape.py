"""First-encounter cost of a batch shape the shipped find-db does not hold (real COCO batches: ~30 padded shapes), by MIOpen find mode:
    python scratch/t_new_shape.py  -> child processes with MIOPEN_FIND_MODE unset / 2 (FAST) / 3, sizes 800x1216 and 1088x800 at b = 8."""
import os, subprocess, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("T_CHILD"):
    import torch
    from detectinblur_amd import kernel_choices, utils
    from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
    kernel_choices.use_shipped_kernel_choices()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = fasterrcnn_resnet50_fpn(num_classes=91, pretrained=False, pretrained_backbone=False).to(dev).train()
    opt = utils.make_sgd([p for p in model.parameters() if p.requires_grad], 0.0004, 0.9, 1e-4)
    out = {}
    for (H, W) in ((800, 1333), (800, 1216), (1088, 800)):
        g = torch.Generator().manual_seed(1)
        imgs = [torch.rand(3, H, W, generator=g).to(dev) for _ in range(8)]
        tg = [{"boxes": torch.tensor([[10.0, 20.0, 300.0, 400.0], [200.0, 100.0, 700.0, 600.0]], device=dev), "labels": torch.tensor([3, 7], device=dev)} for _ in range(8)]
        ts = []
        for it in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            loss = sum(model(list(imgs), [dict(t) for t in tg]).values())
            opt.zero_grad(); loss.backward(); opt.step()
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        out["%dx%d" % (H, W)] = {"first_step_s": round(ts[0], 2), "second_s": round(ts[1], 3), "steady_ms": round(1e3 * sorted(ts[2:])[1], 1)}
    out["report"] = {k: kernel_choices.report()[k] for k in ("miopen_db_growth_bytes", "miopen_foreign_files")}
    print(json.dumps(out))
else:
    for mode in (None, "2", "3", "5"):
        env = dict(os.environ, T_CHILD="1")
        if mode: env["MIOPEN_FIND_MODE"] = mode
        r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True)
        print("MIOPEN_FIND_MODE=%s: %s" % (mode, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]), flush=True)
