"""Where is the host during the multi-second first iteration of a loader-fed epoch?  faulthandler dumps every thread's
Python stack once a second while 3 iterations run."""
import contextlib, faulthandler, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench as B
from detectinblur_amd import engine, utils
from detectinblur_amd.coco_utils import SyntheticCocoDetection
from detectinblur_amd.train import _seed_worker, get_transform

dev = torch.device("cuda", 0)
host = B.make_psfs_host(0)
images, dicts, psfs, _, _ = B.make_workload(0, dev, host)
tr, ddp, opt = B.train_step_bench(images, dicts, psfs, dev, None, 1, 0, 3, 2, account=False)
print("resident", tr["ms_per_step"], file=sys.stderr)
with contextlib.redirect_stdout(sys.stderr):
    tf = get_transform(True, blur=True, blur_type=0.005, blur_ratio=0.75, low_exposure=True)
ds = SyntheticCocoDetection(num_images=8 * 6, size=(800, 1333), transforms=tf)
for rnd in range(2):
    loader = torch.utils.data.DataLoader(ds, batch_size=8, shuffle=False, drop_last=True, num_workers=8, collate_fn=utils.collate_fn,
                                         pin_memory=True, worker_init_fn=_seed_worker)
    t0 = time.perf_counter()
    faulthandler.dump_traceback_later(1.0, repeat=True, file=sys.stderr)
    with contextlib.redirect_stdout(sys.stderr):
        engine.train_one_epoch(ddp, opt, loader, dev, epoch=1, print_freq=1, blur_train=True, early_stop=None, gpu_blur=True, expand_target_boxes=True)
    torch.cuda.synchronize()
    faulthandler.cancel_dump_traceback_later()
    print("round", rnd, "took", time.perf_counter() - t0, file=sys.stderr)
