"""Where the train_e2e line loses its ~7 % against the resident train_step: same model, same engine.train_one_epoch,
(a) a list of ready pinned batches (no loader threads), (b) the DataLoader with 8 workers, (c) resident step() of bench.py."""
import contextlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench as B
from detectinblur_amd import engine, utils
from detectinblur_amd.coco_utils import SyntheticCocoDetection
from detectinblur_amd.train import _seed_worker, get_transform

dev = torch.device("cuda", 0)
host = B.make_psfs_host(0)
images, dicts, psfs, _, _ = B.make_workload(0, dev, host)
tr, ddp, opt = B.train_step_bench(images, dicts, psfs, dev, None, 1, 0, 8, 3, account=False)
print("resident", tr["ms_per_step"], file=sys.stderr)
with contextlib.redirect_stdout(sys.stderr):
    tf = get_transform(True, blur=True, blur_type=0.005, blur_ratio=0.75, low_exposure=True)
W, K = 4, 16
ds = SyntheticCocoDetection(num_images=8 * (W + K), size=(800, 1333), transforms=tf)


class L(list):
    dataset = None


def run(loader, tag):
    timed = B._Stamped(loader, W)
    with contextlib.redirect_stdout(sys.stderr):
        engine.train_one_epoch(ddp, opt, timed, dev, epoch=1, print_freq=10 ** 9, blur_train=True, early_stop=None, gpu_blur=True,
                               expand_target_boxes=True)
    torch.cuda.synchronize()
    print(tag, (time.perf_counter() - timed.t0) / K * 1e3, "ms/step", file=sys.stderr)


ready = L()
for b in range(W + K):
    batch = utils.collate_fn([ds[8 * b + i] for i in range(8)])
    batch = (tuple(i.pin_memory() for i in batch[0]),) + batch[1:]
    ready.append(batch)
run(ready, "list of pinned batches")
run(ready, "list of pinned batches (again)")
for nw in (8, 4):
    loader = torch.utils.data.DataLoader(ds, batch_size=8, shuffle=False, drop_last=True, num_workers=nw, collate_fn=utils.collate_fn,
                                         pin_memory=True, worker_init_fn=_seed_worker)
    run(loader, "DataLoader %d workers" % nw)
# host-only cost of a step: how far ahead of the GPU is the Python thread?
t0 = time.perf_counter()
timed = B._Stamped(ready, 0)
torch.cuda.synchronize()
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
with contextlib.redirect_stdout(sys.stderr):
    engine.train_one_epoch(ddp, opt, ready[:6], dev, epoch=1, print_freq=10 ** 9, blur_train=True, early_stop=None, gpu_blur=True, expand_target_boxes=True)
pr.disable()
pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(35)
