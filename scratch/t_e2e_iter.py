"""Per-iteration anatomy of the loader-fed training loop: wait for the batch (loader) vs the engine's own host time."""
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
tr, ddp, opt = B.train_step_bench(images, dicts, psfs, dev, None, 1, 0, 6, 3, account=False)
print("resident", tr["ms_per_step"], file=sys.stderr)
del images
with contextlib.redirect_stdout(sys.stderr):
    tf = get_transform(True, blur=True, blur_type=0.005, blur_ratio=0.75, low_exposure=True)
N = 24
ds = SyntheticCocoDetection(num_images=8 * N, size=(800, 1333), transforms=tf)


class Probe(object):
    def __init__(self, loader):
        self.loader, self.dataset, self.rows = loader, None, []

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        it = iter(self.loader)
        while True:
            t0 = time.perf_counter()
            try:
                b = next(it)
            except StopIteration:
                return
            t1 = time.perf_counter()
            self.rows.append([t0, t1])
            yield b


def half_collate(batch):
    imgs, tg, bd = utils.collate_fn(batch)
    return tuple(i.half() for i in imgs), tg, bd


for tag, nw, coll, pin in (("f32 8w pin", 8, utils.collate_fn, True), ("f16 8w pin", 8, half_collate, True), ("f32 16w pin", 16, utils.collate_fn, True),
                           ("f32 8w nopin", 8, utils.collate_fn, False)):
    loader = torch.utils.data.DataLoader(ds, batch_size=8, shuffle=False, drop_last=True, num_workers=nw, collate_fn=coll,
                                         pin_memory=pin, worker_init_fn=_seed_worker)
    p = Probe(loader)
    with contextlib.redirect_stdout(sys.stderr):
        engine.train_one_epoch(ddp, opt, p, dev, epoch=1, print_freq=10 ** 9, blur_train=True, early_stop=None, gpu_blur=True, expand_target_boxes=True)
    torch.cuda.synchronize()
    end = time.perf_counter()
    r = p.rows
    waits = [(b - a) * 1e3 for a, b in r]
    iters = [(r[k + 1][0] - r[k][1]) * 1e3 for k in range(len(r) - 1)]
    print(tag, "| wait for batch ms:", " ".join("%.0f" % w for w in waits), file=sys.stderr)
    print(tag, "| engine host ms   :", " ".join("%.0f" % w for w in iters), file=sys.stderr)
    print(tag, "| steady ms/step (its 6..): %.1f" % ((end - r[6][1]) / (len(r) - 6) * 1e3), file=sys.stderr)
