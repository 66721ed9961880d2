"""Throughput of the DataLoader alone (no GPU work): images/s for f32 vs f16 hand-off, pinning on/off, 4/8/16 workers."""
import contextlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from detectinblur_amd import utils
from detectinblur_amd.coco_utils import SyntheticCocoDetection
from detectinblur_amd.train import _seed_worker, get_transform
torch.cuda.init()
with contextlib.redirect_stdout(sys.stderr):
    tf = get_transform(True, blur=True, blur_type=0.005, blur_ratio=0.75, low_exposure=True)
ds = SyntheticCocoDetection(num_images=8 * 24, size=(800, 1333), transforms=tf)
t0 = time.perf_counter(); [ds[i] for i in range(8)]; print("one item in-process: %.1f ms" % ((time.perf_counter() - t0) / 8 * 1e3))


def half_collate(batch):
    imgs, tg, bd = utils.collate_fn(batch)
    return tuple(i.half() for i in imgs), tg, bd


for name, coll in (("f32", utils.collate_fn), ("f16", half_collate)):
    for pin in (True, False):
        for nw in (4, 8, 16):
            loader = torch.utils.data.DataLoader(ds, batch_size=8, shuffle=False, drop_last=True, num_workers=nw, collate_fn=coll,
                                                 pin_memory=pin, worker_init_fn=_seed_worker)
            it = iter(loader)
            for _ in range(4):
                next(it)
            t0 = time.perf_counter()
            n = 0
            for b in it:
                n += len(b[0])
            el = time.perf_counter() - t0
            print("%s pin=%d workers=%2d: %.1f images/s (%.1f ms per batch of 8)" % (name, pin, nw, n / el, el / (n / 8) * 1e3), flush=True)
