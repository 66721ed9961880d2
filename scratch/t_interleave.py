"""Timing experiment (results are NOT valid blurs: bands lose their halo rows): does the ORDER in which the launch's images reach the
dispatcher matter?  The BASELINE batch cut into 4 row bands per image (224 + 3 x 192 rows: whole tile rows), 32 'images' in one launch:
  image-major   band 0..3 of image 0 (heaviest PSF), then of image 1, ...   (= the shipped order, as a control for the cutting)
  interleaved   band 0 of images 0..7, band 1 of images 0..7, ...           (every stretch of the launch holds all eight PSFs)
  light-first   image-major from the lightest PSF to the heaviest"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from detectinblur_amd import blur_ops
dev = torch.device("cuda", 0)
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev)
tables = blur_ops.compact_psfs(psfs, normalize=True)
idx = sorted(range(8), key=lambda k: -dicts[k]["psf_taps"])
cuts = [(0, 224), (224, 416), (416, 608), (608, 800)]
bands = {(k, b): images[k][:, y0:y1, :].contiguous() for k in range(8) for b, (y0, y1) in enumerate(cuts)}
orders = {"whole images, heaviest first (shipped)": [(k, None) for k in idx],
          "bands, image-major": [(k, b) for k in idx for b in range(4)],
          "bands, interleaved": [(k, b) for b in range(4) for k in idx],
          "bands, light-first": [(k, b) for k in reversed(idx) for b in range(4)]}
for rnd in range(3):
    for name, order in orders.items():
        imgs = [images[k] if b is None else bands[(k, b)] for k, b in order]
        tix = [k for k, b in order]
        for _ in range(20): blur_ops.sparse_blur(list(imgs), tix, tables)
        # device time: 20 launches captured in a HIP graph and replayed (the eager loop is host-bound with 32 output tensors per call)
        g, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
        with torch.cuda.stream(side):
            blur_ops.sparse_blur(list(imgs), tix, tables)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=side):
                for _ in range(20): keep = blur_ops.sparse_blur(list(imgs), tix, tables)
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): g.replay()
            e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / 200 * 1e3)
        print("round %d  %-42s %.2f us per launch (graph replay)" % (rnd, name, sorted(ts)[3]), flush=True)
        del g, keep
