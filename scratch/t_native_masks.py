"""Native-size ragged batch: device time of the blur (HIP graph replay) by the stride mask of the 1-D grid (which strides of 32 workgroups
of an XCD's list are walked backwards), all in one process, rounds interleaved.   python scratch/t_native_masks.py [mode]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from detectinblur_amd import blur_ops, _lib
MODE = {"bitexact": 0, "fma16": 2, "fast16": 3}[sys.argv[1] if len(sys.argv) > 1 else "bitexact"]
dev = torch.device("cuda", 0)
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev)
tables = blur_ops.compact_psfs(psfs, normalize=True, vruns=True)
idx = sorted(range(8), key=lambda k: -dicts[k]["psf_taps"])
native = [torch.rand(3, h, w, generator=torch.Generator().manual_seed(31 + i)).half().to(dev) for i, (h, w) in enumerate(bench.COCO_NATIVE_SIZES)]
ordered = [native[k] for k in idx]
l = _lib.lib(); l.dib_debug_set_flat_mask.argtypes = [ctypes.c_int]
masks = {"none": 0, "all rev": 0b1111111, "snake(last rev)": 0b0101010, "snake(first rev)": 0b1010101, "first4 rev": 0b0001111, "last3 rev": 0b1110000, "first2 rev": 0b0000011}
graphs = {}
for name, m in masks.items():
    l.dib_debug_set_flat_mask(m)
    g, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.stream(side):
        blur_ops.sparse_blur(list(ordered), idx, tables, MODE); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            keep = [blur_ops.sparse_blur(list(ordered), idx, tables, MODE) for _ in range(20)]
    graphs[name] = (g, keep)
res = {n: [] for n in masks}
for rnd in range(9):
    for name, (g, _) in graphs.items():
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): g.replay()
        e1.record(); e1.synchronize()
        res[name].append(e0.elapsed_time(e1) / 200 * 1e3)
for name in masks:
    v = sorted(res[name]); print("%-18s mask %3d: median %.2f us (min %.2f max %.2f)" % (name, masks[name], v[4], v[0], v[-1]))
