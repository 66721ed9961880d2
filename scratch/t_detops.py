import sys, time, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
from detectinblur_amd import _lib
from detectinblur_amd.models import detector_ops as ops
l = _lib.lib()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
rs = np.random.RandomState(0)
for n in (1000, 4000, 8000):
    c = rs.uniform(0, 1300, (n, 2)); s = rs.uniform(16, 300, (n, 2))
    boxes = torch.tensor(np.concatenate([c - s / 2, c + s / 2], 1), dtype=torch.float32).cuda()
    scores = torch.rand(n).cuda()
    print("nms n=%d: %.1f us, kept %d" % (n, timeit(lambda: ops.nms(boxes, scores, 0.7)), len(ops.nms(boxes, scores, 0.7))))
for (H, W, scale, lo, hi) in ((200, 336, 0.25, 30, 112), (100, 168, 0.125, 112, 224), (50, 84, 1 / 16., 224, 448)):
    K, C = 1024, 256
    feat = torch.randn(8, C, H, W, device="cuda", requires_grad=True)
    cx = rs.uniform(0, 1333, K); cy = rs.uniform(0, 800, K); w = rs.uniform(lo, hi, K); h = rs.uniform(lo, hi, K)
    rois = torch.tensor(np.stack([rs.randint(0, 8, K), cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1), dtype=torch.float32).cuda()
    gout = torch.randn(K, C, 7, 7, device="cuda")
    res = {}
    for v in (0, 1):
        l.dib_debug_set_roi_bwd_variant(v)
        def f():
            g = torch.zeros_like(feat)
            _lib.check(l.dib_roi_align_backward(gout.data_ptr(), rois.data_ptr(), K, C, H, W, ctypes.c_float(scale), 7, 2, 0, g.data_ptr(), torch.cuda.current_stream().cuda_stream))
            return g
        res[v] = f()
        print("roi bwd %dx%d variant %d: %.1f us (incl. zero-fill)" % (H, W, v, timeit(f)))
    print("  max diff between variants %.3g" % (res[0] - res[1]).abs().max().item())
