import sys, ctypes
sys.path.insert(0, '.')
import torch, bench
from detectinblur_amd import blur_ops, _lib
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, torch.device("cuda", 0))
stack = torch.stack(psfs).contiguous()
l = _lib.lib()
tabs = blur_ops.TapTables(128, 8, stack.device)
st = torch.cuda.current_stream().cuda_stream
for flag, name in ((1, "full"), (0, "no normalise"), (5, "normalise, no segmentation"), (4, "neither")):
    for _ in range(5): l.dib_psf_compact(stack.data_ptr(), 0, 8, 128, flag, tabs.buf.data_ptr(), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): l.dib_psf_compact(stack.data_ptr(), 0, 8, 128, flag, tabs.buf.data_ptr(), st)
    e1.record(); e1.synchronize()
    print("%-28s %.2f us per launch (back-to-back)" % (name, e0.elapsed_time(e1) / 200 * 1e3))
