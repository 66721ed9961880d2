"""NEEDS THE BUILD OF COMMIT 6c83371 (`git show 6c83371:detectinblur_amd/csrc/dib_blur.hip`: blur_quad_f16_norm_kernel +
dib_debug_blur_normalized; removed from the product again, result in profiles/r4_fused_store.txt).
Round-4 experiment: the blur's store fused with the epilogue (float + normalise, planar fp32 batch) vs the two kernels
(blur -> fp16 images, dib_normalize_pad -> fp32 batch) on the BASELINE batch; bit-identity on the image region."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from detectinblur_amd import _lib, blur_ops, utils as U

dev = torch.device("cuda", 0)
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev)
tables = blur_ops.compact_psfs(psfs, normalize=True)
idx = sorted(range(8), key=lambda k: -dicts[k]["psf_taps"])
ordered = [images[k] for k in idx]
means, stds = U.get_norm_params(dicts, False)
means = np.asarray(means, dtype=np.float64)[idx]; stds = np.asarray(stds, dtype=np.float64)[idx]
l = _lib.lib()
l.dib_debug_blur_normalized.restype = ctypes.c_int
fp = ctypes.POINTER(ctypes.c_float)
l.dib_debug_blur_normalized.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int,
                                        ctypes.c_void_p, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
m32 = np.ascontiguousarray(means.astype(np.float32)); s32 = np.ascontiguousarray(stds.astype(np.float32))
out = torch.zeros(8, 3, 800, 1344, device=dev)
ins = _lib.ptr_array([t.data_ptr() for t in ordered]); ti = _lib.int_array(idx)
st = torch.cuda.current_stream().cuda_stream


out_cl = torch.zeros(8, 3, 800, 1344, device=dev).contiguous(memory_format=torch.channels_last)


def fused(k=0):
    _lib.check(l.dib_debug_blur_normalized(ins, out.data_ptr(), 800, 1333, ti, 8, tables.buf.data_ptr(), m32.ctypes.data_as(fp), s32.ctypes.data_as(fp), 800, 1344, 0, st))


def fused_cl(k=0):
    _lib.check(l.dib_debug_blur_normalized(ins, out_cl.data_ptr(), 800, 1333, ti, 8, tables.buf.data_ptr(), m32.ctypes.data_as(fp), s32.ctypes.data_as(fp), 800, 1344, 1, st))


blurred = blur_ops.sparse_blur(list(ordered), idx, tables)
want = blur_ops.normalize_pad(blurred, means, stds, 800, 1344, channels_last=False)
fused(); fused_cl()
torch.cuda.synchronize()
print("fused channels-last == two-kernel path on the image region:", torch.equal(out_cl[..., :1333], want[..., :1333]))
print("fused == two-kernel path on the image region:", torch.equal(out[..., :1333], want[..., :1333]))
for name, fn in (("blur (fp16 out)", lambda k: blur_ops.sparse_blur(list(ordered), idx, tables)),
                 ("normalize_pad planar", lambda k: blur_ops.normalize_pad(blurred, means, stds, 800, 1344, channels_last=False)),
                 ("normalize_pad channels-last", lambda k: blur_ops.normalize_pad(blurred, means, stds, 800, 1344, channels_last=True)),
                 ("fused blur + normalise, planar fp32 store (padding not written)", fused),
                 ("fused blur + normalise, channels-last fp32 store, channel-fastest tiles", fused_cl)):
    for _ in range(50):
        fn(0)
    ms = sorted(bench.kernel_time_ms(fn, 200) for _ in range(5))
    print("%-70s %.2f us (min %.2f max %.2f)" % (name, ms[2] * 1e3, ms[0] * 1e3, ms[-1] * 1e3))
