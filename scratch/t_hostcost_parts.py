"""Where the host time of one eager blur step goes (3 x 70 x 70 images: the GPU is never the limit): wall time of the two
C calls (ctypes + HIP launch), of compact_psfs / sparse_blur around them, and of blur_image_list around those."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, bench
from detectinblur_amd import blur_ops, _lib
from detectinblur_amd.models import blur_functions as BF
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev)
tiny = [torch.rand(3, 70, 70, device=dev).half() for _ in images]
acc = {}
def wrap(obj, name, key):
    fn = getattr(obj, name)
    def w(*a, **k):
        t0 = time.perf_counter_ns()
        r = fn(*a, **k)
        acc[key] = acc.get(key, 0) + time.perf_counter_ns() - t0
        return r
    setattr(obj, name, w)
class LibProxy:
    def __init__(self, l): self._l = l
    def __getattr__(self, n):
        f = getattr(self._l, n)
        if n in ("dib_psf_compact_list", "dib_sparse_blur"):
            def w(*a):
                t0 = time.perf_counter_ns(); r = f(*a); acc["C:" + n] = acc.get("C:" + n, 0) + time.perf_counter_ns() - t0; return r
            return w
        return f
real = _lib.lib()
proxy = LibProxy(real)
_lib.lib = lambda: proxy
wrap(blur_ops, "compact_psfs", "py:compact_psfs (incl. C)")
wrap(blur_ops, "sparse_blur", "py:sparse_blur (incl. C)")
wrap(torch, "empty", "torch.empty")
def step():
    batch = list(tiny)
    BF.blur_image_list(batch, dicts, psfs)
for _ in range(500): step()
torch.cuda.synchronize(); acc.clear()
N = 4000
t0 = time.perf_counter_ns()
for _ in range(N): step()
tot = time.perf_counter_ns() - t0
torch.cuda.synchronize()
print("step (instrumented): %.1f us" % (tot / N / 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-34s %.1f us" % (k, v / N / 1e3))

# ---- sections of sparse_blur / compact_psfs, replicated with timers ----------------------------------------------------
import ctypes
tabs = blur_ops.compact_psfs(psfs, True)
sec = {}
def T(key, t0):
    t1 = time.perf_counter_ns(); sec[key] = sec.get(key, 0) + t1 - t0; return t1
n = len(tiny); tix = list(range(n)); dt = tiny[0].dtype
for it in range(4000):
    t = time.perf_counter_ns()
    outs = list(tiny); act = [i for i in range(n) if tix[i] >= 0]; first = tiny[0]; d, dev_ = first.dtype, first.device
    t = T("sb: prologue", t)
    ins_p, outs_p, Cs, Hs, Ws = [None] * n, [None] * n, [0] * n, [0] * n, [0] * n
    srcs, uniform, shp = [], True, first.shape
    for i in act:
        img = tiny[i]
        if not img.is_cuda: raise RuntimeError
        if img.dtype != d: raise TypeError
        if not img.is_contiguous(): img = img.contiguous()
        sh = img.shape
        if sh != shp: uniform = False
        Cs[i], Hs[i], Ws[i] = sh
        ins_p[i] = img.data_ptr()
        srcs.append(img)
    t = T("sb: per-image loop (8)", t)
    block = torch.empty((n,) + tuple(shp), dtype=d, device=dev_)
    t = T("sb: torch.empty", t)
    base, step_ = block.data_ptr(), block.stride(0) * block.element_size()
    parts = block.unbind(0); outs = list(parts); outs_p = [base + k * step_ for k in range(n)]
    t = T("sb: unbind + pointers", t)
    a = (_lib.ptr_array(ins_p), _lib.ptr_array(outs_p), _lib.int_array(Cs), _lib.int_array(Hs), _lib.int_array(Ws), _lib.int_array(tix))
    t = T("sb: 6 ctypes arrays", t)
    s_ = blur_ops._stream(dev_); bp = tabs.buf.data_ptr()
    t = T("sb: stream + table ptr", t)
    rc = real.dib_sparse_blur(a[0], a[1], a[2], a[3], a[4], a[5], n, 1, bp, tabs.count, tabs.K, 0, s_)
    t = T("sb: C call (ctypes + launch)", t)
    # compact_psfs
    keep, ptrs, want = [], [], psfs[0].shape
    for p in psfs:
        if p.shape != want or p.dtype != dt or not p.is_cuda: raise ValueError
        if not p.is_contiguous(): p = p.contiguous()
        a_ = p.data_ptr()
        keep.append(p); ptrs.append(a_)
    t = T("cp: per-PSF loop (8)", t)
    tb = blur_ops.TapTables(128, n, dev_)
    t = T("cp: TapTables (torch.empty)", t)
    pa = _lib.ptr_array(ptrs); s_ = blur_ops._stream(dev_)
    t = T("cp: array + stream", t)
    rc = real.dib_psf_compact_list(pa, 1, n, 128, 1, tb.buf.data_ptr(), s_)
    t = T("cp: C call (ctypes + launch)", t)
torch.cuda.synchronize()
for k, v in sec.items():
    print("  %-34s %.2f us" % (k, v / 4000 / 1e3))
