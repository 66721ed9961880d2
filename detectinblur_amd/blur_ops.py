"""Host-side plumbing between torch tensors and the C ABI (include/dib.h).

Everything here is stream-ordered on torch's current stream and never synchronises the host:
the reference's per-tap `.item()` syncs (models/blur_functions.py:66-67) have no counterpart.
"""
import torch

from . import _lib

_DT = {torch.float16: _lib.DIB_F16, torch.float32: _lib.DIB_F32}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _require_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU: detectinblur_amd has no CPU path" % what)


def table_words(K):
    return _lib.lib().dib_tap_table_bytes(K) // 4


class TapTables:
    """Device-resident tap tables for a batch of PSFs (see include/dib.h, `Tap tables`)."""

    def __init__(self, K, count, device):
        self.K, self.count = K, count
        self.words = table_words(K)
        if self.words == 0:
            raise ValueError("PSF must be 128 or 256 wide, got %d" % K)
        # `count` tables + the scheduler trailer (zeroed by dib_psf_compact)
        self.buf = torch.empty(_lib.lib().dib_tap_tables_bytes(K, count) // 4, dtype=torch.int32, device=device)

    def ptr(self, i=0):
        return self.buf.data_ptr() + 4 * self.words * i

    def header(self, i):
        """(ntaps, rmin, rmax, cmin, cmax) -- a D2H copy; for tests and debugging only."""
        return self.buf[i * self.words:i * self.words + 5].tolist()

    def segments(self, i):
        """[(first_tap, end_tap, r_first, r_last, cmin, cmax)] of table i -- tests only."""
        nseg = int(self.buf[i * self.words + 7].item())
        off = i * self.words + ((8 + self.K + 1 + 3) & ~3) + 2 * self.K * self.K
        t = self.buf[off:off + 4 * nseg].cpu().view(-1, 4).tolist()
        return [(a, b, c >> 8, c & 255, d >> 8, d & 255) for a, b, c, d in t]

    def ltaps(self, i):
        """per-tap (lds_byte_offset, weight_bits) of table i -- tests only."""
        n = self.header(i)[0]
        off = i * self.words + ((8 + self.K + 1 + 3) & ~3) + 6 * self.K * self.K
        t = self.buf[off:off + n].cpu()
        return t & 0xffff, (t >> 16) & 0xffff

    def taps(self, i):
        """(rows, cols, weight_bits) of table i as CPU tensors -- tests only."""
        h = self.header(i)
        off = i * self.words + ((8 + self.K + 1 + 3) & ~3)
        t = self.buf[off:off + 2 * h[0]].cpu().view(-1, 2)
        rc = t[:, 0]
        return (rc >> 8) & 0xff, rc & 0xff, t[:, 1]


def compact_psfs(psfs, normalize):
    """psfs: list of K x K tensors (same K, same dtype) or one [B,K,K] tensor -> TapTables.
    A list is passed as device pointers (no stacking copy)."""
    l = _lib.lib()
    if isinstance(psfs, (list, tuple)):
        first = psfs[0]
        _require_cuda(first, "PSF")
        K, dt = first.shape[0], first.dtype
        if dt not in _DT:
            raise TypeError("PSF dtype %s not supported (float16 / float32)" % dt)
        keep = []
        for p in psfs:
            if p.dim() != 2 or p.shape[0] != K or p.shape[1] != K or p.dtype != dt or not p.is_cuda:
                raise ValueError("all PSFs of one call must be K x K CUDA tensors of one dtype")
            p = p.contiguous()
            if p.data_ptr() % 16:
                p = p.clone()
            keep.append(p)
        tabs = TapTables(K, len(keep), first.device)
        tabs._pin = keep   # alive until the tables die
        _lib.check(l.dib_psf_compact_list(_lib.ptr_array([p.data_ptr() for p in keep]), _DT[dt], len(keep), K,
                                          int(bool(normalize)), tabs.buf.data_ptr(), _stream()))
        return tabs
    stack = psfs
    _require_cuda(stack, "PSF")
    stack = stack.contiguous()
    if stack.dim() != 3 or stack.shape[1] != stack.shape[2]:
        raise ValueError("expected [B,K,K] PSFs, got %s" % (tuple(stack.shape),))
    if stack.dtype not in _DT:
        raise TypeError("PSF dtype %s not supported (float16 / float32)" % stack.dtype)
    B, K = stack.shape[0], stack.shape[1]
    tabs = TapTables(K, B, stack.device)
    _lib.check(l.dib_psf_compact(stack.data_ptr(), _DT[stack.dtype], B, K, int(bool(normalize)),
                                 tabs.buf.data_ptr(), _stream()))
    return tabs


def sparse_blur(images, table_index, tables, acc_mode=_lib.DIB_ACC_BITEXACT):
    """images: list of C x H x W (or H x W) tensors, all one dtype; table_index[i] < 0 skips image i.
    Returns the list of outputs (new tensors for blurred entries, the input tensor otherwise)."""
    n = len(images)
    outs, ins_p, outs_p, Cs, Hs, Ws = list(images), [], [], [], [], []
    dt = None
    for i, img in enumerate(images):
        if table_index[i] < 0:
            ins_p.append(None); outs_p.append(None); Cs.append(0); Hs.append(0); Ws.append(0)
            continue
        _require_cuda(img, "image")
        if img.dtype not in _DT:
            raise TypeError("image dtype %s not supported (float16 / float32)" % img.dtype)
        if dt is None:
            dt = img.dtype
        elif dt != img.dtype:
            raise TypeError("all images of one call must share a dtype")
        src = img.contiguous()
        shape3 = (1,) + tuple(src.shape) if src.dim() == 2 else tuple(src.shape)
        if len(shape3) != 3:
            raise ValueError("image must be C x H x W, got %s" % (tuple(img.shape),))
        out = torch.empty_like(src)
        outs[i] = out
        ins_p.append(src.data_ptr()); outs_p.append(out.data_ptr())
        Cs.append(shape3[0]); Hs.append(shape3[1]); Ws.append(shape3[2])
        images[i] = src  # keep alive until the launch below
    if dt is None:
        return outs
    _lib.check(_lib.lib().dib_sparse_blur(_lib.ptr_array(ins_p), _lib.ptr_array(outs_p), _lib.int_array(Cs),
                                          _lib.int_array(Hs), _lib.int_array(Ws), _lib.int_array(list(table_index)),
                                          n, _DT[dt], tables.buf.data_ptr(), tables.count, tables.K, acc_mode,
                                          _stream()))
    return outs


# One-entry cache: the engine calls blur_image_list and then expand_targets with the same PSF
# tensors (reference engine.py:101-105); the second call reuses the tables of the first instead
# of re-running normalise + nonzero as the reference does (utils.py:372-374).
_cache = {"key": None, "tables": None}


def invalidate_cache():
    _cache["key"] = _cache["tables"] = None
    _cache.pop("pin", None)


def compact_psfs_cached(psfs, normalize):
    key = (bool(normalize),) + tuple((p.data_ptr(), p._version, p.dtype, tuple(p.shape)) for p in psfs)
    if _cache["key"] == key:
        return _cache["tables"]
    tabs = compact_psfs(psfs, normalize)
    _cache["key"], _cache["tables"] = key, tabs
    # keep the PSF tensors alive while cached so a recycled data_ptr cannot alias the key
    _cache["pin"] = list(psfs)
    return tabs


def expand_boxes(boxes, tables, index, H, W):
    """In place on an [N,4] float32 CUDA tensor (utils.py:376-392)."""
    _require_cuda(boxes, "boxes")
    if boxes.dtype != torch.float32 or boxes.dim() != 2 or boxes.shape[1] != 4 or not boxes.is_contiguous():
        raise ValueError("boxes must be a contiguous [N,4] float32 tensor")
    _lib.check(_lib.lib().dib_expand_boxes(boxes.data_ptr(), boxes.shape[0], tables.ptr(index), H, W, _stream()))


def clamp_boxes(boxes, H, W):
    _require_cuda(boxes, "boxes")
    if boxes.dtype != torch.float32 or boxes.dim() != 2 or boxes.shape[1] != 4 or not boxes.is_contiguous():
        raise ValueError("boxes must be a contiguous [N,4] float32 tensor")
    _lib.check(_lib.lib().dib_clamp_boxes(boxes.data_ptr(), boxes.shape[0], H, W, _stream()))


def rasterize_psfs(traj, fractions, canvas=256, center=True, out_n=None, want64=True, want16=True):
    """traj: [B, iters] complex128 CUDA tensor (or anything torch.as_tensor accepts);
    fractions: B python floats.  Returns (psf64 [B,n,n] float64 | None, psf16 [B,n,n] float16 | None).
    generate_PSF.py:31-83 + :106-123 + transforms.py:334-335 + engine.py:84 in two launches."""
    import ctypes
    traj = torch.as_tensor(traj)
    if traj.dtype != torch.complex128:
        raise TypeError("trajectory must be complex128")
    if not traj.is_cuda:
        traj = traj.cuda(non_blocking=True)
    traj = traj.contiguous()
    if traj.dim() == 1:
        traj = traj.unsqueeze(0)
    B, iters = traj.shape
    if len(fractions) != B:
        raise ValueError("one exposure fraction per trajectory")
    if out_n is None:
        out_n = 128 if (center and canvas == 256) else canvas
    l = _lib.lib()
    ws = torch.empty(l.dib_psf_rasterize_workspace_bytes(B, iters, canvas), dtype=torch.uint8, device=traj.device)
    p64 = torch.empty((B, out_n, out_n), dtype=torch.float64, device=traj.device) if want64 else None
    p16 = torch.empty((B, out_n, out_n), dtype=torch.float16, device=traj.device) if want16 else None
    fr = (ctypes.c_double * B)(*[float(f) for f in fractions])
    _lib.check(l.dib_psf_rasterize(torch.view_as_real(traj).data_ptr(), B, iters, fr, canvas, int(bool(center)), out_n,
                                   p64.data_ptr() if want64 else None, p16.data_ptr() if want16 else None,
                                   ws.data_ptr(), _stream()))
    return p64, p16
