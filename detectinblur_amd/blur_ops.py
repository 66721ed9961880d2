"""Host-side plumbing between torch tensors and the C ABI (include/dib.h).

Everything here is stream-ordered on torch's current stream and never synchronises the host:
the reference's per-tap `.item()` syncs (models/blur_functions.py:66-67) have no counterpart.
"""
import torch

from . import _lib

_DT = {torch.float16: _lib.DIB_F16, torch.float32: _lib.DIB_F32}


def _reissuing(call):
    """Runs `call()` (a library entry point that launches a blur).  DIB_ETIMEOUT from it means that an EARLIER blur step on this
    device gave up waiting inside its single launch (include/dib.h, "Device status"): that step's images are incomplete, THIS
    call launched nothing, and the library has put the device on compaction + blur as two launches.  The process survives: the
    condition is reported as a RuntimeWarning carrying the library's text and the call is issued again."""
    rc = call()
    if rc == _lib.DIB_ETIMEOUT:
        import warnings
        warnings.warn(_lib.lib().dib_last_error().decode("utf-8", "replace"), RuntimeWarning, stacklevel=3)
        rc = call()
    return rc


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(device=None):
    """hipStream_t of torch's current stream as an int.  The raw accessor is ~30x cheaper than
    building a torch.cuda.Stream object (8 us), which matters for a 60 us step (`_lib.stream_of` is the same for a tensor)."""
    if _raw_stream is not None:
        idx = device.index if (device is not None and device.index is not None) else torch.cuda.current_device()
        return _raw_stream(idx)
    return torch.cuda.current_stream().cuda_stream


def _require_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU: detectinblur_amd has no CPU path" % what)


_TABLE_WORDS = {}


def table_words(K):
    w = _TABLE_WORDS.get(K)
    if w is None:
        w = _TABLE_WORDS[K] = _lib.lib().dib_tap_table_bytes(K) // 4
    return w


class TapTables:
    """Device-resident tap tables for a batch of PSFs (see include/dib.h, `Tap tables`)."""

    def __init__(self, K, count, device, large=False, vruns=False):
        self.K, self.count = K, count
        self.large = bool(large)      # compacted for the large LDS window (include/dib.h: DIB_COMPACT_LARGE_WINDOW): the blur must be told
        self.vruns = bool(vruns)      # carries the vertical-run groups DIB_ACC_FAST16 walks (DIB_COMPACT_VRUNS)
        self.words = table_words(K)
        if self.words == 0:
            raise ValueError("PSF must be 128 or 256 wide, got %d" % K)
        self.buf = torch.empty(self.words * count, dtype=torch.int32, device=device)     # == dib_tap_tables_bytes(K, count) / 4

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

    def ltaps(self, i, quad=False):
        """per-tap (lds_byte_offset, weight_bits) of table i, for the 96-word window rows of the 256-wide tiles or
        (quad) the 56-element rows of the default 128-wide tiles -- tests only."""
        n = self.header(i)[0]
        off = i * self.words + ((8 + self.K + 1 + 3) & ~3) + 6 * self.K * self.K + (self.K * self.K + 8 if quad else 0)
        t = self.buf[off:off + n].cpu()
        return t & 0xffff, (t >> 16) & 0xffff

    def vgroups(self, i):
        """The vertical-run groups of table i as the DIB_ACC_FAST16 tap loop walks them (include/dib.h): per segment a list of
        (lds_byte_offset, [weight bits of tap j = 0 .. n - 1]) -- tests only.  None when the table carries none."""
        K = self.K
        base = i * self.words
        if not (int(self.buf[base + 5].item()) >> 17) & 1:
            return None
        off = base + ((((8 + K + 1 + 3) & ~3) + 6 * K * K + 2 * (K * K + 8) + 3) & ~3)
        out = []
        for (t0, t1, _rf, _rl, _cmn, _cmx) in self.segments(i):
            rec = self.buf[off + 4 * t0:off + 4 * t1].cpu().view(-1, 4).tolist()
            own = rec[0][3] & 0xffffffff          # the first record's w: its own offset | (size - 1) << 16
            groups, g = [], 0
            while (own >> 16) & 7 != 4:
                n = ((own >> 16) & 7) + 1
                x, y, z, _w = (v & 0xffffffff for v in rec[g])
                groups.append((own & 0xffff, [y & 0xffff, y >> 16, z & 0xffff, z >> 16][:n]))
                own, g = x, g + 1
            out.append(groups)
        return out

    def taps(self, i):
        """(rows, cols, weight_bits) of table i as CPU tensors -- tests only."""
        h = self.header(i)
        off = i * self.words + ((8 + self.K + 1 + 3) & ~3)
        t = self.buf[off:off + 2 * h[0]].cpu().view(-1, 2)
        rc = t[:, 0]
        return (rc >> 8) & 0xff, rc & 0xff, t[:, 1]


def large_window_pays(blur_dicts, n_images):
    """Scheduling rule (never needed for correctness: both geometries give the same bits): compact for and blur with the
    LARGE LDS window (21 x 64 segments, 4 workgroups per CU) when the launch is small -- at most LARGE_WINDOW_MAX_IMAGES
    images, i.e. most of the chip's workgroup slots stay empty whatever is done -- and the refills saved outweigh the larger
    ones paid.  Measured at batch 1 on 3 x 800 x 1333 (scratch/t_large_window.py, profiles/r4_large_window.txt): a standard
    refill costs ~2.4 us, a large one ~5 us, and the large kernel ~3 us more per launch: a 16 x 48 full-exposure PSF is 19
    standard segments (two per row: segments are runs of row-major consecutive taps) and ONE large one, 79 -> 37 us; an
    85 x 48 PSF is 7 against 5, and stays on the standard window (51 against 63 us).  Needs the `psf_segments` hint
    BlurImage writes (transforms.count_tap_segments); without it the standard window is used."""
    if n_images > LARGE_WINDOW_MAX_IMAGES:
        return False
    std = large = 0
    for bd in blur_dicts:
        seg = bd.get("psf_segments") if bd.get("blurring") else None
        if seg is None:
            if bd.get("blurring"):
                return False
            continue
        std += seg[0]
        large += seg[1]
    return std > 0 and 2.4 * std > 5.0 * large + 6.0


LARGE_WINDOW_MAX_IMAGES = 2


def compact_psfs(psfs, normalize, large_window=False, vruns=False):
    """psfs: list of K x K tensors (same K, same dtype) or one [B,K,K] tensor -> TapTables.
    A list is passed as device pointers (no stacking copy).  `large_window`: segment the taps for the large LDS window of
    the default fp16 tiles (TapTables.large; sparse_blur then runs the large-window kernel).  `vruns`: also write the
    vertical-run groups that DIB_ACC_FAST16 walks (fp16 PSFs on the 128 canvas, standard window)."""
    l = _lib.lib()
    vruns = bool(vruns) and not large_window
    flags = int(bool(normalize)) | (_lib.DIB_COMPACT_LARGE_WINDOW if large_window else 0) | (_lib.DIB_COMPACT_VRUNS if vruns else 0)
    if isinstance(psfs, (list, tuple)):
        first = psfs[0]
        _require_cuda(first, "PSF")
        K, dt = first.shape[0], first.dtype
        if dt not in _DT:
            raise TypeError("PSF dtype %s not supported (float16 / float32)" % dt)
        keep, ptrs, want = [], [], first.shape
        if first.dim() != 2 or want[1] != K:
            raise ValueError("all PSFs of one call must be K x K CUDA tensors of one dtype")
        for p in psfs:
            if p.shape != want or p.dtype != dt or not p.is_cuda:
                raise ValueError("all PSFs of one call must be K x K CUDA tensors of one dtype")
            if not p.is_contiguous():
                p = p.contiguous()
            a = p.data_ptr()
            if a & 15:
                p = p.clone()
                a = p.data_ptr()
            keep.append(p)
            ptrs.append(a)
        tabs = TapTables(K, len(keep), first.device, large_window, vruns and K == 128 and dt == torch.float16)
        tabs._pin = keep   # alive until the tables die
        _lib.check(l.dib_psf_compact_list(_lib.ptr_array(ptrs), _DT[dt], len(keep), K, flags,
                                          tabs.buf.data_ptr(), _stream(first.device)))
        return tabs
    stack = psfs
    _require_cuda(stack, "PSF")
    stack = stack.contiguous()
    if stack.dim() != 3 or stack.shape[1] != stack.shape[2]:
        raise ValueError("expected [B,K,K] PSFs, got %s" % (tuple(stack.shape),))
    if stack.dtype not in _DT:
        raise TypeError("PSF dtype %s not supported (float16 / float32)" % stack.dtype)
    B, K = stack.shape[0], stack.shape[1]
    tabs = TapTables(K, B, stack.device, large_window, vruns and K == 128 and stack.dtype == torch.float16)
    _lib.check(l.dib_psf_compact(stack.data_ptr(), _DT[stack.dtype], B, K, flags,
                                 tabs.buf.data_ptr(), _stream()))
    return tabs


def _describe(images, table_index):
    """The per-image arrays of dib_sparse_blur / dib_blur_step for `images` (table_index[i] < 0 skips image i), with the
    outputs allocated: (outs, ins_p, outs_p, Cs, Hs, Ws, dtype, device, keep) or None when nothing is to be blurred.
    When every blurred image has the same shape the outputs are slices of ONE allocation.
    This loop is on the host's critical path (the eager step is ~15 us of interpreter in front of a 40 us kernel): one pass
    with as few tensor-attribute calls per image as the checks allow."""
    n = len(images)
    outs = list(images)
    act = [i for i in range(n) if table_index[i] >= 0]
    if not act:
        return None
    first = images[act[0]]
    dt, dev = first.dtype, first.device
    if dt not in _DT:
        raise TypeError("image dtype %s not supported (float16 / float32)" % dt)
    ins_p, outs_p, Cs, Hs, Ws = [None] * n, [None] * n, [0] * n, [0] * n, [0] * n
    srcs, uniform, shp = [], True, first.shape
    for i in act:
        img = images[i]
        if not img.is_cuda:
            raise RuntimeError("image must live on the GPU: detectinblur_amd has no CPU path")
        if img.dtype != dt:
            raise TypeError("all images of one call must share a dtype")
        if not img.is_contiguous():
            img = img.contiguous()
        sh = img.shape
        if sh != shp:
            uniform = False
        nd = len(sh)
        if nd == 3:
            Cs[i], Hs[i], Ws[i] = sh
        elif nd == 2:
            Cs[i] = 1
            Hs[i], Ws[i] = sh
        else:
            raise ValueError("image must be C x H x W, got %s" % (tuple(sh),))
        ins_p[i] = img.data_ptr()
        srcs.append(img)      # keeps a .contiguous() copy alive until the launch
    if uniform and len(act) > 1:
        block = torch.empty((len(act),) + tuple(shp), dtype=dt, device=dev)
        base, step = block.data_ptr(), block.stride(0) * block.element_size()
        parts = block.unbind(0)
        if len(act) == n:                      # nothing skipped: the outputs ARE the slices, in order
            outs = list(parts)
            outs_p = [base + k * step for k in range(n)]
        else:
            for k, i in enumerate(act):
                outs[i] = parts[k]
                outs_p[i] = base + k * step
    else:
        for i, src in zip(act, srcs):
            o = torch.empty_like(src)
            outs[i] = o
            outs_p[i] = o.data_ptr()
    return outs, ins_p, outs_p, Cs, Hs, Ws, dt, dev, srcs


def sparse_blur(images, table_index, tables, acc_mode=_lib.DIB_ACC_BITEXACT):
    """images: list of C x H x W (or H x W) tensors, all one dtype; table_index[i] < 0 skips image i.
    Returns the list of outputs (new tensors for blurred entries, the input tensor otherwise)."""
    d = _describe(images, table_index)
    if d is None:
        return list(images)
    outs, ins_p, outs_p, Cs, Hs, Ws, dt, dev, _keep = d
    _await(tables)
    if acc_mode == _lib.DIB_ACC_FAST16 and not tables.vruns:
        raise ValueError("DIB_ACC_FAST16 walks the tables' vertical-run groups: compact with compact_psfs(..., vruns=True) "
                         "(fp16 PSFs on the 128 canvas, standard window); without them the library would run DIB_ACC_FMA16's loop")
    if tables.large:
        acc_mode |= _lib.DIB_WINDOW_LARGE          # the tables hold the large window's segments and offsets
    args = (_lib.ptr_array(ins_p), _lib.ptr_array(outs_p), _lib.int_array(Cs), _lib.int_array(Hs), _lib.int_array(Ws),
            _lib.int_array(table_index), len(images), _DT[dt], tables.buf.data_ptr(), tables.count, tables.K, acc_mode, _stream(dev))
    fn = _lib.lib().dib_sparse_blur
    _lib.check(_reissuing(lambda: fn(*args)))
    return outs


def sparse_blur_normalized(images, table_index, tables, means, stds, Hp, Wp, channels_last=False, acc_mode=_lib.DIB_ACC_BITEXACT, order=None):
    """sparse_blur + normalize_pad as ONE launch (dib_sparse_blur_normalized): the fp32 batch [B,3,Hp,Wp] holding
    (blurred - mean) / std inside each image and 0 in the padding, or None when the library does not serve the batch that way
    (an image with table_index < 0, a padded extent the image's tiles do not cover, the large window, ...): the caller then
    blurs and normalises in two launches.  images: 3 x H x W float16 CUDA tensors; means / stds: [B,3] rows.  `order`: the
    sequence in which the images are handed to the launch (heaviest PSF first lets it end on its cheapest tiles); the batch
    position of every image stays its list position.  Bit-identical to the two launches."""
    import ctypes
    import numpy as np
    B = len(images)
    if B == 0 or tables.large:
        return None
    first = images[0]
    for img in images:
        if not (img.is_cuda and img.dtype == torch.float16 and img.dim() == 3 and img.shape[0] == 3):
            return None
    if any(t < 0 for t in table_index):
        return None
    _await(tables)
    seq = list(order) if order is not None else list(range(B))
    keep = [images[i] if images[i].is_contiguous() else images[i].contiguous() for i in seq]
    m = np.ascontiguousarray(np.asarray(means, dtype=np.float64).reshape(B, 3).astype(np.float32)[seq])
    sd = np.ascontiguousarray(np.asarray(stds, dtype=np.float64).reshape(B, 3).astype(np.float32)[seq])
    fmt = torch.channels_last if channels_last else torch.contiguous_format
    out = torch.empty((B, 3, Hp, Wp), dtype=torch.float32, device=first.device, memory_format=fmt)
    fp = ctypes.POINTER(ctypes.c_float)
    rc = _lib.lib().dib_sparse_blur_normalized(_lib.ptr_array([k.data_ptr() for k in keep]), _lib.int_array([int(k.shape[1]) for k in keep]),
                                               _lib.int_array([int(k.shape[2]) for k in keep]), _lib.int_array([table_index[i] for i in seq]),
                                               _lib.int_array(seq), B, tables.buf.data_ptr(), tables.count, tables.K, acc_mode,
                                               m.ctypes.data_as(fp), sd.ctypes.data_as(fp), out.data_ptr(), Hp, Wp, int(bool(channels_last)),
                                               _stream(first.device))
    if rc == 1:
        return None
    _lib.check(rc)
    return out


def blur_step(images, table_index, psfs, normalize=True, acc_mode=_lib.DIB_ACC_BITEXACT, psfs_complete=False, large_window=False):
    """compact_psfs(psfs) + sparse_blur(images, table_index, tables) behind ONE library call (dib_blur_step), the tables
    in library-owned buffers (two per stream, alternating).  psfs: list of K x K CUDA tensors of one dtype, 16-byte
    aligned and contiguous (anything else is copied first).  `psfs_complete`: the caller states that the PSF buffers are
    complete right now (not the product of work still queued on the current stream): the compaction is then launched
    without a barrier in front of it and overlaps the kernel queued before it -- the previous step's blur.  Under graph
    capture the tables come from the capture's own pool and both launches are ordinary.
    This function IS the host's share of the headline step (~25 us of interpreter in front of a 44 us pair of launches): one
    pass over the images, one over the PSFs, two marshalled arrays (dib_blur_step_packed)."""
    n = len(images)
    first = None
    for i in range(n):
        if table_index[i] >= 0:
            first = images[i]
            break
    if first is None:
        return list(images)
    dt, dev, shp = first.dtype, first.device, first.shape
    if dt not in _DT:
        raise TypeError("image dtype %s not supported (float16 / float32)" % dt)
    # ---- images: pointers and shapes; outputs of a uniform batch are slices of one allocation
    ins_p, Cs, Hs, Ws = [None] * n, [0] * n, [0] * n, [0] * n
    keep, uniform, act = [], True, 0
    for i in range(n):
        if table_index[i] < 0:
            continue
        img = images[i]
        if not img.is_cuda:
            raise RuntimeError("image must live on the GPU: detectinblur_amd has no CPU path")
        if img.dtype != dt:
            raise TypeError("all images of one call must share a dtype")
        if not img.is_contiguous():
            img = img.contiguous()
            keep.append(img)
        sh = img.shape
        if sh != shp:
            uniform = False
        if len(sh) == 3:
            Cs[i], Hs[i], Ws[i] = sh
        elif len(sh) == 2:
            Cs[i] = 1
            Hs[i], Ws[i] = sh
        else:
            raise ValueError("image must be C x H x W, got %s" % (tuple(sh),))
        ins_p[i] = img.data_ptr()
        act += 1
    outs = list(images)
    outs_p = [None] * n
    if uniform and act > 1:
        block = torch.empty((act,) + tuple(shp), dtype=dt, device=dev)
        base, step = block.data_ptr(), block.stride(0) * block.element_size()
        parts = block.unbind(0)
        if act == n:
            outs = list(parts)
            outs_p = [base + k * step for k in range(n)]
        else:
            k = 0
            for i in range(n):
                if table_index[i] >= 0:
                    outs[i] = parts[k]
                    outs_p[i] = base + k * step
                    k += 1
    else:
        for i in range(n):
            if table_index[i] >= 0:
                o = torch.empty(images[i].shape, dtype=dt, device=dev)
                outs[i] = o
                outs_p[i] = o.data_ptr()
    # ---- PSFs
    p0 = psfs[0]
    K, pdt, want = p0.shape[0], p0.dtype, p0.shape
    if pdt not in _DT:
        raise TypeError("PSF dtype %s not supported (float16 / float32)" % pdt)
    if len(want) != 2 or want[1] != K:
        raise ValueError("all PSFs of one call must be K x K CUDA tensors of one dtype")
    if table_words(K) == 0:
        raise ValueError("PSF must be 128 or 256 wide, got %d" % K)
    ptrs = []
    for p in psfs:
        if p.shape != want or p.dtype != pdt or not p.is_cuda:
            raise ValueError("all PSFs of one call must be K x K CUDA tensors of one dtype")
        if not p.is_contiguous():
            p = p.contiguous()
            keep.append(p)
            psfs_complete = False          # the copy was just queued on the current stream
        a = p.data_ptr()
        if a & 15:
            p = p.clone()
            keep.append(p)
            a = p.data_ptr()
            psfs_complete = False
        ptrs.append(a)
    if large_window and (dt != torch.float16 or acc_mode == _lib.DIB_ACC_FP32):
        large_window = False                # the large window serves the default fp16 tiles only
    flags = (_lib.DIB_STEP_PSFS_COMPLETE if psfs_complete else 0) | (_lib.DIB_STEP_LARGE_WINDOW if large_window else 0)
    l = _lib.lib()
    pa = _lib.ptr_array(ptrs + ins_p + outs_p)
    ia = _lib.int_array(Cs + Hs + Ws + list(table_index))
    stream = _stream(dev)
    rc = _reissuing(lambda: l.dib_blur_step_packed(pa, ia, _DT[pdt], len(ptrs), K, int(bool(normalize)), n, _DT[dt], acc_mode, None, flags, stream))
    if rc == _lib.DIB_ECAPTURE:            # the current stream is being captured: tables from the capture's pool
        tabs = TapTables(K, len(ptrs), dev, large_window)
        rc = l.dib_blur_step_packed(pa, ia, _DT[pdt], len(ptrs), K, int(bool(normalize)), n, _DT[dt], acc_mode, tabs.buf.data_ptr(),
                                    flags & _lib.DIB_STEP_LARGE_WINDOW, stream)
    _lib.check(rc)
    return outs


# ---- tap compaction off the critical path -------------------------------------------------------------------
# The compaction of a batch's PSFs (8 workgroups, ~5 us, a latency chain) depends on nothing but the PSFs.  Run on a
# side stream as soon as they are on the device, it overlaps whatever the main stream is still doing -- the previous
# batch's detector in a training loop, the previous batch's blur in bench.py -- and the blur only waits for its
# event.  Tables made this way carry `.ready`; sparse_blur / expand_boxes wait on it before they read the tables.
# Tables are plain objects owned by whoever made them (engine.py hands one set per batch to blur_image_list and
# expand_targets through `tables=`): there is no process-wide table cache, so a PSF buffer rewritten in place by a raw
# kernel (dib_psf_rasterize into a reused tensor) can never meet tables compacted from its previous contents.
_side_streams = {}


def side_stream(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _side_streams:
        _side_streams[idx] = torch.cuda.Stream(device=idx)
    return _side_streams[idx]


def compact_psfs_ahead(psfs, normalize, after_current=True, large_window=False, vruns=False):
    """compact_psfs on the device's side stream.  `after_current`: the side stream first waits for the work already
    queued on the current stream (needed when that work PRODUCES the PSFs, e.g. their host-to-device copy was issued
    on it); pass False when the PSFs are known to be complete (bench.py's resident PSFs), so that the compaction may
    overlap kernels still running on the current stream."""
    first = psfs[0] if isinstance(psfs, (list, tuple)) else psfs
    _require_cuda(first, "PSF")
    main = torch.cuda.current_stream(first.device)
    side = side_stream(first.device)
    if after_current:
        side.wait_stream(main)
    with torch.cuda.stream(side):
        tabs = compact_psfs(psfs, normalize, large_window, vruns)
        tabs.ready = torch.cuda.Event()
        tabs.ready.record(side)
    # the PSFs were allocated on the main stream and are read on the side stream: tell the caching allocator, or a
    # buffer freed right after this call could be handed out again while the compaction still reads it
    for p in (psfs if isinstance(psfs, (list, tuple)) else (psfs,)):
        p.record_stream(side)
    return tabs


def _await(tables):
    """Make the current stream wait for tables compacted on the side stream (no-op otherwise)."""
    ev = getattr(tables, "ready", None)
    if ev is not None:
        cur = torch.cuda.current_stream(tables.buf.device)
        cur.wait_event(ev)
        tables.buf.record_stream(cur)      # the buffer came from the side stream's pool


def expand_boxes(boxes, tables, index, H, W):
    """In place on an [N,4] float32 CUDA tensor (utils.py:376-392)."""
    _require_cuda(boxes, "boxes")
    if boxes.dtype != torch.float32 or boxes.dim() != 2 or boxes.shape[1] != 4 or not boxes.is_contiguous():
        raise ValueError("boxes must be a contiguous [N,4] float32 tensor")
    _await(tables)
    _lib.check(_lib.lib().dib_expand_boxes(boxes.data_ptr(), boxes.shape[0], tables.ptr(index), H, W, _stream()))


def clamp_boxes(boxes, H, W):
    _require_cuda(boxes, "boxes")
    if boxes.dtype != torch.float32 or boxes.dim() != 2 or boxes.shape[1] != 4 or not boxes.is_contiguous():
        raise ValueError("boxes must be a contiguous [N,4] float32 tensor")
    _lib.check(_lib.lib().dib_clamp_boxes(boxes.data_ptr(), boxes.shape[0], H, W, _stream()))


def post_ops(image, noise_var=None, block_scale=None):
    """noise + clamp and / or nearest-neighbour block artefacts on one C x H x W (or H x W) CUDA image, one launch
    (reference models/blur_functions.py:72-81).  The noise field's key is drawn from torch's HOST generator (reproducible
    under torch.manual_seed, no device synchronisation)."""
    _require_cuda(image, "image")
    if image.dtype not in _DT:
        raise TypeError("image dtype %s not supported (float16 / float32)" % image.dtype)
    img = image if image.is_contiguous() else image.contiguous()
    if img.dim() == 3:
        C, H, W = img.shape
    elif img.dim() == 2:
        C, (H, W) = 1, img.shape
    else:
        raise ValueError("image must be C x H x W, got %s" % (tuple(image.shape),))
    out = torch.empty_like(img)
    seed = int(torch.empty((), dtype=torch.int64).random_().item()) if noise_var is not None else 0
    _lib.check(_lib.lib().dib_post_ops(img.data_ptr(), out.data_ptr(), C, H, W, _DT[img.dtype], float(noise_var or 0.0), seed,
                                       float(block_scale or 0.0), _stream(img.device)))
    return out


def jpeg_roundtrip(image, q_luma, q_chroma):
    """One 3 x H x W CUDA image in [0, 1] (float16 / float32) through the JPEG round trip in one launch; q_luma / q_chroma:
    the 8 x 8 tables times the quality factor (anything numpy reads).  Returns a float16 CUDA tensor of the same shape
    (reference transforms.py:467-493 + models/jpeg/DiffJPEG.py)."""
    import ctypes
    import numpy as np
    _require_cuda(image, "image")
    if image.dtype not in _DT or image.dim() != 3 or image.shape[0] != 3:
        raise ValueError("jpeg_roundtrip needs a 3 x H x W float16 / float32 CUDA image")
    img = image if image.is_contiguous() else image.contiguous()
    out = torch.empty(img.shape, dtype=torch.float16, device=img.device)
    fp = ctypes.POINTER(ctypes.c_float)
    qy = np.ascontiguousarray(np.asarray(q_luma, dtype=np.float32).reshape(64))
    qc = np.ascontiguousarray(np.asarray(q_chroma, dtype=np.float32).reshape(64))
    _lib.check(_lib.lib().dib_jpeg_roundtrip(img.data_ptr(), out.data_ptr(), int(img.shape[1]), int(img.shape[2]), _DT[img.dtype],
                                             qy.ctypes.data_as(fp), qc.ctypes.data_as(fp), _stream(img.device)))
    return out


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


def normalize_pad(images, means, stds, Hp, Wp, channels_last=False, out_sizes=None):
    """images: list of 3 x H x W CUDA tensors (float16 or float32, one dtype); means / stds: [B,3] rows (anything
    numpy can read; rounded to float32 like torch.as_tensor(row, dtype=float32)).  Returns the fp32 batch
    [B,3,Hp,Wp] (memory format channels_last on request) holding (x - mean) / std inside each image and 0 in the
    padding: engine.py:107-110 + net_transforms.py:112-121 + :238-247 in one launch.  `out_sizes` ([(Ho, Wo)] per image): each
    image is also resized to that size on the way (bilinear, align_corners=False, scale recomputed from the sizes:
    net_transforms.py:151-175 as ATen computes it on the GPU); an image whose size is already (Ho, Wo) is not interpolated."""
    import ctypes
    import numpy as np
    B = len(images)
    first = images[0]
    _require_cuda(first, "image")
    if first.dtype not in _DT:
        raise TypeError("image dtype %s not supported (float16 / float32)" % first.dtype)
    keep, ptrs, Hs, Ws = [], [], [], []
    for img in images:
        if img.dtype != first.dtype or not img.is_cuda or img.dim() != 3 or img.shape[0] != 3:
            raise ValueError("normalize_pad needs 3 x H x W CUDA images of one dtype")
        img = img if img.is_contiguous() else img.contiguous()
        keep.append(img); ptrs.append(img.data_ptr()); Hs.append(int(img.shape[1])); Ws.append(int(img.shape[2]))
    m = np.ascontiguousarray(np.asarray(means, dtype=np.float64).reshape(B, 3).astype(np.float32))
    sd = np.ascontiguousarray(np.asarray(stds, dtype=np.float64).reshape(B, 3).astype(np.float32))
    fmt = torch.channels_last if channels_last else torch.contiguous_format
    out = torch.empty((B, 3, Hp, Wp), dtype=torch.float32, device=first.device, memory_format=fmt)
    fp = ctypes.POINTER(ctypes.c_float)
    if out_sizes is not None and any((int(oh), int(ow)) != (h, w) for (oh, ow), h, w in zip(out_sizes, Hs, Ws)):
        _lib.check(_lib.lib().dib_normalize_resize_pad(_lib.ptr_array(ptrs), _DT[first.dtype], _lib.int_array(Hs), _lib.int_array(Ws),
                                                       _lib.int_array([int(o[0]) for o in out_sizes]), _lib.int_array([int(o[1]) for o in out_sizes]),
                                                       B, m.ctypes.data_as(fp), sd.ctypes.data_as(fp), out.data_ptr(), Hp, Wp,
                                                       int(bool(channels_last)), _stream(first.device)))
        return out
    _lib.check(_lib.lib().dib_normalize_pad(_lib.ptr_array(ptrs), _DT[first.dtype], _lib.int_array(Hs), _lib.int_array(Ws), B,
                                            m.ctypes.data_as(fp), sd.ctypes.data_as(fp), out.data_ptr(), Hp, Wp,
                                            int(bool(channels_last)), _stream(first.device)))
    return out
