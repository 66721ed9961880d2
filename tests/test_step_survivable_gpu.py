"""No kernel of the library traps (include/dib.h, "Device status"): a hand-off of the blur step's single launch that never
arrives, and a tap table of the wrong window geometry, leave a code in the device's status word instead of killing the GPU
context -- the process survives, the next call reports it, the step goes on as two launches (reference call site:
engine.py:101, `blur_image_list` inside a DDP rank's training loop).  Plus: the single launch beside a competing stream, the
caller-workspace form of the step and the bound on the library's private table buffers."""
import ctypes
import warnings

import numpy as np
import pytest
import torch

import dib_oracle as O

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _hooks():
    from detectinblur_amd import _lib
    l = _lib.lib()
    l.dib_debug_set_step_fused.argtypes = [ctypes.c_int]; l.dib_debug_set_step_fused.restype = None
    l.dib_debug_set_step_poll_budget.argtypes = [ctypes.c_uint]; l.dib_debug_set_step_poll_budget.restype = None
    l.dib_debug_set_step_nosignal.argtypes = [ctypes.c_int]; l.dib_debug_set_step_nosignal.restype = None
    l.dib_debug_step_single_launch.argtypes = [ctypes.c_int]; l.dib_debug_step_single_launch.restype = ctypes.c_int
    return l


def _psf(rs, n, spread, K=128):
    a = np.zeros((K, K), np.float64)
    c = K // 2 - 1
    a[np.clip(rs.randint(-spread, spread + 1, n) + c, 0, K - 1), np.clip(rs.randint(-spread, spread + 1, n) + c, 0, K - 1)] = rs.random_sample(n) + 0.05
    return O.to_half_like_torch(a)


def _batch(rs, count, H=200, W=300):
    imgs = [rs.random_sample((3, H + 8 * i, W + 16 * i)).astype(np.float16) for i in range(count)]
    psfs = [_psf(rs, 5 + 9 * i, 2 + 3 * i) for i in range(count)]
    return imgs, psfs


def _same(a, b):
    return all(torch.equal(x.view(torch.int16), y.view(torch.int16)) for x, y in zip(a, b))


def test_a_hand_off_that_never_arrives_is_reported_and_survived():
    from detectinblur_amd import _lib, blur_ops
    l = _hooks()
    rs = np.random.RandomState(3)
    imgs, psfs = _batch(rs, 4)
    t_imgs, t_psfs = [_dev(a) for a in imgs], [_dev(p) for p in psfs]
    idx = list(range(4))
    l.dib_debug_set_step_fused(0)
    want = blur_ops.blur_step(t_imgs, idx, t_psfs)
    l.dib_debug_set_step_fused(1)
    assert l.dib_debug_step_single_launch(-1) == 1
    torch.cuda.synchronize()
    assert l.dib_device_status(0) == 0
    try:
        # a single launch whose compacting workgroups stay silent, with a budget of ~2,000 polls: every blur workgroup gives up
        l.dib_debug_set_step_nosignal(1)
        l.dib_debug_set_step_poll_budget(2000)
        broken = blur_ops.blur_step(t_imgs, idx, t_psfs)
        torch.cuda.synchronize()          # the context is alive: no trap, no hipErrorLaunchFailure
        del broken
        assert l.dib_device_status(0) == _lib.DIB_ETIMEOUT
        l.dib_debug_set_step_nosignal(0)
        l.dib_debug_set_step_poll_budget(0)
        # the next step reports it (a warning carrying the library's text), launches the batch again -- as two launches -- and is right
        with warnings.catch_warnings(record=True) as seen:
            warnings.simplefilter("always")
            got = blur_ops.blur_step(t_imgs, idx, t_psfs)
        torch.cuda.synchronize()
        assert any("poll budget" in str(w.message) and issubclass(w.category, RuntimeWarning) for w in seen), [str(w.message) for w in seen]
        assert _same(got, want)
        assert l.dib_device_status(0) == 0
        assert l.dib_debug_step_single_launch(-1) == 0          # out of service on this device
        assert _same(blur_ops.blur_step(t_imgs, idx, t_psfs), want)
        # the raw entry point returns DIB_ETIMEOUT exactly once and launches nothing
        l.dib_debug_step_single_launch(1)
        l.dib_debug_set_step_nosignal(1)
        l.dib_debug_set_step_poll_budget(2000)
        blur_ops.blur_step(t_imgs, idx, t_psfs)
        torch.cuda.synchronize()
        l.dib_debug_set_step_nosignal(0)
        l.dib_debug_set_step_poll_budget(0)
        tabs = blur_ops.compact_psfs(t_psfs, normalize=True)
        with pytest.raises(_lib.DibStepTimeout):
            _lib.check(l.dib_sparse_blur(_lib.ptr_array([t.data_ptr() for t in t_imgs]), _lib.ptr_array([torch.empty_like(t).data_ptr() for t in t_imgs]),
                                         _lib.int_array([3] * 4), _lib.int_array([t.shape[1] for t in t_imgs]), _lib.int_array([t.shape[2] for t in t_imgs]),
                                         _lib.int_array(idx), 4, _lib.DIB_F16, tabs.buf.data_ptr(), 4, 128, 0, blur_ops._stream()))
        assert l.dib_device_status(0) == 0
        assert _same(blur_ops.sparse_blur(t_imgs, idx, tabs), want)
    finally:
        l.dib_debug_set_step_nosignal(0)
        l.dib_debug_set_step_poll_budget(0)
        l.dib_debug_step_single_launch(1)
        l.dib_device_status(1)
        torch.cuda.synchronize()
    assert _same(blur_ops.blur_step(t_imgs, idx, t_psfs), want)       # back in service: the single launch, same bits


def test_a_table_of_the_other_window_geometry_is_reported_not_trapped():
    from detectinblur_amd import _lib, blur_ops
    l = _hooks()
    rs = np.random.RandomState(5)
    imgs, psfs = _batch(rs, 2)
    t_imgs, t_psfs = [_dev(a) for a in imgs], [_dev(p) for p in psfs]
    std = blur_ops.compact_psfs(t_psfs, normalize=True)
    want = blur_ops.sparse_blur(t_imgs, [0, 1], std)
    torch.cuda.synchronize()
    assert l.dib_device_status(1) == 0
    outs = [torch.empty_like(t) for t in t_imgs]

    def raw(tables, mode):
        return l.dib_sparse_blur(_lib.ptr_array([t.data_ptr() for t in t_imgs]), _lib.ptr_array([o.data_ptr() for o in outs]), _lib.int_array([3, 3]),
                                 _lib.int_array([t.shape[1] for t in t_imgs]), _lib.int_array([t.shape[2] for t in t_imgs]), _lib.int_array([0, 1]), 2,
                                 _lib.DIB_F16, tables.buf.data_ptr(), 2, 128, mode, blur_ops._stream())
    try:
        assert raw(std, _lib.DIB_WINDOW_LARGE) == 0             # standard tables, the large window's kernel: the launch itself is accepted
        torch.cuda.synchronize()                                 # ... and the context lives
        assert l.dib_device_status(0) == _lib.DIB_EINVAL
        with pytest.raises(_lib.DibError, match="standard LDS window"):
            _lib.check(raw(std, 0))
        assert raw(std, 0) == 0
        torch.cuda.synchronize()
        assert _same(outs, want)
        large = blur_ops.compact_psfs(t_psfs, normalize=True, large_window=True)
        assert raw(large, 0) == 0                                # and the other way round
        torch.cuda.synchronize()
        with pytest.raises(_lib.DibError, match="large LDS window"):
            _lib.check(raw(large, _lib.DIB_WINDOW_LARGE))
        assert _same(blur_ops.sparse_blur(t_imgs, [0, 1], large), want)      # the Python layer passes the geometry its tables carry
    finally:
        l.dib_device_status(1)


def test_single_launch_beside_a_competing_stream_equals_the_two_launch_path():
    """~2,000 steps of the single launch while a second stream keeps the chip busy with GEMMs and a third with convolutions (the
    reference's call site shares the GPU with a detector's kernels and RCCL's): every sampled output bit-identical to the
    two-launch path, no status raised."""
    from detectinblur_amd import blur_ops
    l = _hooks()
    rs = np.random.RandomState(11)
    sets = []
    for k in range(6):
        imgs, psfs = _batch(rs, 4, 180 + 20 * k, 260 + 30 * k)
        sets.append(([_dev(a) for a in imgs], [_dev(p) for p in psfs]))
    idx = list(range(4))
    l.dib_debug_set_step_fused(0)
    want = [blur_ops.blur_step(i, idx, p) for i, p in sets]
    l.dib_debug_set_step_fused(1)
    torch.cuda.synchronize()
    side, side2 = torch.cuda.Stream(), torch.cuda.Stream()
    a = torch.randn(4096, 4096, device="cuda", dtype=torch.float16)
    x = torch.randn(8, 64, 200, 200, device="cuda")
    w = torch.randn(64, 64, 3, 3, device="cuda")
    bad = 0
    for step in range(2000):
        if step % 4 == 0:
            with torch.cuda.stream(side):
                a @ a
            with torch.cuda.stream(side2):
                torch.nn.functional.conv2d(x, w, padding=1)
        k = step % len(sets)
        got = blur_ops.blur_step(sets[k][0], idx, sets[k][1])
        if step % 50 == 0:
            bad += not _same(got, want[k])
    torch.cuda.synchronize()
    assert bad == 0
    assert l.dib_device_status(0) == 0
    assert l.dib_debug_step_single_launch(-1) == 1


def test_caller_workspace_step_equals_library_buffer_step():
    from detectinblur_amd import _lib, blur_ops
    l = _hooks()
    rs = np.random.RandomState(17)
    ws_bytes = l.dib_blur_step_workspace_bytes(128, 6)
    assert ws_bytes >= l.dib_tap_tables_bytes(128, 6)
    ws = torch.empty(ws_bytes + 256, dtype=torch.uint8, device="cuda")
    base = (ws.data_ptr() + 255) & ~255
    state = ctypes.c_ulonglong(0)
    for rep in range(12):
        count = 1 + rep % 6
        imgs, psfs = _batch(rs, count, 100 + 30 * rep, 150 + 20 * rep)
        t_imgs, t_psfs = [_dev(a) for a in imgs], [_dev(p) for p in psfs]
        want = blur_ops.blur_step(t_imgs, list(range(count)), t_psfs)
        outs = [torch.empty_like(t) for t in t_imgs]
        rc = l.dib_blur_step_ws(_lib.ptr_array([p.data_ptr() for p in t_psfs]), _lib.DIB_F16, count, 128, 1,
                                _lib.ptr_array([t.data_ptr() for t in t_imgs]), _lib.ptr_array([o.data_ptr() for o in outs]), _lib.int_array([3] * count),
                                _lib.int_array([t.shape[1] for t in t_imgs]), _lib.int_array([t.shape[2] for t in t_imgs]), _lib.int_array(list(range(count))),
                                count, _lib.DIB_F16, 0, base, ws_bytes, ctypes.byref(state), 0, blur_ops._stream())
        _lib.check(rc)
        assert _same(outs, want), rep
    assert state.value >> 32 == 1 and (state.value & 0xffffffff) == sum(1 + r % 6 for r in range(12))
    # too small a workspace is refused
    assert l.dib_blur_step_ws(_lib.ptr_array([t_psfs[0].data_ptr()]), _lib.DIB_F16, 1, 128, 1, _lib.ptr_array([t_imgs[0].data_ptr()]),
                              _lib.ptr_array([outs[0].data_ptr()]), _lib.int_array([3]), _lib.int_array([t_imgs[0].shape[1]]), _lib.int_array([t_imgs[0].shape[2]]),
                              _lib.int_array([0]), 1, _lib.DIB_F16, 0, base, 1024, ctypes.byref(state), 0, blur_ops._stream()) == _lib.DIB_EINVAL


def test_a_thousand_short_lived_streams_do_not_grow_the_librarys_buffers():
    from detectinblur_amd import blur_ops
    rs = np.random.RandomState(23)
    imgs, psfs = _batch(rs, 2, 90, 130)
    t_imgs, t_psfs = [_dev(a) for a in imgs], [_dev(p) for p in psfs]
    want = blur_ops.blur_step(t_imgs, [0, 1], t_psfs)
    torch.cuda.synchronize()
    free0 = None
    for k in range(1000):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            got = blur_ops.blur_step(t_imgs, [0, 1], t_psfs)
        if k % 100 == 0:
            s.synchronize()
            assert _same(got, want), k
        if k == 100:
            torch.cuda.synchronize()
            free0 = torch.cuda.mem_get_info()[0]
        del s
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    # 900 further streams: the library holds at most 8 x 2 buffers per device, so nothing (beyond allocator noise) went missing
    assert free0 - free1 < 64 << 20, (free0, free1)
