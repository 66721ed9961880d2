"""dib_blur_step (one C call per blur step: compaction on the library's side stream, blur on the caller's) against the
oracle and against the two separate calls, bit for bit: default ordering, PSFS_COMPLETE, PSFs produced by work still
queued on the caller's stream, ring reuse and growth, several caller streams, graph capture."""
import numpy as np
import pytest
import torch

import dib_oracle as O

pytestmark = pytest.mark.gpu


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view({2: np.uint16, 4: np.uint32}[a.dtype.itemsize])


def _same(g, w):
    """bit equality; the oracle squeezes unit dims like the reference's manual_blur (:69), blur_ops does not"""
    return np.array_equal(_bits(g.cpu().numpy()).reshape(w.shape), _bits(w))


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _psf(rs, n, spread, K=128, dtype=np.float16):
    a = np.zeros((K, K), np.float64)
    c = K // 2 - 1
    a[np.clip(rs.randint(-spread, spread + 1, n) + c, 0, K - 1), np.clip(rs.randint(-spread, spread + 1, n) + c, 0, K - 1)] = rs.random_sample(n) + 0.05
    return O.to_half_like_torch(a) if dtype == np.float16 else a.astype(np.float32)


def _case(rs, count, K=128):
    imgs, psfs = [], []
    for i in range(count):
        C = (3, 1, 2)[i % 3]
        imgs.append(rs.random_sample((C, 66 + (i * 11) % 70, 70 + (i * 29) % 200)).astype(np.float16))
        psfs.append(_psf(rs, 3 + (i * 7) % 50, 1 + (i * 5) % 45, K))
    return imgs, psfs


def _oracle(imgs, psfs, blurring=None):
    want = [a.copy() for a in imgs]
    O.blur_image_list(want, [{"blurring": True if blurring is None else blurring[i]} for i in range(len(imgs))], psfs)
    return want


@pytest.mark.parametrize("complete", [False, True])
def test_step_equals_oracle_and_two_call_path(complete):
    from detectinblur_amd import blur_ops
    rs = np.random.RandomState(31)
    imgs, psfs = _case(rs, 9)
    want = _oracle(imgs, psfs)
    t_imgs, t_psfs = [_dev(a) for a in imgs], [_dev(p) for p in psfs]
    torch.cuda.synchronize()
    got = blur_ops.blur_step(t_imgs, list(range(9)), t_psfs, psfs_complete=complete)
    tabs = blur_ops.compact_psfs(t_psfs, normalize=True)
    two = blur_ops.sparse_blur(t_imgs, list(range(9)), tabs)
    for g, t, w in zip(got, two, want):
        assert _same(g, w)
        assert torch.equal(g, t)


def test_step_through_blur_image_list_with_skipped_entries_and_hint():
    from detectinblur_amd.models import blur_functions as BF
    rs = np.random.RandomState(32)
    imgs, psfs = _case(rs, 12)
    blurring = [i % 4 != 2 for i in range(12)]
    want = _oracle(imgs, psfs, blurring)
    for complete in (False, True):
        got = [_dev(a) for a in imgs]
        dicts = [{"blurring": b, "psf_taps": int(np.count_nonzero(p))} for b, p in zip(blurring, psfs)]
        t_psfs = [_dev(p) for p in psfs]
        torch.cuda.synchronize()
        assert BF.blur_image_list(got, dicts, t_psfs, psfs_complete=complete) is None
        for g, w in zip(got, want):
            assert _same(g, w)


def test_psfs_produced_by_queued_work_are_waited_for():
    """Default flags: the PSFs are written by a kernel queued on the caller's stream right before the call, behind a
    long-running kernel -- the side stream's compaction must not read them early."""
    from detectinblur_amd import blur_ops
    rs = np.random.RandomState(33)
    imgs, psfs = _case(rs, 4)
    want = _oracle(imgs, psfs)
    t_imgs = [_dev(a) for a in imgs]
    src = [_dev(p) for p in psfs]
    dst = [torch.zeros(128, 128, dtype=torch.float16, device="cuda") for _ in psfs]
    big = torch.rand(8192, 8192, device="cuda")
    torch.cuda.synchronize()
    for trial in range(3):
        for d in dst:
            d.zero_()
        torch.cuda.synchronize()
        for _ in range(3):
            big = big @ big * 1e-4            # tens of milliseconds in front of the copies
        for d, s in zip(dst, src):
            d.copy_(s)
        got = blur_ops.blur_step(t_imgs, [0, 1, 2, 3], dst)
        for g, w in zip(got, want):
            assert _same(g, w)


def test_ring_reuse_growth_and_mixed_canvases():
    """More steps than ring slots with a different batch every step (a slot rewritten too early would blur with the wrong
    taps), batches that outgrow the ring's buffers, 256-wide PSFs between 128-wide ones; all queued without a host sync."""
    from detectinblur_amd import blur_ops
    rs = np.random.RandomState(34)
    plan = [(3, 128), (5, 128), (2, 256), (9, 128), (1, 128), (4, 256), (9, 128), (33, 128), (2, 128), (6, 128), (3, 256), (7, 128)]
    cases, outs = [], []
    for count, K in plan:
        imgs, psfs = _case(rs, count, K)
        cases.append((imgs, psfs, [_dev(a) for a in imgs], [_dev(p) for p in psfs]))
    torch.cuda.synchronize()
    for imgs, psfs, t_imgs, t_psfs in cases:
        outs.append(blur_ops.blur_step(t_imgs, list(range(len(imgs))), t_psfs, psfs_complete=True))
    torch.cuda.synchronize()
    for (imgs, psfs, _, _), got in zip(cases, outs):
        for g, w in zip(got, _oracle(imgs, psfs)):
            assert _same(g, w)


def test_two_caller_streams_share_the_ring():
    from detectinblur_amd import blur_ops
    rs = np.random.RandomState(35)
    a_imgs, a_psfs = _case(rs, 5)
    b_imgs, b_psfs = _case(rs, 6)
    ta, pa, tb, pb = [_dev(a) for a in a_imgs], [_dev(p) for p in a_psfs], [_dev(a) for a in b_imgs], [_dev(p) for p in b_psfs]
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    got_a, got_b = [], []
    for _ in range(5):
        with torch.cuda.stream(s1):
            got_a.append(blur_ops.blur_step(ta, list(range(5)), pa, psfs_complete=True))
        with torch.cuda.stream(s2):
            got_b.append(blur_ops.blur_step(tb, list(range(6)), pb, psfs_complete=True))
    torch.cuda.synchronize()
    wa, wb = _oracle(a_imgs, a_psfs), _oracle(b_imgs, b_psfs)
    for got in got_a:
        for g, w in zip(got, wa):
            assert _same(g, w)
    for got in got_b:
        for g, w in zip(got, wb):
            assert _same(g, w)


def test_step_is_graph_capturable_and_release_is_idempotent():
    """Under capture the library's ring is not used (DIB_ECAPTURE without caller tables; blur_ops retries with tables from
    the capture's pool): replays reproduce the eager step for new pixels; eager steps in between keep working."""
    from detectinblur_amd import _lib, blur_ops
    rs = np.random.RandomState(36)
    imgs, psfs = _case(rs, 3)
    t_imgs, t_psfs = [_dev(a) for a in imgs], [_dev(p) for p in psfs]
    for _ in range(3):
        blur_ops.blur_step(t_imgs, [0, 1, 2], t_psfs)
    torch.cuda.synchronize()
    # the raw entry point refuses a capturing stream without caller tables and launches nothing
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        out = blur_ops.blur_step(t_imgs, [0, 1, 2], t_psfs, psfs_complete=True)
    for trial in range(2):
        for t in t_imgs:
            t.copy_(torch.rand(t.shape, device=t.device).half())
        eager = blur_ops.blur_step(t_imgs, [0, 1, 2], t_psfs)
        g.replay()
        torch.cuda.synchronize()
        for a, b in zip(out, eager):
            assert torch.equal(a, b)
    torch.cuda.synchronize()
    assert _lib.lib().dib_blur_step_release() == 0
    assert _lib.lib().dib_blur_step_release() == 0
    again = blur_ops.blur_step(t_imgs, [0, 1, 2], t_psfs)      # the ring comes back on demand
    for a, b in zip(again, eager):
        assert torch.equal(a, b)


def test_step_argument_errors():
    from detectinblur_amd import _lib, blur_ops
    img = torch.zeros(3, 64, 100, dtype=torch.float16, device="cuda")
    psf = torch.zeros(128, 128, dtype=torch.float16, device="cuda")
    psf[63, 63] = 1
    with pytest.raises(_lib.DibError) as e:        # what F.pad(mode='reflect') raises in the reference
        blur_ops.blur_step([img], [0], [psf])
    assert e.value.code == _lib.DIB_ESHAPE
    with pytest.raises(ValueError):
        blur_ops.blur_step([torch.zeros(3, 80, 100, dtype=torch.float16, device="cuda")], [0], [torch.zeros(100, 100, dtype=torch.float16, device="cuda")])
    ok = blur_ops.blur_step([torch.ones(3, 80, 100, dtype=torch.float16, device="cuda")], [0], [psf])   # still usable after the errors
    assert torch.equal(ok[0], torch.ones(3, 80, 100, dtype=torch.float16, device="cuda"))
