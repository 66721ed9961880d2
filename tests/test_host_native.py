"""CPU tests of the native host library (libdib_host.so) and the Trajectory / PSF classes against
the golden vectors from the real reference and against the oracle.  Bit-exact."""
import numpy as np
import pytest

import dib_oracle as O
import golden_inputs as GI


@pytest.mark.parametrize("param", GI.PARAMS)
@pytest.mark.parametrize("seed", GI.TRAJ_SEEDS)
def test_trajectory_class_golden(golden, param, seed):
    from detectinblur_amd.motion_blur.generate_trajectory import Trajectory
    np.random.seed(seed)
    tr = Trajectory(canvas=256, max_len=96, expl=param).fit()
    key = "traj_p%g_s%d" % (param, seed)
    assert np.array_equal(tr.x, golden.traj[key + "_fit1"])
    tr = tr.fit()
    assert np.array_equal(tr.x, golden.traj[key + "_fit2"])
    assert [tr.tot_length, tr.big_expl_count] == list(golden.traj[key + "_len"])
    # the numpy global stream was advanced exactly as the reference advances it
    assert [np.random.uniform(), np.random.randn()] == list(golden.traj[key + "_next"])


def test_trajectory_expl_none_and_big(golden):
    from detectinblur_amd.motion_blur.generate_trajectory import Trajectory
    np.random.seed(7)
    tr = Trajectory(canvas=64, iters=500, max_len=60).fit()
    assert np.array_equal(tr.x, golden.traj["traj_none_s7"])
    assert [tr.expl, tr.tot_length, tr.big_expl_count] == list(golden.traj["traj_none_s7_expl"])
    np.random.seed(11)
    tr = Trajectory(canvas=256, iters=2000, max_len=96, expl=0.9).fit()
    assert np.array_equal(tr.x, golden.traj["traj_big_s11"])
    assert [tr.tot_length, tr.big_expl_count] == list(golden.traj["traj_big_s11_len"])
    assert np.array_equal(tr.unprocessedX + complex(128, 128), tr.x)


@pytest.mark.parametrize("param", GI.PARAMS)
@pytest.mark.parametrize("fi", range(len(GI.FRACTIONS)))
def test_psf_class_golden(golden, param, fi):
    from detectinblur_amd.motion_blur.generate_PSF import PSF
    from detectinblur_amd.motion_blur.generate_trajectory import Trajectory
    np.random.seed(GI.psf_seed(param, fi))
    tr = Trajectory(canvas=256, max_len=96, expl=param).fit()
    tr = tr.fit()
    p = PSF(canvas=256, trajectory=tr, fraction=[GI.FRACTIONS[fi]])
    raw = p.fit()[0].copy()
    assert np.array_equal(raw, GI.golden_psf(param, fi, "raw"))
    p.centerPSF()
    assert np.array_equal(p.PSFs[0], GI.golden_psf(param, fi, "cen"))


def test_psf_multi_fraction_golden(golden):
    from detectinblur_amd.motion_blur.generate_PSF import PSF

    class T:
        x = golden.psf["psf_multi_traj"]
    out = PSF(canvas=128, trajectory=T, fraction=[1 / 100, 1 / 10, 1 / 2, 1]).fit()
    assert len(out) == 4
    for i, a in enumerate(out):
        assert np.array_equal(a, golden.psf["psf_multi_%d" % i])


def test_psf_out_of_canvas_raises():
    from detectinblur_amd.motion_blur.generate_PSF import PSF

    class T:
        x = np.array([10 + 10j, 63.5 + 10j], dtype=np.complex128)
    with pytest.raises(IndexError):
        PSF(canvas=64, trajectory=T, fraction=[1]).fit()


def test_native_rng_matches_numpy_stream():
    from detectinblur_amd import _hostlib
    np.random.seed(123)
    want = [np.random.uniform(), np.random.randn(), np.random.randn(), np.random.uniform(), np.random.randn()]
    np.random.seed(123)
    l = _hostlib.lib()
    with _hostlib.NumpyGlobalStream() as r:
        got = [l.dib_rng_uniform(r), l.dib_rng_gauss(r), l.dib_rng_gauss(r), l.dib_rng_uniform(r), l.dib_rng_gauss(r)]
    assert got == want
    # and the state written back continues the same stream (625 draws forces a twist)
    a = [np.random.uniform() for _ in range(700)]
    np.random.seed(123)
    [np.random.uniform(), np.random.randn(), np.random.randn(), np.random.uniform(), np.random.randn()]
    assert a == [np.random.uniform() for _ in range(700)]


def test_center_vs_oracle_random():
    from detectinblur_amd import _hostlib
    rs = np.random.RandomState(4)
    for canvas in (64, 100, 256):
        a = np.zeros((canvas, canvas))
        n = 200
        a[rs.randint(2, canvas - 2, n), rs.randint(2, canvas - 2, n)] = rs.random_sample(n)
        want = O.psf_center(a)
        got = a.copy()
        off = (np.ctypeslib.ctypes.c_int * 2)()
        assert _hostlib.lib().dib_psf_center(_hostlib.dptr(got), canvas, off) == 0
        assert np.array_equal(got, want)
        assert (off[0], off[1]) == O.psf_center_offsets(a)
