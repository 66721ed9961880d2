"""Random camera-shake trajectories [Boracchi & Foi 2012] -- drop-in for the reference's
motion_blur/generate_trajectory.py (`Trajectory`, :8-104).

Same constructor, attributes (`x`, `tot_length`, `big_expl_count`, `unprocessedX`) and RNG
consumption (numpy's legacy global stream, draw for draw), so `np.random.seed(s)` gives the
reference's trajectory bit for bit.  The 1999-step walk itself runs in native code
(libdib_host.so, include/dib_host.h) instead of a Python loop: ~30 ms -> ~40 us per fit.
"""
import numpy as np

from .. import _hostlib


class Trajectory(object):
    def __init__(self, canvas=64, iters=2000, max_len=60, expl=None):
        """canvas: size of the domain; iters: samples; max_len: path length in pixels;
        expl: big-shake / perturbation scale (the reference recommends 0.005); None draws
        0.1 * U(0,1) from the global stream (reference :28-31)."""
        self.canvas = canvas
        self.iters = iters
        self.max_len = max_len
        if expl is None:
            self.expl = 0.1 * np.random.uniform(0, 1)
        else:
            self.expl = expl
        self.tot_length = None
        self.big_expl_count = None
        self.x = None
        self.unprocessedX = None

    def fit(self):
        """One random walk; returns self (reference :38-98)."""
        x = np.empty((self.iters, 2), dtype=np.float64)
        raw = np.empty((self.iters, 2), dtype=np.float64)
        stats = np.zeros(2, dtype=np.float64)
        with _hostlib.NumpyGlobalStream() as rng:
            rc = _hostlib.lib().dib_trajectory_fit(rng, int(self.canvas), int(self.iters), float(self.max_len),
                                                   float(self.expl), _hostlib.dptr(x), _hostlib.dptr(raw),
                                                   _hostlib.dptr(stats))
        if rc != 0:
            raise ValueError("Trajectory.fit: bad arguments (iters must be >= 2)")
        self.unprocessedX = raw.view(np.complex128).reshape(self.iters)
        self.x = x.view(np.complex128).reshape(self.iters)
        self.tot_length = float(stats[0])
        self.big_expl_count = int(stats[1])
        return self

    def applyscale_factor(self):
        """Rescale the raw walk to fill the canvas (reference :100-104; unused by the hot path)."""
        x = self.unprocessedX
        half = (self.canvas / 2) - 2
        scaling = np.max([np.max(-1 * x.real / half), np.max(-1 * x.imag / half), np.max(x.real / half),
                          np.max(x.imag / half)])
        self.x = x / scaling
        self.x = self.x + complex(self.canvas / 2, self.canvas / 2)
