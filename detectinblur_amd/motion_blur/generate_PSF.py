"""Trajectory -> PSF rasteriser -- drop-in for the reference's motion_blur/generate_PSF.py
(`PSF`, :9-148).

`fit()` / `centerPSF()` keep the reference's host-side contract (lists of float64 numpy arrays in
`self.PSFs`), computed in native code (libdib_host.so) instead of a 2000-iteration Python loop:
this is what DataLoader workers call.  The batched device path (HIP, bit-identical float64) is
`detectinblur_amd.blur_ops.rasterize_psfs`, used by the engine when PSFs are generated on the GPU
and by the offline PSF-store builder.
"""
import numpy as np

from .. import _hostlib
from .generate_trajectory import Trajectory


class PSF(object):
    def __init__(self, canvas=None, trajectory=None, fraction=None, path_to_save=None):
        self.canvas = (canvas, canvas)
        if trajectory is None:
            # the reference's default branch calls Trajectory.fit(show=, save=) which does not exist
            # (generate_PSF.py:16-18); here it simply generates one
            self.trajectory_obj = Trajectory(canvas=canvas, expl=0.005).fit()
            self.trajectory = self.trajectory_obj.x
        else:
            self.trajectory = trajectory.x
        if fraction is None:
            self.fraction = [1 / 100, 1 / 10, 1 / 2, 1]
        else:
            self.fraction = fraction
        self.path_to_save = path_to_save
        self.PSFnumber = len(self.fraction)
        self.iters = len(self.trajectory)
        self.PSFs = []

    def fit(self, show=False, save=False):
        """Appends one canvas x canvas float64 PSF per exposure fraction (cumulative windows) to
        self.PSFs and returns the list (reference :31-83)."""
        canvas = int(self.canvas[0])
        traj = np.ascontiguousarray(self.trajectory, dtype=np.complex128).view(np.float64).reshape(-1, 2)
        fr = np.ascontiguousarray([float(f) for f in self.fraction], dtype=np.float64)
        out = np.empty((len(fr), canvas, canvas), dtype=np.float64)
        rc = _hostlib.lib().dib_psf_fit(_hostlib.dptr(traj), int(self.iters), _hostlib.dptr(fr), len(fr), canvas,
                                        _hostlib.dptr(out))
        if rc == -2:
            raise IndexError("trajectory leaves the %d x %d canvas" % (canvas, canvas))
        if rc != 0:
            raise ValueError("PSF.fit: bad arguments")
        for j in range(len(fr)):
            self.PSFs.append(out[j])
        if show or save:
            self.plot_canvas(show, save)
        return self.PSFs

    def plot_canvas(self, show, save):
        if len(self.PSFs) == 0:
            raise Exception("Please run fit() method first.")
        import matplotlib.pyplot as plt
        if show:
            plt.close()
            fig, axes = plt.subplots(1, self.PSFnumber, figsize=(10, 10))
            axes = np.atleast_1d(axes)
            for i in range(self.PSFnumber):
                axes[i].imshow(self.PSFs[i], cmap="gray")
        if show and save:
            if self.path_to_save is None:
                raise Exception("Please create Trajectory instance with path_to_save")
            plt.savefig(self.path_to_save)
            plt.show()
        elif save:
            from PIL import Image
            img = np.uint8(255 * (self.PSFs[0] / np.max(self.PSFs[0])))
            Image.fromarray(img).save(self.path_to_save or "psf.png")
        elif show:
            plt.show()

    def centerPSF(self):
        """Rolls PSFs[0] so that its weighted centroid sits at the canvas centre (reference :106-123)."""
        psf = np.ascontiguousarray(self.PSFs[0], dtype=np.float64).copy()
        rc = _hostlib.lib().dib_psf_center(_hostlib.dptr(psf), int(psf.shape[0]), None)
        if rc != 0:
            raise ValueError("PSF.centerPSF failed")
        self.PSFs[0] = psf

    def findOffsets(self):
        """[left, top, right, bottom] extent of the PSF support around the centre (reference :125-148)."""
        psf = self.PSFs[0]
        ys, xs = np.nonzero(psf > 0)
        left = right = bottom = top = 0
        for cx, cy in zip(xs, ys):
            ox = cx - (self.canvas[0] / 2 - 1)
            if ox > 0 and ox > right:
                right = ox
            elif ox <= 0 and -ox > left:
                left = -ox
            oy = cy - (self.canvas[1] / 2 - 1)
            if oy > 0 and oy > bottom:
                bottom = oy
            elif oy <= 0 and -oy > top:
                top = -oy
        return [left, top, right, bottom]
