"""`--cpu_blur`: the reference's Fourier-domain PSF blur, kept as the CPU comparison path
(reference motion_blur/blur_image.py:23-154, `BlurImageHandler`).

This is host code by definition (BASELINE.md section 4 times it on the host cores next to the GPU
numbers); it is not on the accelerated path and is numerically NOT comparable with `--gpu_blur`
(it min-max stretches the image twice).  cv2 is not a dependency: `cv2.normalize(NORM_MINMAX)` is
restated with numpy, the final `cv2.resize(..., INTER_LANCZOS4)` of the small-image branch with
PIL's Lanczos filter (a documented, approximate substitute).
"""
import math

import numpy as np
from PIL import Image
from scipy import signal


def _minmax01(a):
    """cv2.normalize(src, dst, 0, 1, NORM_MINMAX, CV_32F): global min/max over all channels."""
    a = np.asarray(a, dtype=np.float64)
    lo, hi = float(a.min()), float(a.max())
    scale = 1.0 / (hi - lo) if hi > lo else 0.0
    return ((a - lo) * scale).astype(np.float32)


class BlurImageHandler(object):
    def __init__(self, image_path, PSFs=None, pillowImage=None, part=None, path__to_save=None, buffPadImage=True):
        """image_path / pillowImage: RGB (or grey) image; PSFs: list of k x k kernels; part: which one."""
        if PSFs is None:
            raise ValueError("PSFs must be given (the reference's default branch needs an image shape it never sets)")
        self.PSFs = PSFs
        self.path_to_save = path__to_save
        if pillowImage is None:
            self.original = Image.open(image_path)
        else:
            self.original = pillowImage
            self.originalPillowImage = self.original

        # images smaller than the kernel are upscaled first (reference :55-69)
        self.originalSize = self.original.size
        yN, xN = self.original.size
        key, kex = self.PSFs[0].shape
        if yN - key < 0 or xN - kex < 0:
            ratio = max(key / yN, kex / xN)
            self.original = self.original.resize((math.ceil(ratio * yN), math.ceil(ratio * xN)), Image.BICUBIC)
        else:
            self.originalSize = None
        self.original = np.array(self.original)

        self.buffPadImage = buffPadImage
        if buffPadImage:                                                 # reference :78-85
            pr, pc = round(self.PSFs[0].shape[0] / 2), round(self.PSFs[0].shape[1] / 2)
            pad = ((pr, pr), (pc, pc), (0, 0)) if self.original.ndim > 2 else ((pr, pr), (pc, pc))
            self.original = np.pad(self.original, pad_width=pad, mode="edge")
        if self.original.ndim < 3:                                       # grey -> RGB (reference :91-97)
            self.original = np.repeat(self.original[:, :, None], 3, axis=2).astype(np.float64)
        self.shape = self.original.shape
        self.part = part
        self.result = []
        self.pilImageResult = None

    def blur_image(self, save=False, show=False, oldDeltaPad=False):
        psf = self.PSFs[0] if self.part is None else self.PSFs[self.part]
        yN, xN, _ = self.shape
        key, kex = self.PSFs[0].shape
        dY, dX = yN - key, xN - kex
        if oldDeltaPad:
            tmp = np.pad(psf, dX // 2, "constant")
        else:                                                            # reference :117-123
            tmp = np.pad(psf, ((dY // 2, math.ceil(dY / 2)), (math.ceil(dX / 2), dX // 2)), "constant")
        tmp = _minmax01(tmp)                                             # :128
        blurred = _minmax01(self.original)                               # :129-130
        for ch in range(3):                                              # :131-133
            blurred[:, :, ch] = signal.fftconvolve(blurred[:, :, ch], tmp, "same")
        blurred = _minmax01(blurred)                                     # :134
        if self.buffPadImage:                                            # :137-140
            pr, pc = round(self.PSFs[0].shape[0] / 2), round(self.PSFs[0].shape[1] / 2)
            blurred = blurred[pr:blurred.shape[0] - pr, pc:blurred.shape[1] - pc, :]
        if self.originalSize is not None:                                # :142-143
            img = Image.fromarray((np.clip(blurred, 0, 1) * 255).astype(np.uint8)).resize(self.originalSize, Image.LANCZOS)
            blurred = np.asarray(img).astype(np.float32) / 255
        self.result = [np.abs(blurred)]
        self.pilImageResult = Image.fromarray((blurred * 255).astype(np.uint8))   # :147
        return True
