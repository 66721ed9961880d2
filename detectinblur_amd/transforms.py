"""Dataset-side transforms -- drop-in for the hot-path parts of the reference's transforms.py.

Every transform has the reference's 3-argument signature `(image, target, blur_dict) ->
(image, target, blur_dict)` (reference transforms.py:35-66, 173-176, 223).  `BlurImage` keeps the
reference's constructor, its Python-`random` / `np.random` draw order and
the keys it writes into `blur_dict`; what changes is where the time goes:

  * trajectory + PSF generation run in native code (libdib_host.so): ~0.15 s -> ~0.3 ms per image
    inside a DataLoader worker, bit-identical float64 results;
  * `--gpu_blur` (blur_image_in_transform=False) ships only the 128 x 128 PSF, as the reference
    does; the blur itself happens on the GPU in models/blur_functions.py;
  * `--cpu_blur` keeps the reference's FFT path (motion_blur/blur_image.py) for comparison runs.

Extra, optional keys (ignored by reference-shaped consumers): `blur_dict["psf_extent"]` =
(rmin, rmax, cmin, cmax) of the PSF support, `blur_dict["psf_taps"]` = its non-zero count and `blur_dict["psf_segments"]` =
the number of LDS window refills it costs the blur on the (standard, large) window geometry: free to compute here and used
as scheduling hints by the GPU blur (heaviest image first; the large window for small launches of refill-heavy PSFs).
"""
import copy
import math
import os
import random

import numpy as np
import torch

from .motion_blur.generate_PSF import PSF
from .motion_blur.generate_trajectory import Trajectory

PARAMS = [0.005, 0.001, 0.00005]          # blur types P1..P3          (reference transforms.py:249)
FRACTIONS = [1 / 18, 1 / 10, 1 / 5, 1 / 2, 1]  # exposure fractions E0..E4 (reference transforms.py:250)
STORED_PSF_COUNT = 12000                  # files per P?E? directory     (reference transforms.py:298)


class Compose(object):
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, image, target, blur_dict={}, epoch_number=None, dryRun=False):
        blur_dict = copy.deepcopy(blur_dict)
        blur_dict["epoch_number"] = epoch_number
        blur_dict["dryRun"] = dryRun
        for t in self.transforms:
            image, target, blur_dict = t(image, target, blur_dict)
        return image, target, blur_dict


class RandomHorizontalFlip(object):
    """Flips image, boxes (x' = W - x), masks and keypoints with probability `prob`
    (reference transforms.py:49-66)."""

    def __init__(self, prob):
        self.prob = prob

    def __call__(self, image, target, blur_dict={}):
        if random.random() < self.prob:
            height, width = image.shape[-2:]
            image = image.flip(-1)
            bbox = target["boxes"]
            bbox[:, [0, 2]] = width - bbox[:, [2, 0]]
            target["boxes"] = bbox
            if "masks" in target:
                target["masks"] = target["masks"].flip(-1)
            if "keypoints" in target:
                target["keypoints"] = _flip_coco_person_keypoints(target["keypoints"], width)
        return image, target, blur_dict


def _flip_coco_person_keypoints(kps, width):
    flip_inds = [0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15]
    flipped = kps[:, flip_inds]
    flipped[..., 0] = width - flipped[..., 0]
    inds = flipped[..., 2] == 0     # COCO convention: invisible keypoints sit at the origin
    flipped[inds] = 0
    return flipped


def to_tensor(pic):
    """PIL image / HxWxC uint8 ndarray -> float32 CxHxW in [0, 1] (torchvision's F.to_tensor, which
    the reference calls at transforms.py:175; torchvision is not a dependency here)."""
    if isinstance(pic, torch.Tensor):
        return pic
    arr = np.asarray(pic)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    t = torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))
    if t.dtype == torch.uint8:
        return t.to(torch.float32).div(255)
    return t.to(torch.float32)


class ToTensor(object):
    def __call__(self, image, target, blur_dict={}):
        return to_tensor(image), target, blur_dict


def sigmoid(x):
    return 1 / (1 + math.exp(-x))


def psf_axis_stats(psf):
    """Principal axes of the PSF support (coordinates of psf > 0, unweighted): returns
    (theta_rad, scale_factor_lambda1, scale_factor_lambda2, (rmin, rmax, cmin, cmax)).
    Reference transforms.py:366-385."""
    ys, xs = np.nonzero(psf > 0)
    yp = ys - ys.mean()
    xp = xs - xs.mean()
    cov = (yp * xp).mean()
    var_x = (xp * xp).mean()
    var_y = (yp * yp).mean()
    root = math.sqrt(math.pow((var_x - var_y) / 2, 2) + math.pow(cov, 2))
    lambda1 = (var_x + var_y) / 2 + root
    lambda2 = (var_x + var_y) / 2 - root
    s1 = 1 - (sigmoid(math.sqrt(lambda1) / 10) - 0.5) * 0.6
    s2 = 1 - (sigmoid(math.sqrt(lambda2) / 10) - 0.5) * 0.6
    theta = -math.atan2(lambda1 - var_x, -cov)
    return theta, s1, s2, (int(ys.min()), int(ys.max()), int(xs.min()), int(xs.max()))


# (row span, column span) limits of a tap segment for the blur's two LDS window geometries (csrc/dib_common.h: SEG_ROWS /
# SEG_COLS and SEG_ROWS_L / SEG_COLS_L; tests/test_blur_gpu.py checks the library's segments against the same numbers)
STANDARD_WINDOW, LARGE_WINDOW = (12, 24), (20, 63)


def count_tap_segments(psf, limits):
    """How many segments the tap compaction cuts this PSF into for a window geometry: greedy runs of row-major consecutive
    non-zeros whose rows span at most limits[0] + 1 and whose columns span at most limits[1] + 1 (csrc/dib_compact.hip).  A
    host-side scheduling hint (a weight that underflows in the normalisation would drop out on the device)."""
    rr, cc = np.nonzero(psf)
    if len(rr) == 0:
        return 0
    n, r0, lo, hi = 1, int(rr[0]), int(cc[0]), int(cc[0])
    for r, c in zip(rr.tolist(), cc.tolist()):
        nlo, nhi = min(lo, c), max(hi, c)
        if r - r0 > limits[0] or nhi - nlo > limits[1]:
            n, r0, nlo, nhi = n + 1, r, c, c
        lo, hi = nlo, nhi
    return n


def make_psf(param, fraction, center=True):
    """One on-the-fly PSF exactly as the reference builds it (transforms.py:316-335): two trajectory
    fits (the first only advances numpy's global stream), rasterise on a 256 canvas, centre, crop."""
    trajectory = Trajectory(canvas=256, max_len=96, expl=param).fit().fit()
    psf_object = PSF(canvas=256, trajectory=trajectory, fraction=[fraction])
    psf_object.fit()
    if not center:
        return psf_object.PSFs[0]
    psf_object.centerPSF()
    return np.ascontiguousarray(psf_object.PSFs[0][64:128 + 64, 64:128 + 64])


class BlurImage(object):
    def __init__(self, prob=0.5, blur_type=None, blur_exposure=None, use_stored_psfs=False, stored_psf_directory=None,
                 blur_image_in_transform=True, dont_center_psf=False, low_exposure=False, high_exposure=False,
                 dilate_psf=False, LEHE_blur_seg=False, stored_psf_count=STORED_PSF_COUNT):
        self.prob = prob
        self.blur_type = blur_type
        self.blur_exposure = blur_exposure
        self.use_stored_psf = use_stored_psfs
        self.stored_psf_directory = stored_psf_directory
        self.blur_image_in_transform = blur_image_in_transform
        self.dont_center_psf = dont_center_psf
        self.LEHE_blur_seg = LEHE_blur_seg
        self.low_exposure = low_exposure
        self.high_exposure = high_exposure
        self.dilate_psf = dilate_psf
        # the reference hard-codes 12000 files per directory; synthetic stores are smaller
        self.stored_psf_count = stored_psf_count
        if self.blur_image_in_transform:
            print("Blurring internally in transform on CPU")
        else:
            print("Not blurring internally on CPU.")
        self.count = 0

    @staticmethod
    def _not_blurred(blur_dict, with_inverse_warp=False):
        blur_dict["blurring"] = False
        blur_dict["psf"] = [0]
        if with_inverse_warp:
            blur_dict["inverseWarp"] = None
        blur_dict["theta_rad"] = 0
        blur_dict["scale_factor_lambda1"] = 1
        blur_dict["scale_factor_lambda2"] = 1
        blur_dict["param_index"] = None
        blur_dict["fraction_index"] = None
        return blur_dict

    def _exposure_index(self, stored):
        # random.choice(range(n)) and random.choice([..]) of the same length consume the stream alike
        if self.high_exposure:
            return random.choice([3, 4])
        if self.low_exposure:
            return random.choice([0, 1, 2])
        if self.LEHE_blur_seg:
            return random.choices([0, 1, 2, 3, 4], weights=[0.0625, 0.0625, 0.0625, 0.375, 0.375])[0]
        return random.choice([0, 1, 2, 3, 4])

    def _load_stored(self, param_index, fraction_index, psf_index):
        """One stored PSF as a 128 x 128 float16 array: from the packed per-directory file written by
        dataset_utils/generate_PSFs.py --packed when there is one (a memory-mapped row, no open() per
        image), else from the reference's one-file-per-PSF layout (transforms.py:298-309)."""
        key = (param_index, fraction_index)
        packs = self.__dict__.setdefault("_packs", {})
        if key not in packs:
            packed = "%s/P%sE%s.npy" % (self.stored_psf_directory, param_index, fraction_index)
            packs[key] = np.load(packed, mmap_mode="r") if os.path.isfile(packed) else None
        if packs[key] is not None and psf_index < packs[key].shape[0]:
            return np.array(packs[key][psf_index])
        path = "%s/P%sE%s/I%06d" % (self.stored_psf_directory, param_index, fraction_index, psf_index)
        with open(path, "rb") as f:
            psf = np.load(f)
        if psf.shape[0] > 128:
            psf = psf[64:128 + 64, 64:128 + 64]
        return psf

    def __call__(self, image, target=None, blur_dict={}):
        if "preBlurred" in blur_dict and blur_dict["preBlurred"]:      # reference :225-235
            return image, target, self._not_blurred(blur_dict, with_inverse_warp=True)

        threshold = (1 - 0.0625) if self.LEHE_blur_seg else self.prob   # :238-241
        if not random.random() < threshold:                            # :244
            return image, target, self._not_blurred(blur_dict)

        # ---- what to blur with (draw order: fraction, then type) ------------------------------  :248-273
        fraction_index = None
        if self.blur_exposure is not None:
            fraction = self.blur_exposure
        else:
            fraction_index = self._exposure_index(stored=False)
            fraction = FRACTIONS[fraction_index]
        param_index = None
        if self.blur_type is not None:
            param = self.blur_type
        else:
            param_index = random.choice(range(len(PARAMS)))
            param = PARAMS[param_index]

        if self.use_stored_psf:                                         # :276-309
            # stored PSFs live in P{1..3}E{0..4}/I{000000..}; both indices are re-drawn
            param_index = self.blur_type if self.blur_type is not None else random.choice([1, 2, 3])
            if self.blur_exposure is not None:
                fraction_index = self.blur_exposure
            else:
                fraction_index = self._exposure_index(stored=True)
            psf_index = random.randint(0, self.stored_psf_count - 1)
            psf = self._load_stored(param_index, fraction_index, psf_index)
        else:                                                           # :316-335
            psf = make_psf(param, fraction, center=not self.dont_center_psf)

        if self.dilate_psf:                                             # :338-342 (defocus)
            import scipy.ndimage
            sigma = np.random.uniform(low=0, high=3)
            psf = scipy.ndimage.gaussian_filter(psf, sigma)
            psf = psf / psf.max()

        output_image = image
        if self.blur_image_in_transform:                                # --cpu_blur  :344-361
            from .motion_blur.blur_image import BlurImageHandler
            handler = BlurImageHandler(image_path=None, PSFs=[psf.astype(np.float32)], pillowImage=image)
            if not handler.blur_image() or handler.pilImageResult is None:
                print("Error in blurring.")
            output_image = handler.pilImageResult
            self.pilImageResult = output_image

        theta_rad, s1, s2, extent = psf_axis_stats(psf)                 # :366-385
        self.count += 1

        blur_dict["blurring"] = True
        blur_dict["psf"] = psf
        blur_dict["theta_rad"] = theta_rad
        blur_dict["scale_factor_lambda1"] = s1
        blur_dict["scale_factor_lambda2"] = s2
        blur_dict["psf_extent"] = extent
        blur_dict["psf_taps"] = int(np.count_nonzero(psf))
        blur_dict["psf_segments"] = (count_tap_segments(psf, STANDARD_WINDOW), count_tap_segments(psf, LARGE_WINDOW))

        if self.blur_type is not None:                                  # :418-428 nearest-type binning
            param_index = int(np.argmin(np.abs(np.asarray(PARAMS) - self.blur_type)))
        blur_dict["param_index"] = param_index - 1 if self.use_stored_psf else param_index   # :427-435
        if self.blur_exposure is not None:                              # :437-446
            fraction_index = int(np.argmin(np.abs(np.asarray(FRACTIONS) - self.blur_exposure)))
            if self.blur_exposure < 1 / 90:
                fraction_index = -1
        blur_dict["fraction_index"] = fraction_index
        return output_image, target, blur_dict


FUSE_JPEG = True      # False: the module-by-module torch path on the GPU as well (what the fused kernel is tested against)


def add_jpeg_artifact_to_image(image_GPU, jpeg_compressor, quality):
    """Reflect-pad to a multiple of 16, JPEG round trip at `quality`, crop back; returns a Half tensor on the HOST, as the
    reference does (transforms.py:467-493).  A CUDA image and this package's non-differentiable `DiffJPEG` take one HIP
    launch (csrc/dib_jpeg.hip); anything else runs the module."""
    from .models.jpeg import DiffJPEG
    if (FUSE_JPEG and image_GPU.is_cuda and image_GPU.dim() == 3 and image_GPU.shape[0] == 3 and type(jpeg_compressor) is DiffJPEG
            and jpeg_compressor.rounding is torch.round and image_GPU.dtype in (torch.float16, torch.float32)):
        from . import blur_ops
        jpeg_compressor.setQuality(quality)
        f = np.float32(jpeg_compressor.factor)
        tables = jpeg_compressor.__dict__.get("_tables_host")
        if tables is None:        # host copies of the two 8 x 8 tables, once per module (its buffers live on the device)
            tables = jpeg_compressor.__dict__["_tables_host"] = (jpeg_compressor.luma.detach().cpu().numpy().copy(),
                                                                 jpeg_compressor.chroma.detach().cpu().numpy().copy())
        out = blur_ops.jpeg_roundtrip(image_GPU, tables[0] * f, tables[1] * f)
        return out.cpu().detach().squeeze()
    image_GPU = image_GPU.unsqueeze(0)
    w, h = image_GPU.shape[3], image_GPU.shape[2]
    wp, hp = 16 - w % 16, 16 - h % 16
    left, right, top, bottom = math.floor(wp / 2), math.ceil(wp / 2), math.floor(hp / 2), math.ceil(hp / 2)
    padded = torch.nn.functional.pad(image_GPU, (left, right, top, bottom), mode="reflect")
    ph, pw = padded.shape[2], padded.shape[3]
    jpeg_compressor.setQuality(quality)
    jpeg_compressor.setRes(ph, pw)
    comp = jpeg_compressor(padded.float())
    out = comp[:, :, top:ph - bottom, left:pw - right].cpu()
    return out.half().detach().squeeze()
