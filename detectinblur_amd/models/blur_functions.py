"""`--gpu_blur`: drop-in for the reference's models/blur_functions.py.

Same names, argument meaning and return conventions as the reference:
  * manual_blur(image_GPU, psf_GPU, ...)            reference models/blur_functions.py:11-89
  * blur_image_list(images_GPU, blur_dicts, psfs_GPU, ...)   reference :92-100
The Python roll loop (10 launches + 2 host syncs per tap) is replaced by two stream-ordered
launches per batch through libdib_hip.so: tap compaction and the tiled sparse correlation.
Results are bit-identical to the reference's Half arithmetic (tests/test_blur_gpu.py).
"""
import math

import numpy as np
import torch

from .. import _lib, blur_ops


FUSE_POST_OPS = True      # False: the stock torch ops below on the GPU as well (what the fused kernel is tested against)


def _post_ops(output, add_noise, noise_level, add_block, add_jpeg_artifact, jpeg_compressor):
    """reference models/blur_functions.py:72-87.  The host draws (noise variance, coin flips, scale factor, JPEG quality)
    are the reference's, in its order on numpy's global stream.  On the GPU noise + clamp + block run as ONE HIP pass
    (csrc/dib_postops.hip; block path bit-identical to torch's two `interpolate` calls, noise from the kernel's own
    counter-based generator keyed by a draw from torch's host generator); CPU tensors take the stock torch ops."""
    noise_var = block_scale = None
    if add_noise:
        noise_var = np.random.uniform(0.00000001, noise_level)
    if add_block:
        if np.random.uniform(0, 1) > 0.5:
            block_scale = np.random.uniform(0.6, 1)
    if noise_var is not None or block_scale is not None:
        if FUSE_POST_OPS and output.is_cuda and output.dtype in blur_ops._DT and output.dim() in (2, 3):
            output = blur_ops.post_ops(output, noise_var, block_scale)
        else:
            if noise_var is not None:
                output = torch.clamp(output + (torch.randn_like(output) * math.sqrt(noise_var)), 0, 1)
            if block_scale is not None:
                original_shape = output.shape
                output = torch.nn.functional.interpolate(output.unsqueeze(0), scale_factor=(block_scale, block_scale),
                                                         mode="nearest").squeeze()
                output = torch.nn.functional.interpolate(output.unsqueeze(0), size=original_shape[1:],
                                                         mode="nearest").squeeze()
    if add_jpeg_artifact:
        if np.random.uniform(0, 1) > 0.35:
            quality = np.random.uniform(20, 90)
            from .. import transforms
            output = transforms.add_jpeg_artifact_to_image(output, jpeg_compressor, quality)
    return output


def _check_shapes(image_GPU, K):
    H, W = image_GPU.shape[-2], image_GPU.shape[-1]
    if K <= 129 and not (H < 64 or W < 64) and (H == 64 or W == 64):
        # what F.pad(mode='reflect') raises in the reference (blur_functions.py:59)
        raise RuntimeError("Padding size should be less than the corresponding input dimension, "
                           "but got: padding (63, 64) at dimension of input %s" % list(image_GPU.shape))


def manual_blur(image_GPU, psf_GPU, add_noise=False, noise_level=0.001, add_block=False,
                add_jpeg_artifact=False, jpeg_compressor=None, acc_mode=_lib.DIB_ACC_BITEXACT):
    """image: C x H x W; psf: k x k with k = 128 or 256, already normalised (sums to one).
    Returns the blurred image with every size-1 dim squeezed, like the reference (:69)."""
    K = psf_GPU.shape[0]
    _check_shapes(image_GPU, K)
    if psf_GPU.dtype != image_GPU.dtype:
        psf_GPU = psf_GPU.to(image_GPU.dtype)   # torch's `roll(image) * psf[r, c]` promotes a 0-dim tensor this way
    tables = blur_ops.compact_psfs([psf_GPU], normalize=False, vruns=acc_mode == _lib.DIB_ACC_FAST16)
    out = blur_ops.sparse_blur([image_GPU], [0], tables, acc_mode)[0]
    out = out.squeeze()
    return _post_ops(out, add_noise, noise_level, add_block, add_jpeg_artifact, jpeg_compressor)


def blur_image_list(images_GPU, blur_dicts, psfs_GPU, add_noise=False, noise_level=0.001, add_block=False,
                    add_jpeg_artifact=False, jpeg_compressor=None, acc_mode=_lib.DIB_ACC_BITEXACT, tables=None,
                    psfs_complete=False):
    """In place: images_GPU[i] is replaced by its blurred version when blur_dicts[i]["blurring"].
    PSFs arrive un-normalised and are divided by their sum here (reference :98).  Returns None.
    `tables` (beyond the reference's signature): tap tables of exactly the blurring PSFs, in order, compacted ahead
    of time with blur_ops.compact_psfs_ahead (normalize=True) -- the compaction then overlaps earlier GPU work, and
    `utils.expand_targets(..., tables=)` can share them (what this repo's engine.py does: compaction on a side stream, then
    `blur_quad_f16_kernel` on the finished tables).  Without it -- the reference's own call, engine.py:101 -- the PSFs are
    compacted here, every call, by the one library call that also blurs (`blur_ops.blur_step` -> `dib_blur_step`): ONE launch,
    the grid's first workgroups compact and the blur's workgroups behind them wait inside it (`blur_step_f16_kernel`); a
    hand-off that ever timed out is reported as a RuntimeWarning and the batch re-issued as two launches (include/dib.h,
    "Device status").
    `psfs_complete` (beyond the reference's signature as well): the caller states that the PSF tensors are complete when
    this call is made -- e.g. resident PSFs, or the reference's own `torch.HalfTensor(psf).to(device)` (engine.py:84: a
    synchronous copy) -- and not the product of kernels still queued on the current stream; the compaction then need not
    wait for that stream and overlaps the previous batch's blur."""
    idx = [i for i, bd in enumerate(blur_dicts) if bd["blurring"]]
    if not idx:
        return None
    if tables is not None:
        if tables.count != len(idx) or tables.K != psfs_GPU[idx[0]].shape[0]:
            raise ValueError("tables do not belong to the blurring PSFs of this batch")
        _blur_group(images_GPU, psfs_GPU, idx, acc_mode, blur_dicts, tables)
        if add_noise or add_block or add_jpeg_artifact:
            for i in idx:
                images_GPU[i] = _post_ops(images_GPU[i], add_noise, noise_level, add_block, add_jpeg_artifact, jpeg_compressor)
        return None
    p0 = psfs_GPU[idx[0]]
    K0, dt0 = p0.shape[0], p0.dtype
    mixed = False
    for i in idx:
        p = psfs_GPU[i]
        if p.shape[0] != K0 or p.dtype != dt0:
            mixed = True
            break
    if mixed:
        Ks = {psfs_GPU[i].shape[0] for i in idx}
        dts = {psfs_GPU[i].dtype for i in idx}
        # mixed canvases / dtypes in one batch: one launch group per (K, dtype)
        for K in Ks:
            for dt in dts:
                sub = [i for i in idx if psfs_GPU[i].shape[0] == K and psfs_GPU[i].dtype == dt]
                if sub:
                    _blur_group(images_GPU, psfs_GPU, sub, acc_mode, blur_dicts, psfs_complete=psfs_complete)
    else:
        _blur_group(images_GPU, psfs_GPU, idx, acc_mode, blur_dicts, psfs_complete=psfs_complete)
    if add_noise or add_block or add_jpeg_artifact:
        for i in idx:
            images_GPU[i] = _post_ops(images_GPU[i], add_noise, noise_level, add_block, add_jpeg_artifact,
                                      jpeg_compressor)
    return None


def _blur_group(images_GPU, psfs_GPU, idx, acc_mode, blur_dicts=None, tables=None, psfs_complete=False):
    # (an image that is exactly 64 high or wide under a 128-wide PSF raises in the library what the reference's reflect padding
    # raises: DibError is a RuntimeError carrying "Padding size should be less than the corresponding input dimension")
    psfs = None
    if tables is None:
        psfs = []
        idt = images_GPU[idx[0]].dtype
        for i in idx:
            p = psfs_GPU[i]
            if p.dtype != idt:
                p = p.to(images_GPU[i].dtype)
                psfs_complete = False           # the conversion was just queued on the current stream
            psfs.append(p)
    # Scheduling hint (optional, host-side, never needed for correctness): `BlurImage` records the PSF's
    # tap count in blur_dict["psf_taps"].  Tiles are dispatched in descriptor order, so handing the
    # images over heaviest first lets the launch end on its cheapest tiles (~4 % at BASELINE shapes).
    perm = list(range(len(idx)))
    if blur_dicts is not None:
        try:
            taps = [int(blur_dicts[i]["psf_taps"]) for i in idx]
        except KeyError:
            taps = None
        if taps is not None:
            perm.sort(key=taps.__getitem__, reverse=True)      # stable: equal tap counts keep their order
    if tables is None:
        large = (blur_dicts is not None and len(idx) <= blur_ops.LARGE_WINDOW_MAX_IMAGES
                 and blur_ops.large_window_pays([blur_dicts[i] for i in idx], len(idx)))
        outs = blur_ops.blur_step([images_GPU[idx[k]] for k in perm], perm, psfs, True, acc_mode, psfs_complete, large)
    else:
        outs = blur_ops.sparse_blur([images_GPU[idx[k]] for k in perm], perm, tables, acc_mode)
    for j, k in enumerate(perm):
        o = outs[j]
        images_GPU[idx[k]] = o.squeeze() if 1 in o.shape else o       # reference :69 squeezes every unit dim
