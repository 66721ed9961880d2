"""Offline PSF-store builder -- drop-in for the reference's dataset_utils/generate_PSFs.py.

Same command line (`--destination_path --worker_index --num_workers --total_num_psfs`), same seeds
(`np.random.seed(1337 * worker_index)`, reference :29-30), same draw order (type-major, then
exposure, then index; two trajectory fits per PSF, :46-47) and the same on-disk format
(`<dest>psfs/P{1..3}E{0..4}/I{index:06d}`, `np.save` of the centred
256 x 256 PSF as float16, no extension, :58-60): the files are byte-identical to the reference's.

What changes is the cost: trajectory walk, rasteriser and centring run in native code
(libdib_host.so, ~0.6 ms per PSF instead of ~150 ms), so the reference's 180,000-file store is a
couple of minutes of one core instead of a day of twelve workers.  Two optional extras:

  --device cuda   rasterise + centre on the GPU in batches (libdib_hip.so, bit-identical float64;
                  the trajectories still come from the host stream, they are sequential by nature)
  --packed        additionally write one `<dest>psfs/P{p}E{e}.npy` per directory: a
                  [count, 128, 128] float16 array of the centre crops `BlurImage` actually uses
                  (transforms.py:307-309), readable with np.load(mmap_mode="r") -- one file instead of
                  12,000, 393 MB instead of 1.5 GB per directory.  `BlurImage` prefers it when present.
"""
import argparse
import os
import random
import time

import numpy as np

from ..motion_blur.generate_PSF import PSF
from ..motion_blur.generate_trajectory import Trajectory

PARAMS = [0.005, 0.001, 0.00005]
FRACTIONS = [1 / 18, 1 / 10, 1 / 5, 1 / 2, 1]
GPU_BATCH = 256


def _host_psf(param, exposure):
    trajectory = Trajectory(canvas=256, max_len=96, expl=param).fit().fit()
    psf_object = PSF(canvas=256, trajectory=trajectory, fraction=[exposure])
    psf_object.fit()
    psf_object.centerPSF()
    return psf_object.PSFs[0]


def _gpu_psfs(param, exposure, count):
    """`count` PSFs of one (type, exposure): trajectories from the host stream, everything else on
    the GPU; yields float64 256 x 256 arrays in order."""
    import torch
    from .. import blur_ops
    done = 0
    while done < count:
        n = min(GPU_BATCH, count - done)
        traj = np.empty((n, 2000), dtype=np.complex128)
        for i in range(n):
            traj[i] = Trajectory(canvas=256, max_len=96, expl=param).fit().fit().x
        psf64, _ = blur_ops.rasterize_psfs(torch.from_numpy(traj), [exposure] * n, canvas=256, center=True, out_n=256,
                                           want64=True, want16=False)
        for a in psf64.cpu().numpy():
            yield a
        done += n


def main(args):
    path = args.destination_path
    slice_index = args.worker_index
    slice_size = int(args.total_num_psfs / args.num_workers)
    start_index = slice_size * slice_index
    end_index = start_index + slice_size

    np.random.seed(1337 * slice_index)
    random.seed(1337 * slice_index)

    for param_index in range(len(PARAMS)):
        for fraction_index in range(len(FRACTIONS)):
            os.makedirs(path + "psfs/P" + str(param_index + 1) + "E" + str(fraction_index), exist_ok=True)

    start_time = time.perf_counter()
    written = 0
    for param_index, param in enumerate(PARAMS):
        for exposure_index, exposure in enumerate(FRACTIONS):
            folder_path = path + "psfs/P" + str(param_index + 1) + "E" + str(exposure_index)
            source = (_gpu_psfs(param, exposure, slice_size) if args.device == "cuda"
                      else (_host_psf(param, exposure) for _ in range(slice_size)))
            packed = np.empty((slice_size, 128, 128), dtype=np.float16) if args.packed else None
            for index, psf in zip(range(start_index, end_index), source):
                half = psf.astype(np.float16)
                with open(folder_path + "/I" + "{:06d}".format(index), "wb") as f:
                    np.save(f, half)
                if packed is not None:
                    packed[index - start_index] = half[64:128 + 64, 64:128 + 64]
                written += 1
                if index % 200 == 0:
                    elapsed = time.perf_counter() - start_time
                    print("Image Index %d %d %d Elapsed Time: %s Average Time/Image: %.2fms"
                          % (param_index, exposure_index, index, time.strftime("%H:%M:%S", time.gmtime(elapsed)),
                             elapsed * 1000 / max(written, 1)))
            if packed is not None:
                name = folder_path + (".npy" if args.num_workers == 1 else ".w%03d.npy" % slice_index)
                np.save(name, packed)
    return written


def get_parser():
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    parser.add_argument("--destination_path", type=str, default="./")
    parser.add_argument("--worker_index", type=int, default=0)
    parser.add_argument("--num_workers", type=int, default=12)
    parser.add_argument("--total_num_psfs", type=int, default=12000)
    parser.add_argument("--device", choices=["cpu", "cuda"], default="cpu")
    parser.add_argument("--packed", action="store_true")
    return parser


if __name__ == "__main__":
    main(get_parser().parse_args())
