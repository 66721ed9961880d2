"""Offline PSF store: the builder must write the reference's files
byte for byte (SHA-256 of every file recorded from the real dataset_utils/generate_PSFs.py by
oracle/gen_goldens.py), and BlurImage must read both layouts."""
import hashlib
import os
import random

import numpy as np
import pytest

from detectinblur_amd.dataset_utils import generate_PSFs as G

RUNS = {"w0_of1_n2": (0, 1, 2), "w1_of2_n4": (1, 2, 4)}


def _digests(root):
    out = {}
    for d, _, files in os.walk(os.path.join(root, "psfs")):
        for fn in files:
            if fn.endswith(".npy"):
                continue          # packed extras are not part of the reference layout
            full = os.path.join(d, fn)
            with open(full, "rb") as f:
                out[os.path.relpath(full, root)] = hashlib.sha256(f.read()).hexdigest()
    return out


def _build(tmp_path, name, extra=()):
    w, nw, tot = RUNS[name]
    dest = str(tmp_path / name) + "/"
    os.makedirs(dest)
    args = G.get_parser().parse_args(["--destination_path", dest, "--worker_index", str(w), "--num_workers", str(nw),
                                      "--total_num_psfs", str(tot)] + list(extra))
    assert G.main(args) == 15 * (tot // nw)
    return dest


@pytest.mark.parametrize("name", sorted(RUNS))
def test_store_is_byte_identical_to_reference(golden, tmp_path, name, capsys):
    dest = _build(tmp_path, name)
    want = golden.meta["psf_store"][name]
    got = _digests(dest)
    assert sorted(got) == sorted(want)
    assert got == want


def test_packed_store_and_blurimage_reader(tmp_path, capsys):
    from detectinblur_amd import transforms as T
    dest = _build(tmp_path, "w0_of1_n2", ["--packed"])
    store = dest + "psfs"
    packed = np.load(store + "/P2E3.npy")
    assert packed.shape == (2, 128, 128) and packed.dtype == np.float16
    for idx in range(2):
        with open("%s/P2E3/I%06d" % (store, idx), "rb") as f:
            full = np.load(f)
        assert full.shape == (256, 256) and full.dtype == np.float16
        assert np.array_equal(packed[idx], full[64:192, 64:192])
    # the transform draws the same indices and returns the same PSF from either layout
    results = []
    for use_packed in (True, False):
        if not use_packed:
            for p in range(1, 4):
                for e in range(5):
                    os.remove("%s/P%dE%d.npy" % (store, p, e))
        random.seed(7); np.random.seed(7)
        t = T.BlurImage(prob=1.0, use_stored_psfs=True, stored_psf_directory=store, blur_image_in_transform=False,
                        stored_psf_count=2)
        res = [t(None, None, {})[2] for _ in range(6)]
        results.append(res)
    for a, b in zip(*results):
        assert np.array_equal(a["psf"], b["psf"]) and a["psf"].shape == (128, 128)
        assert a["param_index"] == b["param_index"] and a["fraction_index"] == b["fraction_index"]
        assert a["theta_rad"] == b["theta_rad"]


@pytest.mark.gpu
def test_store_built_on_gpu_is_byte_identical(golden, tmp_path, capsys):
    dest = _build(tmp_path, "w1_of2_n4", ["--device", "cuda"])
    assert _digests(dest) == golden.meta["psf_store"]["w1_of2_n4"]
