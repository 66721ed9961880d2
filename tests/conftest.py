import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The DataLoader workers of the driver tests come from a fork server that must exist before this process touches the
    # GPU (detectinblur_amd.utils.loader_context: a fork of a GPU process stalls the GPU for tens of seconds).
    from detectinblur_amd import kernel_choices, utils
    utils.loader_context()
    # the detector tests compare runs with each other and with pinned numbers: same kernel choices in every process
    kernel_choices.use_shipped_kernel_choices()


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, e.g. a plain `pytest tests/`
    in the CPU container; `-m gpu` on the GPU box runs them."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    import json
    d = os.path.join(ROOT, "tests", "golden")

    class G:
        traj = np.load(os.path.join(d, "traj.npz"))
        psf = np.load(os.path.join(d, "psf.npz"))
        blur = np.load(os.path.join(d, "blur.npz"))
        boxes = np.load(os.path.join(d, "boxes.npz"))
        norm = np.load(os.path.join(d, "norm.npz"))
        fft = np.load(os.path.join(d, "fft.npz"))
        blurdict = np.load(os.path.join(d, "blurdict.npz"))
        warper = np.load(os.path.join(d, "warper.npz"))
        jpeg = np.load(os.path.join(d, "jpeg.npz"))
        coco = np.load(os.path.join(d, "coco.npz"))
        meta = json.load(open(os.path.join(d, "meta.json")))
    return G
