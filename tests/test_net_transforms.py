"""A12: the detector's input transform against the REFERENCE's own models/net_transforms.py (tests/golden/
net_transforms.npz, written by oracle/gen_goldens.py from the imported reference with torchvision's `_is_tracing`
and `ImageList` supplied by the harness): normalise with per-image statistics -> resize -> zero-padded batch, box
rescaling, the blur estimator's crop batcher, eval-mode postprocess.  Bit-exact on the CPU; on the GPU within 2e-6
absolute (the bilinear resize runs in a different instruction order there; inputs are O(1) after normalisation)."""
import os

import numpy as np
import pytest
import torch

import gen_goldens as GG
from detectinblur_amd.models.net_transforms import GeneralizedRCNNTransform

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "net_transforms.npz"))
MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def _run(device, channels_last=False):
    imgs, tgts, means, stds = GG.net_transform_inputs()
    out = {}
    for tag, training, use_stats in (("train", True, True), ("eval", False, False)):
        t = GeneralizedRCNNTransform(64, 100, MEAN, STD, training=training)
        t.channels_last = channels_last
        torch.manual_seed(5)
        src = [{k: v.clone().to(device) for k, v in d.items()} for d in tgts]
        il, res = t([i.clone().to(device) for i in imgs], src, newMeans=means if use_stats else None,
                    newSTDs=stds if use_stats else None)
        out["nt_%s_batch" % tag] = il.tensors
        out["nt_%s_sizes" % tag] = torch.tensor(il.image_sizes)
        for k, d in enumerate(res):
            out["nt_%s_boxes%d" % (tag, k)] = d["boxes"]
            assert torch.equal(src[k]["boxes"].cpu(), tgts[k]["boxes"])           # the caller's targets stay untouched
    t = GeneralizedRCNNTransform(64, 100, MEAN, STD, training=False)
    one = torch.rand(3, 64, 100, generator=torch.Generator().manual_seed(8))
    out["nt_unit_batch"] = t([one.to(device)], None)[0].tensors
    t = GeneralizedRCNNTransform(64, 100, MEAN, STD, crop_images=True)
    torch.manual_seed(5)
    il, _ = t([i.clone().to(device) for i in imgs], None)
    out["nt_crop_batch"], out["nt_crop_sizes"] = il.tensors, torch.tensor(il.image_sizes)
    t = GeneralizedRCNNTransform(64, 100, MEAN, STD, training=False)
    res = t.postprocess([{"boxes": torch.tensor([[4.0, 8.0, 60.0, 50.0]], device=device)},
                         {"boxes": torch.tensor([[2.0, 2.0, 40.0, 90.0]], device=device)}],
                        [(64, 90), (100, 75)], [(50, 70), (60, 45)])
    for k, d in enumerate(res):
        out["nt_post_boxes%d" % k] = d["boxes"]
    return {k: v.detach().cpu().numpy() for k, v in out.items()}


def test_matches_reference_bit_for_bit_on_cpu():
    got = _run("cpu")
    assert sorted(got) == sorted(G.files)
    for k in G.files:
        assert got[k].shape == G[k].shape and np.array_equal(got[k], G[k]), k
    # the channels-last batch the MI355X model uses holds the same values
    cl = _run("cpu", channels_last=True)
    for k in G.files:
        assert np.array_equal(cl[k], G[k]), k


@pytest.mark.gpu
def test_matches_reference_on_gpu():
    got = _run("cuda", channels_last=True)
    for k in G.files:
        if k.endswith("sizes"):
            assert np.array_equal(got[k], G[k]), k
        else:
            assert got[k].shape == G[k].shape and np.abs(got[k] - G[k]).max() <= 2e-6 * max(1.0, np.abs(G[k]).max()), k
