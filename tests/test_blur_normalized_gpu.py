"""dib_sparse_blur_normalized -- the blur with the input transform's float + normalise + zero-padded batch as its store phase
(reference engine.py:101, :107-110 + models/net_transforms.py:112-121, :238-247) -- against the two launches it replaces
(dib_sparse_blur + dib_normalize_pad), bit for bit; and the training loop's opt-in that uses it."""
import numpy as np
import pytest
import torch

import dib_oracle as O

pytestmark = pytest.mark.gpu


def _psf(rs, n, spread):
    a = np.zeros((128, 128), np.float64)
    a[np.clip(rs.randint(-spread, spread + 1, n) + 63, 0, 127), np.clip(rs.randint(-spread, spread + 1, n) + 63, 0, 127)] = rs.random_sample(n) + 0.05
    return torch.from_numpy(O.to_half_like_torch(a)).cuda()


def _rows(rs, n):
    return rs.uniform(0.2, 0.6, (n, 3)), rs.uniform(0.15, 0.35, (n, 3))


@pytest.mark.parametrize("channels_last", [False, True])
@pytest.mark.parametrize("shape", [(8, 800, 1333, 800, 1344), (3, 96, 250, 96, 256), (5, 70, 130, 96, 256), (2, 64 + 1, 128, 96, 128)])
def test_blur_normalized_equals_blur_then_normalize_pad(shape, channels_last):
    from detectinblur_amd import _lib, blur_ops
    B, H, W, Hp, Wp = shape
    rs = np.random.RandomState(B * 1000 + H)
    images = [torch.from_numpy(rs.random_sample((3, H, W)).astype(np.float16)).cuda() for _ in range(B)]
    psfs = [_psf(rs, 5 + 9 * i, 2 + 3 * i) for i in range(B)]
    tables = blur_ops.compact_psfs(psfs, normalize=True)
    means, stds = _rows(rs, B)
    index = list(range(B))
    order = sorted(range(B), key=lambda i: -i)
    for acc in (_lib.DIB_ACC_BITEXACT, _lib.DIB_ACC_FMA16):
        want = blur_ops.normalize_pad(blur_ops.sparse_blur(list(images), index, tables, acc), means, stds, Hp, Wp, channels_last)
        got = blur_ops.sparse_blur_normalized(images, index, tables, means, stds, Hp, Wp, channels_last, acc, order=order)
        assert got is not None and got.shape == want.shape and got.stride() == want.stride()
        assert torch.equal(got, want)                  # every pixel AND every padding zero


def test_blur_normalized_with_ragged_images_and_not_served_batches():
    from detectinblur_amd import blur_ops
    rs = np.random.RandomState(7)
    sizes = [(90, 250), (96, 256), (70, 200)]            # one padded batch of 96 x 256: every image's tiles cover it
    images = [torch.from_numpy(rs.random_sample((3, h, w)).astype(np.float16)).cuda() for h, w in sizes]
    psfs = [_psf(rs, 12, 5) for _ in sizes]
    tables = blur_ops.compact_psfs(psfs, normalize=True)
    means, stds = _rows(rs, 3)
    want = blur_ops.normalize_pad(blur_ops.sparse_blur(list(images), [0, 1, 2], tables), means, stds, 96, 256, True)
    got = blur_ops.sparse_blur_normalized(images, [0, 1, 2], tables, means, stds, 96, 256, True)
    assert got is not None and torch.equal(got, want)
    # not served: an image that is not blurred; a padded extent beyond an image's own tiles
    assert blur_ops.sparse_blur_normalized(images, [0, -1, 2], tables, means, stds, 96, 256, True) is None
    assert blur_ops.sparse_blur_normalized(images, [0, 1, 2], tables, means, stds, 128, 256, True) is None


def test_training_loop_opt_in_gives_the_same_weights(monkeypatch):
    """engine.train_one_epoch with FUSE_BLUR_EPILOGUE on and off: same losses and weights after three steps, bit for bit (the toy
    detector of the reference pins with this repo's input transform in front; 96 x 160 images, min_size 96: no resize, so the fused
    launch serves every batch), and the transform reports which path it took."""
    import contextlib
    import io
    import pin_inputs as PI
    from detectinblur_amd import engine
    from detectinblur_amd.models.net_transforms import GeneralizedRCNNTransform

    class Wrapped(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.core = PI.ToyDetector(3)
            self.transform = GeneralizedRCNNTransform(96, 160, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
            self.transform.channels_last = True

        def forward(self, images, targets=None, thetas=None, lambda1s=None, lambda2s=None, newMeans=None, newSTDs=None):
            batch, targets = self.transform(images, targets, newMeans, newSTDs)
            imgs = [batch.tensors[i] * 0.25 + 0.5 for i in range(batch.tensors.shape[0])]
            n = len(imgs)
            return self.core(imgs, targets, newMeans=np.zeros((n, 3)), newSTDs=np.ones((n, 3)))

    def loader():
        rs = np.random.RandomState(99)
        out = PI.ListLoader()
        for k in range(3):
            images = tuple(torch.from_numpy(rs.random_sample((3, 96, 160)).astype(np.float32)) for _ in range(2))
            targets = tuple(PI._target(rs, 96, 160, 3, 10 * k + j) for j in range(2))
            dicts = tuple(dict(PI._blur_dict(rs, (k + j) % 3, (2 * k + j) % 5, True), psf_taps=5 + j) for j in range(2))
            out.append((images, targets, dicts))
        return out

    def run(flag):
        monkeypatch.setattr(engine, "FUSE_BLUR_EPILOGUE", flag)
        torch.manual_seed(0)
        np.random.seed(0)
        model = Wrapped().cuda()
        opt = torch.optim.SGD(model.parameters(), lr=0.04, momentum=0.9)
        with contextlib.redirect_stdout(io.StringIO()):
            engine.train_one_epoch(model, opt, loader(), torch.device("cuda"), epoch=1, print_freq=10, writer=PI.RecordingWriter(),
                                   blur_train=True, gpu_blur=True, expand_target_boxes=True, use_custom_image_norm=True)
        return {k: v.detach().clone() for k, v in model.state_dict().items()}, getattr(model.transform, "last_epilogue", None)

    plain, how_plain = run(False)
    fused, how_fused = run(True)
    assert how_plain is None and how_fused == "fused into the blur"
    for k in plain:
        assert torch.equal(plain[k], fused[k]), k
