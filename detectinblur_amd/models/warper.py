""""Squint" warper (`--warp_in_model`): stretches the image along the blur's principal axes before
the backbone and un-stretches every feature map after it -- SURVEY.md section 8f-2, reference
models/warper.py:13-52 and models/generalized_rcnn.py:131-141.  It consumes the
`theta_rad / scale_factor_lambda1 / scale_factor_lambda2` that `transforms.BlurImage` already
computes for the hot path.

The transform, per image, in homogeneous 3 x 3 form (all matrices Half, as the reference keeps them):
    S = diag(l1, l2, 1)                                   anisotropic scale
    R = rotation by -theta
    T = identity with (width, height) in the LAST ROW     (the reference's placement, kept as is)
    F = R @ T,  G = S @ F,  M = inv(inv(F) @ G)           both inverses in float32, results cast to Half
and the top two rows of M drive `affine_grid` + bilinear `grid_sample` (zeros outside, align_corners
False), computed in Half and returned as float32.  Stock PyTorch ops only.
"""
import torch
from torch import nn
import torch.nn.functional as F


def squint_matrices(thetas, lambda1s, lambda2s, width, height):
    """[B, 2, 3] Half affine matrices of the warp described in the module docstring."""
    B = lambda1s.shape[0]
    dt, dev = lambda1s.dtype, lambda1s.device

    def eye():
        m = torch.zeros((B, 3, 3), dtype=dt, device=dev)
        m[:, 0, 0] = 1; m[:, 1, 1] = 1; m[:, 2, 2] = 1
        return m

    scale = eye()
    scale[:, 0, 0] = lambda1s
    scale[:, 1, 1] = lambda2s
    t = -thetas
    c, s = torch.cos(t), torch.sin(t)
    rot = eye()
    rot[:, 0, 0] = c; rot[:, 0, 1] = -s
    rot[:, 1, 0] = s; rot[:, 1, 1] = c
    trans = eye()
    trans[:, 2, 0] = torch.ones_like(lambda1s) * width
    trans[:, 2, 1] = torch.ones_like(lambda1s) * height
    fwd = torch.bmm(rot, trans)
    fwd_scaled = torch.bmm(scale, fwd)
    overall = torch.bmm(torch.inverse(fwd.float()).to(dt), fwd_scaled)
    overall = torch.inverse(overall.float()).to(dt)
    return overall[:, 0:2, :]


class Warper(nn.Module):
    def forward(self, x, thetas, lambda1s, lambda2s):
        height, width = x.shape[-2], x.shape[-1]
        m = squint_matrices(thetas, lambda1s, lambda2s, width, height)
        grid = F.affine_grid(theta=m, size=x.shape, align_corners=False).float().half()
        out = F.grid_sample(x.half(), grid, mode="bilinear", padding_mode="zeros", align_corners=False)
        return out.float()
