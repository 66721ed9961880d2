"""JPEG round trip (compress to quantised 8x8 DCT coefficients, decompress) as a tensor op -- the last
stage of the post-blur corruption chain, SURVEY.md section 8f-1: reference models/jpeg/{DiffJPEG,
compression,decompression,utils}.py (itself taken from github.com/mlomnitz/DiffJPEG), driven by
`--add_jpeg_artefacts` through transforms.add_jpeg_artifact_to_image (transforms.py:467-493).

Pipeline on x in [0, 1], N x 3 x H x W with H, W multiples of 16:
    255 x -> YCbCr (JFIF matrix, +128 on chroma) -> 2x2 mean-pooled chroma (4:2:0) -> 8x8 blocks ->
    DCT-II of (block - 128) -> divide by (table * factor), round -> multiply back -> inverse DCT + 128 ->
    blocks to planes -> chroma repeated 2x2 -> RGB -> clamp to [0, 255] -> / 255
with the standard luminance / chrominance tables and factor = (5000 / q) / 100 for q < 50,
(200 - 2 q + 0.01) / 100 otherwise.  Stock torch ops only; the DCTs are one tensordot each with the
8x8x8x8 cosine basis, as in the reference, so that coefficients land on the same side of .5 when rounded.
"""
import itertools
import math

import numpy as np
import torch
from torch import nn

_LUMA = np.array([[16, 11, 10, 16, 24, 40, 51, 61], [12, 12, 14, 19, 26, 58, 60, 55], [14, 13, 16, 24, 40, 57, 69, 56],
                  [14, 17, 22, 29, 51, 87, 80, 62], [18, 22, 37, 56, 68, 109, 103, 77], [24, 35, 55, 64, 81, 104, 113, 92],
                  [49, 64, 78, 87, 103, 121, 120, 101], [72, 92, 95, 98, 112, 100, 103, 99]], dtype=np.float32).T
_CHROMA = np.full((8, 8), 99, dtype=np.float32)
_CHROMA[:4, :4] = np.array([[17, 18, 24, 47], [18, 21, 26, 66], [24, 26, 56, 99], [47, 66, 99, 99]], dtype=np.float32).T


def quality_to_factor(quality):
    q = 5000.0 / quality if quality < 50 else (200.0 - quality * 2) + 0.01
    return q / 100.0


def diff_round(x):
    """round(x) with a cubic bump so that gradients flow (the `differentiable=True` variant)."""
    return torch.round(x) + (x - torch.round(x)) ** 3


def _cos_basis(forward):
    t = np.zeros((8, 8, 8, 8), dtype=np.float32)
    for a, b, c, d in itertools.product(range(8), repeat=4):
        if forward:     # [x, y, u, v]
            t[a, b, c, d] = np.cos((2 * a + 1) * c * np.pi / 16) * np.cos((2 * b + 1) * d * np.pi / 16)
        else:           # [u, v, x, y]
            t[a, b, c, d] = np.cos((2 * c + 1) * a * np.pi / 16) * np.cos((2 * d + 1) * b * np.pi / 16)
    return torch.from_numpy(t)


class DiffJPEG(nn.Module):
    def __init__(self, height, width, differentiable=False, quality=80):
        super().__init__()
        self.height, self.width = height, width
        self.rounding = diff_round if differentiable else torch.round
        self.factor = quality_to_factor(quality)
        alpha = np.array([1.0 / np.sqrt(2)] + [1] * 7)
        self.register_buffer("to_ycc", torch.from_numpy(np.array([[0.299, 0.587, 0.114], [-0.168736, -0.331264, 0.5],
                                                                   [0.5, -0.418688, -0.081312]], dtype=np.float32).T))
        self.register_buffer("ycc_shift", torch.tensor([0.0, 128.0, 128.0]))
        self.register_buffer("to_rgb", torch.from_numpy(np.array([[1.0, 0.0, 1.402], [1, -0.344136, -0.714136], [1, 1.772, 0]],
                                                                  dtype=np.float32).T))
        self.register_buffer("rgb_shift", torch.tensor([0.0, -128.0, -128.0]))
        self.register_buffer("dct", _cos_basis(True))
        self.register_buffer("idct", _cos_basis(False))
        self.register_buffer("dct_scale", torch.from_numpy(np.outer(alpha, alpha) * 0.25).float())
        self.register_buffer("idct_alpha", torch.from_numpy(np.outer(alpha, alpha)).float())
        self.register_buffer("luma", torch.from_numpy(_LUMA.copy()))
        self.register_buffer("chroma", torch.from_numpy(_CHROMA.copy()))

    def setQuality(self, quality):
        self.factor = quality_to_factor(quality)

    def setRes(self, new_height, new_width):
        self.height, self.width = new_height, new_width

    # ---- helpers ------------------------------------------------------------------------------------
    @staticmethod
    def _blocks(plane):                       # [B, H, W] -> [B, H*W/64, 8, 8]
        B, H, W = plane.shape
        return plane.view(B, H // 8, 8, -1, 8).permute(0, 1, 3, 2, 4).contiguous().view(B, -1, 8, 8)

    @staticmethod
    def _planes(blocks, H, W):                # inverse of _blocks
        B = blocks.shape[0]
        return blocks.view(B, H // 8, W // 8, 8, 8).permute(0, 1, 3, 2, 4).contiguous().view(B, H, W)

    def _code(self, plane, table):
        coef = self.dct_scale * torch.tensordot(self._blocks(plane) - 128, self.dct, dims=2)
        return self.rounding(coef.float() / (table * self.factor))

    def _decode(self, q, table, H, W):
        coef = q * (table * self.factor)
        return self._planes(0.25 * torch.tensordot(coef * self.idct_alpha, self.idct, dims=2) + 128, H, W)

    def forward(self, x):
        ycc = torch.tensordot((x * 255).permute(0, 2, 3, 1), self.to_ycc, dims=1) + self.ycc_shift     # [B, H, W, 3]
        pooled = nn.functional.avg_pool2d(ycc.permute(0, 3, 1, 2)[:, 1:3], kernel_size=2, stride=2, count_include_pad=False)
        y = self._decode(self._code(ycc[..., 0], self.luma), self.luma, self.height, self.width)
        h2, w2 = int(self.height / 2), int(self.width / 2)
        cb = self._decode(self._code(pooled[:, 0], self.chroma), self.chroma, h2, w2)
        cr = self._decode(self._code(pooled[:, 1], self.chroma), self.chroma, h2, w2)
        up = lambda c: c.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)                          # noqa: E731
        img = torch.stack([y, up(cb), up(cr)], dim=3)
        rgb = torch.tensordot(img + self.rgb_shift, self.to_rgb, dims=1).permute(0, 3, 1, 2)
        return torch.min(255 * torch.ones_like(rgb), torch.max(torch.zeros_like(rgb), rgb)) / 255
