"""Non-local-means denoise — mirror of the reference's isp/denoise.py entry points, backed by the
single-launch HIP kernel (csrc/isp_nlm.hip). Only the configuration the reference's ISP uses is
built: gray-weighted NLM with an 11x11 search window and a 5x5 patch (isp/filters.py:577)."""
import torch
import torch.nn as nn

from .. import _lib
from .isp_function import isp_apply

EPS = 1e-8


def rgb_to_luminance(rgb_tensor):
    """0.299 R + 0.587 G + 0.114 B of the clamped image (reference isp/denoise.py:11-17). Small tensors only:
    inside the NLM kernel the luminance is computed on the fly and never materialised."""
    rgb_tensor = torch.clip(rgb_tensor, 0.0, 1.0)
    return 0.299 * rgb_tensor[:, :1] + 0.587 * rgb_tensor[:, 1:2] + 0.114 * rgb_tensor[:, 2:]


class NonLocalMeansGray(nn.Module):
    """forward(rgb, h): rgb [B,3,H,W] in [0,1], h [B,1,1,1] (or [B,1]) filter strength."""

    def __init__(self, search_window_size=11, patch_size=5):
        super().__init__()
        if (search_window_size, patch_size) != (11, 5):
            raise NotImplementedError("the HIP NLM kernel is specialised for search 11 / patch 5, the only "
                                      "configuration on the reference's ISP path (isp/filters.py:577)")
        self.r = search_window_size // 2

    def forward(self, rgb, h):
        return isp_apply(rgb, h.reshape(h.shape[0], -1), _lib.OP_NLM, clip=False)
