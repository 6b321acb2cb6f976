"""Non-local-means denoise — mirror of the reference's isp/denoise.py entry points, backed by the HIP kernels of
csrc/isp_nlm.hip: the tuned single-launch kernel for the configuration the reference's ISP uses (gray-weighted NLM, 11x11
search window, 5x5 patch: isp/filters.py:577) and a plain gather kernel for every other odd (search, patch) pair the class
accepts (its own default is 21 / 7, isp/denoise.py:94)."""
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
        if search_window_size % 2 != 1 or patch_size % 2 != 1 or search_window_size < 1 or patch_size < 1:
            raise ValueError("window size must be odd")                  # BoxFilter's assertion, isp/denoise.py:52
        self.search_window_size, self.patch_size = int(search_window_size), int(patch_size)
        self.r = search_window_size // 2

    def forward(self, rgb, h):
        if (self.search_window_size, self.patch_size) == (11, 5):
            # the ISP's configuration: tuned kernel, differentiable in h (rgb in [0,1], as DenoiseFilter.process hands it over)
            return isp_apply(rgb, h.reshape(h.shape[0], -1), _lib.OP_NLM, clip=False)
        if torch.is_grad_enabled() and (h.requires_grad or rgb.requires_grad):
            raise NotImplementedError("only the 11 / 5 configuration (the one on the training path) has a backward kernel")
        return _lib.nlm_general(rgb, h.reshape(h.shape[0], -1), self.search_window_size, self.patch_size)
