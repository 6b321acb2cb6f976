"""Critic network — host-side mirror of the reference's value.py (Value, FeatureExtractor)."""
import torch
import torch.nn as nn

from .nets import FeatureExtractor as _Trunk
from .nets import Pool64


class FeatureExtractor(_Trunk):
    def __init__(self, shape=(17, 64, 64), mid_channels=32, output_dim=4096):
        super().__init__(shape=shape, mid_channels=mid_channels, output_dim=output_dim, dropout_prob=None)


class Value(nn.Module):
    """V(image, states) -> [B,1] (reference value.py:48-99). Input planes: the 64x64-pooled image, the
    state vector and three hand-made statistics (mean luminance, luminance variance, mean saturation)."""

    def __init__(self, cfg, shape=(19, 64, 64)):
        super().__init__()
        self.cfg = cfg
        self.feature_extractor = FeatureExtractor(shape=shape, mid_channels=cfg.base_channels,
                                                  output_dim=cfg.feature_extractor_dims)
        self.fc1 = nn.Linear(cfg.feature_extractor_dims, cfg.fc1_size)
        self.lrelu = nn.LeakyReLU(negative_slope=0.2)
        self.fc2 = nn.Linear(cfg.fc1_size, 1)
        self.tanh = nn.Tanh()          # defined, not applied (as in the reference)
        self.down_sample = Pool64((shape[1], shape[2]))

    def forward(self, images, states=None, pooled=None):
        """`pooled` (optional) is a precomputed 64x64 pooling of `images` (e.g. the fused output of the
        previous ISP step); it saves one pass over the full-resolution tensor."""
        small = self.down_sample(images) if pooled is None else pooled
        lum = (small[:, 0] * 0.27 + small[:, 1] * 0.67 + small[:, 2] * 0.06 + 1e-5)[:, None]
        luminance = torch.mean(lum, dim=(1, 2, 3))
        contrast = torch.var(lum, dim=(1, 2, 3))
        clipped = torch.clip(small, min=0.0, max=1.0)
        i_max, i_min = clipped.max(dim=1)[0], clipped.min(dim=1)[0]
        sat = (i_max - i_min) / (torch.minimum(i_max + i_min, 2.0 - i_max - i_min) + 1e-2)
        saturation = torch.mean(sat, dim=[1, 2])
        stats = torch.stack([luminance, contrast, saturation], dim=1)
        if states is None:
            states = stats
        else:
            assert states.dim() == stats.dim()
            states = torch.cat([states, stats], dim=1)
        planes = states[:, :, None, None].expand(-1, -1, small.shape[2], small.shape[3])
        feature = self.feature_extractor(torch.cat([small, planes], dim=1))
        return self.fc2(self.lrelu(self.fc1(feature)))
