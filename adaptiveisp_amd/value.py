"""Critic network — host-side mirror of the reference's value.py (Value, FeatureExtractor)."""
import torch
import torch.nn as nn

from .nets import FeatureExtractor as _Trunk
from .nets import Pool64
from . import trunk_train


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

    def _planes(self, images, states, pooled):
        """The critic's input: the 64x64 pooling and the state vector extended by the three hand statistics."""
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
        return small, states

    def forward(self, images, states=None, pooled=None):
        """`pooled` (optional) is a precomputed 64x64 pooling of `images` (e.g. the fused output of the
        previous ISP step); it saves one pass over the full-resolution tensor."""
        fe = self.feature_extractor
        small = self.down_sample(images) if pooled is None else pooled
        if trunk_train.serves(fe, small, states, extra=3):
            # statistics + trunk, forward and backward, in one autograd node on the HIP kernels
            feature = trunk_train.trunk_features([fe], [small], [states], critic_planes=True)[0]
        else:
            small, states = self._planes(images, states, small)
            planes = states[:, :, None, None].expand(-1, -1, small.shape[2], small.shape[3])
            feature = fe(torch.cat([small, planes], dim=1))
        return self.fc2(self.lrelu(self.fc1(feature)))

    def forward_pair(self, images_a, states_a, images_b, states_b):
        """(V(images_a, states_a), V(images_b, states_b)) — the two critic calls of an RL iteration (train.py:282-283) — with
        both statistics + trunk passes in ONE autograd node on the HIP kernels (BatchNorm statistics per call, running
        statistics updated call by call, as two module calls would) and the two fully-connected layers on the stacked
        features. Falls back to two plain calls when the kernels do not serve the trunk (eval mode, SyncBatchNorm, CPU)."""
        fe = self.feature_extractor
        small_a, small_b = self.down_sample(images_a), self.down_sample(images_b)
        if not (trunk_train.serves(fe, small_a, states_a, extra=3) and small_a.shape == small_b.shape
                and (states_a is None) == (states_b is None) and (states_a is None or states_a.shape == states_b.shape)):
            return self.forward(images_a, states_a, pooled=small_a), self.forward(images_b, states_b, pooled=small_b)
        feats = trunk_train.trunk_features([fe, fe], [small_a, small_b], [states_a, states_b], share_params=True,
                                           critic_planes=True)
        B = small_a.shape[0]
        v = self.fc2(self.lrelu(self.fc1(feats.view(2 * B, -1))))
        return v[:B], v[B:]
