"""64x64 policy/critic trunk shared by Agent and Value, plus the HIP-backed 64x64 down-sampler.

Trunk (reference agent.py:26-60, value.py:6-44): [Conv2d k4 s2 p1 -> BatchNorm2d -> LeakyReLU(0.2)]
repeated while the map is larger than 4x4, channels mid, 2*mid, ... with the last stage producing
output_dim/16 channels; flattened channel-major to `output_dim`. The layers live in `self.layers`
(an nn.Sequential) so state-dict keys are `layers.{0,1,3,4,6,7,9,10}.*` as in the reference.
These are tiny (64x64 inputs, ~67 MFLOP/img): they stay PyTorch-ROCm modules.
"""
import torch
import torch.nn as nn

from . import _lib


class FeatureExtractor(nn.Module):
    def __init__(self, shape=(16, 64, 64), mid_channels=32, output_dim=4096, dropout_prob=None):
        super().__init__()
        smallest = 4
        assert output_dim % (smallest ** 2) == 0, 'output dim=%d' % output_dim
        self.output_dim = output_dim
        size, cin, cout = int(shape[2]) // 2, int(shape[0]), mid_channels
        stages = [nn.Conv2d(cin, cout, kernel_size=4, stride=2, padding=1), nn.BatchNorm2d(cout),
                  nn.LeakyReLU(negative_slope=0.2)]
        while size > smallest:
            cin = cout
            cout = output_dim // (smallest ** 2) if size == smallest * 2 else cout * 2
            assert size % 2 == 0
            size //= 2
            stages += [nn.Conv2d(cin, cout, kernel_size=4, stride=2, padding=1), nn.BatchNorm2d(cout),
                       nn.LeakyReLU(negative_slope=0.2)]
        self.layers = nn.Sequential(*stages)
        # the agent's trunk ends in dropout (attribute name as in the reference, agent.py:54); the critic's has none
        if dropout_prob is not None:
            self.droupout = nn.Dropout(p=dropout_prob)

    def forward(self, x):
        x = torch.reshape(self.layers(x), [-1, self.output_dim])
        return self.droupout(x) if hasattr(self, "droupout") else x


class _Pool64Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.hw = (int(x.shape[2]), int(x.shape[3]))
        return _lib.pool64(x)

    @staticmethod
    def backward(ctx, grad):
        return _lib.pool64_backward(grad.contiguous(), *ctx.hw)


class Pool64(nn.Module):
    """nn.AdaptiveAvgPool2d((64,64)) as one HIP launch (adaisp_pool64), differentiable like the reference's module:
    the critic's V(retouch, new_states) reaches the filter parameters through this pooling when cfg.use_TD
    (train.py:281-305). Inputs that do not require grad (the dataset images) take the plain launch."""

    def __init__(self, size=(64, 64)):
        super().__init__()
        if tuple(size) != (64, 64):
            raise NotImplementedError("the pooling kernel is built for the reference's 64x64 policy input")

    def forward(self, x):
        if torch.is_grad_enabled() and x.requires_grad:
            return _Pool64Fn.apply(x)
        return _lib.pool64(x.detach())
