"""Filter modules of the ISP stack — host-side mirror of the reference's isp/filters.py.

Same class names, constructor signatures, attributes and state-dict keys as the reference
(`Filter.process(img, param)`, `Filter.forward(img, img_features, specified_parameter, high_res)`,
heads `fc1 / fc_filter / fc_mask`), so callers written against it run unchanged. What differs is
where the pixels are touched: `process` is ONE HIP kernel launch from csrc/libadaisp.so
(op code = `Filter.op_code`, include/adaisp.h) instead of a chain of ATen ops, and `forward` fuses
the final clip into the same launch. The small [B,n] parameter regressions (tanh_range, sigmoid,
exp) stay in PyTorch so autograd links parameter gradients back to the heads; image gradients are
not produced (the reference's training never needs them: train.py:341-342, imgs is a constant).
"""
import math

import torch
import torch.nn as nn

from .. import _lib
from .isp_function import isp_apply


def rgb2lum(image):
    """0.27 R + 0.67 G + 0.06 B, channel dim kept (reference isp/filters.py:12-14). Small tensors only."""
    return (0.27 * image[:, 0] + 0.67 * image[:, 1] + 0.06 * image[:, 2])[:, None]


def lerp(a, b, l):
    return (1 - l) * a + l * b


def tanh01(x):
    return torch.tanh(x) * 0.5 + 0.5


def tanh_range(l, r, initial=None):
    """Squash to (l, r); `initial` is the value produced by a zero input (reference isp/filters.py:25-34)."""
    bias = 0 if initial is None else math.atanh(2 * (initial - l) / (r - l) - 1)

    def activation(x):
        return tanh01(x + bias) * (r - l) + l

    return activation


class Filter(nn.Module):
    """Base class (reference isp/filters.py:37-212)."""

    op_code = None  # kernel op code, include/adaisp.h

    def __init__(self, cfg, short_name, num_filter_parameters, predict=False):
        super().__init__()
        self.cfg = cfg
        self.channels = 3
        self.num_filter_parameters = num_filter_parameters
        self.short_name = short_name
        self.filter_parameters = None
        self.mask = None
        self.mask_parameters = None
        self.predict = predict
        if predict:
            self.fc1 = nn.Linear(cfg.feature_extractor_dims, cfg.fc1_size)
            self.lrelu = nn.LeakyReLU(negative_slope=0.2)
            self.fc_filter = nn.Linear(cfg.fc1_size, self.get_num_filter_parameters())
            self.fc_mask = nn.Linear(cfg.fc1_size, self.get_num_mask_parameters())

    # -- introspection ---------------------------------------------------------------------------
    def get_short_name(self):
        assert self.short_name
        return self.short_name

    def get_num_filter_parameters(self):
        assert self.num_filter_parameters
        return self.num_filter_parameters

    def get_num_mask_parameters(self):
        return 6

    def use_masking(self):
        return False

    def debug_info_batched(self):
        return False

    def no_high_res(self):
        return False

    # -- parameter heads ---------------------------------------------------------------------------
    def extract_parameters(self, features):
        hidden = self.lrelu(self.fc1(features))
        return self.fc_filter(hidden), self.fc_mask(hidden)

    def filter_param_regressor(self, features):
        raise AssertionError("filter_param_regressor must be implemented by the filter class")

    # (kind, lo, hi, initial) of the regressor for the fused policy kernel (kinds: include/adaisp.h); None = no fused form
    _regressor = None

    def regressor_spec(self):
        """(op, n, kind, lo, scale, bias) consumed by adaisp_policy_finish; mirrors filter_param_regressor."""
        if self._regressor is None:
            raise NotImplementedError(f"{type(self).__name__} has no fused regressor")
        kind, lo, hi, initial = self._regressor(self.cfg) if callable(self._regressor) else self._regressor
        bias = 0.0 if initial is None else math.atanh(2 * (initial - lo) / (hi - lo) - 1)
        return int(self.op_code), int(self.get_num_filter_parameters()), kind, float(lo), float(hi - lo), float(bias)

    # -- pixels ----------------------------------------------------------------------------------
    def process(self, img, param):
        """Whole-image filter, no mask, no clip: one HIP launch (adaisp_process / adaisp_forward)."""
        if self.op_code is None:
            raise NotImplementedError("process not implement")
        return isp_apply(img, param, self.op_code, clip=False)

    def _process_clipped(self, img, param):
        # lerp(img, process(img, p), mask) with mask == 1 is process(img, p); Filter.forward then clips to [0,1]
        return isp_apply(img, param, self.op_code, clip=True)

    def get_mask(self, img, mask_parameters=None):
        if self.use_masking():
            raise NotImplementedError("spatial masks are disabled in the reference (cfg.masking=False, "
                                      "isp/filters.py:161-162) and not built here")
        return torch.ones((1, 1, 1, 1), dtype=torch.float32, device=img.device)

    def forward(self, img, img_features=None, specified_parameter=None, high_res=None):
        if self.predict:
            assert (img_features is None) ^ (specified_parameter is None)
        if img_features is not None:
            filter_features, mask_parameters = self.extract_parameters(img_features)
            filter_parameters = self.filter_param_regressor(filter_features)
        else:
            assert not self.use_masking()
            filter_parameters = specified_parameter
            mask_parameters = torch.zeros(1, self.get_num_mask_parameters(), dtype=torch.float32)
        debug_info = {
            "filter_parameters": filter_parameters if self.debug_info_batched() else filter_parameters[0]}
        self.mask_parameters = mask_parameters
        self.mask = self.get_mask(img, mask_parameters)
        debug_info["mask"] = self.mask[0]
        low_res_output = self._process_clipped(img, filter_parameters)
        high_res_output = None
        if high_res is not None:
            if self.no_high_res():
                high_res_output = high_res
            else:
                self.high_res_mask = self.get_mask(high_res, mask_parameters)
                high_res_output = self._process_clipped(high_res, filter_parameters)
        return low_res_output, high_res_output, debug_info

    def run(self, img, param):
        self.mask = self.get_mask(img)
        return self.process(img, param)

    def run_v2(self, img, param):
        self.mask = self.get_mask(img)
        return self.process(img, param[None, :])

    def predict_param(self, img, img_features):
        filter_features, _ = self.extract_parameters(img_features)
        self.mask = self.get_mask(img)
        return self.process(img, self.filter_param_regressor(filter_features))

    # -- drawing (needs OpenCV, which this image does not ship; out of the hot path) -------------------
    def visualize_filter(self, debug_info, canvas):
        raise NotImplementedError("visualisation needs cv2 and is outside the ISP hot path")

    def visualize_mask(self, debug_info, res):
        raise NotImplementedError("visualisation needs cv2 and is outside the ISP hot path")


class ExposureFilter(Filter):
    op_code = _lib.OP_EXPOSURE
    _regressor = staticmethod(lambda cfg: (0, -cfg.exposure_range, cfg.exposure_range, 0))

    def __init__(self, cfg, predict=False):
        super().__init__(cfg, "E", 1, predict)

    def filter_param_regressor(self, features):
        r = self.cfg.exposure_range
        return tanh_range(-r, r, initial=0)(features)


class GammaFilter(Filter):
    op_code = _lib.OP_GAMMA
    _regressor = staticmethod(lambda cfg: (1, -math.log(cfg.gamma_range), math.log(cfg.gamma_range), None))

    def __init__(self, cfg, predict=False):
        super().__init__(cfg, "G", 1, predict)

    def filter_param_regressor(self, features):
        lg = float(math.log(self.cfg.gamma_range))
        return torch.exp(tanh_range(-lg, lg)(features))


class ImprovedWhiteBalanceFilter(Filter):
    op_code = _lib.OP_WB
    _regressor = (4, -0.5, 0.5, None)

    def __init__(self, cfg, predict=False):
        super().__init__(cfg, "W", 3, predict)
        self.num_filter_parameters = self.channels
        # R gain pinned to 1; non-persistent so the state dict keeps the reference's keys (and no H2D copy per call)
        self.register_buffer("_keep", torch.tensor([[0.0, 1.0, 1.0]], dtype=torch.float32), persistent=False)

    def filter_param_regressor(self, features):
        log_wb_range = 0.5
        scaling = torch.exp(tanh_range(-log_wb_range, log_wb_range)(features * self._keep.to(features.device)))
        lum = 1e-5 + 0.27 * scaling[:, 0] + 0.67 * scaling[:, 1] + 0.06 * scaling[:, 2]
        return scaling * (1.0 / lum)[:, None]


class ColorFilter(Filter):
    op_code = _lib.OP_COLOR
    _regressor = staticmethod(lambda cfg: (0, cfg.color_curve_range[0], cfg.color_curve_range[1], 1))

    def __init__(self, cfg, predict=False):
        super().__init__(cfg, "C", 3 * cfg.curve_steps, predict)
        self.curve_steps = cfg.curve_steps

    def filter_param_regressor(self, features):
        curve = torch.reshape(features, (-1, self.cfg.curve_steps, self.channels))[:, :, :, None, None]
        return tanh_range(*self.cfg.color_curve_range, initial=1)(curve)


class ToneFilter(Filter):
    op_code = _lib.OP_TONE
    _regressor = staticmethod(lambda cfg: (0, cfg.tone_curve_range[0], cfg.tone_curve_range[1], None))

    def __init__(self, cfg, predict=False):
        super().__init__(cfg, "T", cfg.curve_steps, predict)
        self.curve_steps = cfg.curve_steps
        if cfg.curve_steps != 8:
            raise ValueError("the tone/colour kernels are built for cfg.curve_steps == 8")

    def filter_param_regressor(self, features):
        curve = torch.reshape(features, (-1, self.cfg.curve_steps, 1))[:, :, :, None, None]
        return tanh_range(*self.cfg.tone_curve_range)(curve)


class ToneFilterV2(ToneFilter):
    """Same curve; `process` takes the flat [B,8] parameter layout (reference isp/filters.py:365-387)."""


class ContrastFilter(Filter):
    op_code = _lib.OP_CONTRAST
    _regressor = (3, 0.0, 1.0, None)

    def __init__(self, cfg, predict=False):
        super().__init__(cfg, "Ct", 1, predict)

    def filter_param_regressor(self, features):
        return torch.tanh(features)


class WNBFilter(Filter):
    op_code = _lib.OP_WNB
    _regressor = (2, 0.0, 1.0, None)

    def __init__(self, cfg, predict=False):
        super().__init__(cfg, "BW", 1, predict)

    def filter_param_regressor(self, features):
        return torch.sigmoid(features)


class SaturationPlusFilter(Filter):
    op_code = _lib.OP_SATPLUS
    _regressor = (2, 0.0, 1.0, None)

    def __init__(self, cfg, predict=False):
        super().__init__(cfg, "S+", 1, predict)

    def filter_param_regressor(self, features):
        return torch.sigmoid(features)


class DenoiseFilter(Filter):
    """Non-local means, gray weights, 11x11 search / 5x5 patch (reference isp/filters.py:571-586)."""
    op_code = _lib.OP_NLM
    _regressor = (2, 0.0, 1.0, None)

    def __init__(self, cfg, predict=False):
        super().__init__(cfg, "NLM", 1, predict)
        from .denoise import NonLocalMeansGray
        self.denoise = NonLocalMeansGray(search_window_size=11, patch_size=5)

    def filter_param_regressor(self, features):
        return torch.sigmoid(features)


class SharpenUSMFilter(Filter):
    op_code = _lib.OP_USM
    _regressor = staticmethod(lambda cfg: (0, cfg.usm_sharpen_range[0], cfg.usm_sharpen_range[1], None))

    def __init__(self, cfg, predict=False):
        super().__init__(cfg, "USM", 2, predict)

    def filter_param_regressor(self, features):
        return tanh_range(*self.cfg.usm_sharpen_range)(features)


class SharpenFilter(Filter):
    op_code = _lib.OP_SHARPEN
    _regressor = staticmethod(lambda cfg: (0, cfg.sharpen_range[0], cfg.sharpen_range[1], None))

    def __init__(self, cfg, predict=False):
        super().__init__(cfg, "Shr", 1, predict)

    def filter_param_regressor(self, features):
        return tanh_range(*self.cfg.sharpen_range)(features)


class SharpenFilterV2(SharpenFilter):
    op_code = _lib.OP_SHARPEN_V2


def _fused_only(name):
    def fn(*_a, **_k):
        raise NotImplementedError(f"{name} exists only fused inside a HIP kernel here (ops SATPLUS / CCM); "
                                  "no standalone tensor version is shipped")
    fn.__name__ = name
    return fn


rgb2hsv = _fused_only("rgb2hsv")
hsv2rgb = _fused_only("hsv2rgb")
color_correction_matrix = _fused_only("color_correction_matrix")


class CCMFilter(Filter):
    """3x3 colour-correction matrix; the kernel divides every row by its sum (reference isp/filters.py:694-708)."""
    op_code = _lib.OP_CCM
    _regressor = staticmethod(lambda cfg: (0, cfg.ccm_range[0], cfg.ccm_range[1], None))

    def __init__(self, cfg, predict=False):
        super().__init__(cfg, "CCM", 9, predict)

    def filter_param_regressor(self, features):
        return tanh_range(*self.cfg.ccm_range)(features)
