"""Default configuration of the hot path: every key of the reference's config.py:5-87 with the same
values (filter order, parameter ranges, RL and network hyper-parameters). `cfg.filters` holds filter
CLASSES; position in the list is the policy's action id."""
from .isp.filters import (CCMFilter, ContrastFilter, DenoiseFilter, ExposureFilter, GammaFilter,
                          ImprovedWhiteBalanceFilter, SaturationPlusFilter, SharpenFilter, ToneFilter, WNBFilter)
from .util import Dict

cfg = Dict()

# logging / checkpoint cadence
cfg.val_freq = 1000
cfg.save_model_freq = 1000
cfg.print_freq = 100
cfg.summary_freq = 100
cfg.show_img_num = 2

cfg.parameter_lr_mul = 1
cfg.value_lr_mul = 1
cfg.critic_lr_mul = 1

# ---- filters ------------------------------------------------------------------------------------
cfg.filters = [ExposureFilter, GammaFilter, CCMFilter, SharpenFilter, DenoiseFilter,
               ToneFilter, ContrastFilter, SaturationPlusFilter, WNBFilter, ImprovedWhiteBalanceFilter]
cfg.filter_runtime_penalty = False
cfg.filters_runtime = [1.7, 2.0, 1.9, 6.3, 10, 2.7, 2.1, 2.0, 1.9, 1.7]
cfg.filter_runtime_penalty_lambda = 0.01

cfg.curve_steps = 8
cfg.gamma_range = 3
cfg.exposure_range = 3.5
cfg.wb_range = 1.1
cfg.color_curve_range = (0.90, 1.10)
cfg.lab_curve_range = (0.90, 1.10)
cfg.tone_curve_range = (0.5, 2)
cfg.usm_sharpen_range = (0.0, 2.0)
cfg.sharpen_range = (0.0, 10.0)
cfg.ccm_range = (-2.0, 2.0)
cfg.denoise_range = (0.0, 1.0)

cfg.masking = False
cfg.minimum_strength = 0.3
cfg.maximum_sharpness = 1
cfg.clamp = False

# ---- RL -----------------------------------------------------------------------------------------
cfg.critic_logit_multiplier = 100
cfg.discount_factor = 1.0
cfg.filter_usage_penalty = 1.0
cfg.use_TD = True
cfg.replay_memory_size = 128
cfg.maximum_trajectory_length = 7
cfg.over_length_keep_prob = 0.5
cfg.all_reward = 1.0
cfg.img_include_states = True
cfg.exploration = 0.05
cfg.exploration_penalty = 0.05
cfg.early_stop_penalty = 1.0
cfg.detect_loss_weight = 1.0

# ---- policy / critic networks -------------------------------------------------------------------
cfg.base_channels = 32
cfg.dropout_keep_prob = 0.5
cfg.shared_feature_extractor = True
cfg.fc1_size = 128
cfg.bnw = False
cfg.feature_extractor_dims = 4096
cfg.use_penalty = True
cfg.z_type = 'uniform'
cfg.z_dim_per_filter = 16

cfg.num_state_dim = 3 + len(cfg.filters)
cfg.z_dim = 3 + len(cfg.filters) * cfg.z_dim_per_filter
cfg.test_steps = 5
