/*
 * adaisp.h — C-ABI of the MI355X-native AdaptiveISP filter stack (libadaisp.so).
 *
 * This is the drop-in boundary for the ISP hot path. The reference has no FFI of its own: its
 * boundary is the Python class API (`Filter.process(img, param)`, `Filter.forward`,
 * `Agent.forward`). Each entry point below cites the reference call it replaces
 * (paths relative to the reference checkout).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller, fp32, contiguous;
 *   - images are planar CHW:  img[b][c][y][x],  c in {R,G,B},  shape [B,3,H,W];
 *   - params are the REGRESSED filter parameters (after tanh_range / sigmoid / exp),
 *     one row of `param_stride` floats per image;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *     no call allocates, synchronises, or keeps global mutable state — all are re-entrant;
 *   - return 0 on success, a negative ADAISP_E* code otherwise (adaisp_strerror() names it).
 */
#ifndef ADAISP_H_
#define ADAISP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADAISP_ABI_VERSION 8

/* Kernel op codes. 0..9 follow the reference's default filter order (config.py:19-22). */
enum adaisp_op {
    ADAISP_OP_ZERO       = -1, /* all-zero one-hot: reference pdf_sample u==0 edge, agent.py:12-16,154 */
    ADAISP_OP_EXPOSURE   = 0,  /* ExposureFilter.process               isp/filters.py:223-224  n=1  */
    ADAISP_OP_GAMMA      = 1,  /* GammaFilter.process                  isp/filters.py:244-245  n=1  */
    ADAISP_OP_CCM        = 2,  /* CCMFilter.process                    isp/filters.py:703-708  n=9  */
    ADAISP_OP_SHARPEN    = 3,  /* SharpenFilter / adjust_sharpness     isp/sharpen.py:105-142  n=1  */
    ADAISP_OP_NLM        = 4,  /* DenoiseFilter / NonLocalMeansGray    isp/denoise.py:93-119   n=1  */
    ADAISP_OP_TONE       = 5,  /* ToneFilter.process                   isp/filters.py:337-347  n=8  */
    ADAISP_OP_CONTRAST   = 6,  /* ContrastFilter.process               isp/filters.py:415-419  n=1  */
    ADAISP_OP_SATPLUS    = 7,  /* SaturationPlusFilter.process         isp/filters.py:546-560  n=1  */
    ADAISP_OP_WNB        = 8,  /* WNBFilter.process                    isp/filters.py:435-437  n=1  */
    ADAISP_OP_WB         = 9,  /* ImprovedWhiteBalanceFilter.process   isp/filters.py:271-272  n=3  */
    ADAISP_OP_USM        = 10, /* SharpenUSMFilter / unsharp_mask      isp/sharpen.py:84-102   n=2  */
    ADAISP_OP_SHARPEN_V2 = 11, /* SharpenFilterV2 / sharpness          isp/sharpen.py:145-182  n=1  */
    ADAISP_OP_COLOR      = 12, /* ColorFilter.process                  isp/filters.py:293-303  n=24 */
    ADAISP_OP_COUNT      = 13
};

#define ADAISP_MAX_PARAMS 24

/* flags */
#define ADAISP_CLIP01 1u /* clamp the result to [0,1]: Filter.forward's final clip, isp/filters.py:125 */
#define ADAISP_NLM_EXACT 2u /* NLM: add the 25 patch terms in the reference's single running-sum order
                               (isp/denoise.py:60-63) instead of the default 5x5 separable association; ~3.5x slower */
#define ADAISP_NLM_SEP_V1 4u /* NLM: the compiler-scheduled form of the separable kernel (measurement / cross-check) */
#define ADAISP_NO_USM 8u /* adaisp_forward: no image of the batch selects ADAISP_OP_USM (saves its empty launch) */
#define ADAISP_NLM_TILE32 16u /* NLM: the 32-row tile of the default kernel (2 workgroups per CU) instead of the 24-row
                                 one (3 per CU); same arithmetic, same results (measurement / cross-check) */

/* error codes */
#define ADAISP_OK          0
#define ADAISP_EINVAL     -1 /* null pointer / non-positive size / bad stride  */
#define ADAISP_EOP        -2 /* unknown op code (host-known op only)           */
#define ADAISP_EALIAS     -3 /* out aliases img for a stencil op               */
#define ADAISP_ESHAPE     -4 /* shape unsupported by the op (e.g. 3x3 on H<3)  */
#define ADAISP_ELAUNCH    -5 /* hipLaunchKernel failed                         */

/*
 * One RL step of the ISP: image b is filtered by op filter_id[b] with params[b].
 * Replaces the reference's "run all filters, stack, one-hot select" of Agent.forward
 * (agent.py:103-116,154) — only the selected filter is computed.
 * `filter_id` lives on the device (no host sync is needed to pick the work); -1 — and any id outside enum adaisp_op —
 * writes zeros (the reference's all-zero one-hot row).
 * If `pooled64_next` != NULL it receives AdaptiveAvgPool2d((64,64)) of `out`
 * ([B,3,64,64]; agent.py:97 / value.py:63) for the next step's policy input — from the SAME launch for the pointwise
 * and 3x3 / 5x5 stencil ops (the kernels are cut along the pool windows; rows 16-byte aligned, H, W >= 64, pool columns
 * <= 64 px wide), bit-identical to adaisp_pool64(out); NLM images and other geometries take a pooling launch.
 * `out` must not alias `img`.
 */
int adaisp_forward(const float* img, float* out, float* pooled64_next,
                   const int32_t* filter_id, const float* params, int param_stride,
                   int B, int H, int W, unsigned flags, void* stream);

/*
 * adaisp_forward when the HOST knows the op of the step (teacher-forced schedules, `selected_filter_id` of
 * Agent.forward, agent.py:88,150-153): one op for the whole batch, per-image params, optional fused pooling. Exactly one
 * kernel is enqueued (two for NLM with pooling) instead of one per kernel family. Same results as adaisp_forward with
 * filter_id[b] == op.
 */
int adaisp_forward_uniform(int op, const float* img, float* out, float* pooled64_next,
                           const float* params, int param_stride,
                           int B, int H, int W, unsigned flags, void* stream);

/*
 * Same arithmetic with ONE host-known op for the whole batch: Filter.process(img, param)
 * (flags = 0) and the image part of Filter.forward (flags = ADAISP_CLIP01), isp/filters.py:81,115-125.
 * Pointwise ops may run in place (out == img).
 */
int adaisp_process(int op, const float* img, float* out,
                   const float* params, int param_stride,
                   int B, int H, int W, unsigned flags, void* stream);

/*
 * Parameter gradients of adaisp_forward: grad_params[b][k] = sum_px grad_out * d out / d params[b][k]
 * through the selected filter and (if ADAISP_CLIP01) the clip. This is the only gradient the
 * reference's training needs (train.py:341-342: imgs is a constant leaf).
 * grad_params ([B,param_stride]) is zero-filled by the callee.
 */
int adaisp_backward_params(const float* img, const float* grad_out,
                           const int32_t* filter_id, const float* params, int param_stride,
                           float* grad_params,
                           int B, int H, int W, unsigned flags, void* stream);

/*
 * Bayer demosaic front-end — an EXTENSION (north_star's raw-Bayer entry; the reference has no demosaic, only the
 * inverse packing `mosaic` / `reconstruct_bayer`, isp/unprocess_np.py:82-128).
 * raw: uint16 [B,H,W] colour-filter-array samples; pattern = 2*ry + rx, the position of the RED sample in the 2x2
 * cell (ADAISP_CFA_RGGB 0, GRBG 1, GBRG 2, BGGR 3). out: planar fp32 [B,3,H,W], bilinear interpolation of
 * (raw - black_level) / (white_level - black_level); mirrored borders. H and W even.
 */
#define ADAISP_CFA_RGGB 0
#define ADAISP_CFA_GRBG 1
#define ADAISP_CFA_GBRG 2
#define ADAISP_CFA_BGGR 3
int adaisp_demosaic(const uint16_t* raw, float* out, int B, int H, int W, int pattern,
                    float black_level, float white_level, void* stream);

/*
 * NonLocalMeansGray(search_window_size, patch_size).forward(rgb, h) for ANY odd sizes — isp/denoise.py:93-119 (class default
 * 21 / 7; the ISP's DenoiseFilter constructs 11 / 5, isp/filters.py:577, which ADAISP_OP_NLM serves with the tuned kernel).
 * Luminance 0.299 R + 0.587 G + 0.114 B of the CLIPPED image (rgb_to_luminance :11-17), patch distance = box sum of squared
 * differences under torch.roll's circular wrap, weight exp(-sqrt(relu(D)) / (relu(h) + 1e-8)), colours = the input as given,
 * result clamped to [0, 1]. h: one value per image, `h_stride` floats apart. `workspace` (adaisp_nlm_general_workspace_bytes:
 * one fp32 luminance plane per image) is caller-owned scratch. A plain gather kernel — O(search^2 patch^2) per pixel — for
 * configurations the ISP path does not use. `out` must not overlap `img`.
 */
size_t adaisp_nlm_general_workspace_bytes(int B, int H, int W);
int adaisp_nlm_general(const float* img, float* out, const float* h, int h_stride, void* workspace, size_t workspace_bytes,
                       int B, int H, int W, int search_window_size, int patch_size, void* stream);

/* AdaptiveAvgPool2d((64,64)) of a [B,3,H,W] image: agent.py:85,97, value.py:61,63. */
int adaisp_pool64(const float* img, float* pooled, int B, int H, int W, void* stream);

/*
 * Backward of adaisp_pool64: grad_pooled [B,3,64,64] -> grad_img [B,3,H,W] (every pixel written). The critic sees the
 * retouched image through this pooling (value.py:61-63) and, with cfg.use_TD, the agent loss back-propagates through
 * V(retouch, new_states) into the filter parameters (train.py:281-305): this gradient enters adaisp_backward_params
 * as part of grad_out.
 */
int adaisp_pool64_backward(const float* grad_pooled, float* grad_img, int B, int H, int W, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Fused eval path of one policy step (Agent.forward in eval mode, agent.py:88-285; FeatureExtractor
 * agent.py:26-60; Filter.extract_parameters / filter_param_regressor isp/filters.py:65-73 and per class).
 * Weights are the module parameters with BatchNorm folded into the convs (eval statistics), fp32.
 * --------------------------------------------------------------------------------------------------------- */
#define ADAISP_POLICY_MAX_FILTERS 16

enum adaisp_regressor_kind {
    ADAISP_REG_TANH_RANGE     = 0, /* tanh01(x + bias) * scale + lo          (E, CCM, Shr, T, USM, C) */
    ADAISP_REG_EXP_TANH_RANGE = 1, /* exp(tanh01(x + bias) * scale + lo)     (G)                      */
    ADAISP_REG_SIGMOID        = 2, /* 1 / (1 + exp(-x))                      (NLM, S+, BW)            */
    ADAISP_REG_TANH           = 3, /* tanh(x)                                (Ct)                     */
    ADAISP_REG_WB             = 4  /* exp(tanh_range(x*[0,1,1])) / luminance (W), isp/filters.py:258-269 */
};

typedef struct adaisp_regressor {
    int32_t op;      /* kernel op code of the filter (enum adaisp_op)  */
    int32_t n;       /* number of regressed parameters                 */
    int32_t kind;    /* enum adaisp_regressor_kind                     */
    float lo, scale, bias;
} adaisp_regressor;

typedef struct adaisp_policy_finish_args {
    /* inputs */
    const float* hidden;       /* [B][F+1][hid]  LeakyReLU(fc1) of the F filter heads, then the selector's   */
    const float* w_filter;     /* [rows][hid]    fc_filter weights of all heads, stacked in filter order       */
    const float* b_filter;     /* [rows]                                                                        */
    const int32_t* row_filter; /* [rows]         filter index of each stacked row                               */
    const int32_t* row_slot;   /* [rows]         parameter slot of each stacked row inside its filter           */
    const float* w_sel;        /* [F][hid]       selector fc2                                                   */
    const float* b_sel;        /* [F]                                                                           */
    const float* noise;        /* [B][noise_stride]  z; column 0 is the selection noise (agent.py:92)           */
    const float* states;       /* [B][3+F]                                                                      */
    const float* runtime;      /* [F] or NULL (cfg.filter_runtime_penalty off)                                  */
    /* outputs */
    float* params_all;         /* [B][F][param_width] regressed parameters of every filter                      */
    float* packed;             /* [B][param_width]    row of the selected filter (input of adaisp_forward)      */
    int32_t* op_ids;           /* [B]                 op code of the selected filter (-1: all-zero one-hot)     */
    long long* selected;       /* [B]                 selected filter id (int64 like the reference)             */
    float* pdf_out;            /* [B][F]                                                                        */
    float* surrogate;          /* [B]                                                                           */
    float* new_states;         /* [B][3+F]                                                                      */
    float* penalty;            /* [B]                                                                           */
    /* scalars */
    int32_t num_filters, num_rows, hid, param_width, noise_stride;
    int32_t train_mode;        /* 1: pdf_sample, 0: argmax                                                      */
    int32_t forced_id;         /* >= 0: teacher-forced selected_filter_id                                       */
    float one_minus_exploration, exploration_over_f;
    float entropy_coef;        /* (1 - progress) * cfg.exploration_penalty                                      */
    float log_num_filters, test_steps, filter_usage_penalty, early_stop_penalty, runtime_lambda;
    adaisp_regressor reg[ADAISP_POLICY_MAX_FILTERS];
} adaisp_policy_finish_args;

/*
 * One folded trunk layer: out = LeakyReLU_0.2(conv2d(in, w, k4 s2 p1) + bias), G independent trunks per launch.
 *   in  [G][B][Cin][Hin][Hin], w [G][Cout][Cin][4][4], bias [G][Cout], out [G][B][Cout][Hin/2][Hin/2].
 * First layer: pass states != NULL; `in` is then the 64x64-pooled image [B,3,Hin,Hin] shared by all trunks and
 * input channels 3..Cin-1 are the per-image constants states[b][c-3] (enrich_image_input, util.py:58-63).
 * Cout must be a multiple of 8.
 */
int adaisp_policy_conv(const float* in, const float* states, int n_state, const float* w, const float* bias,
                       float* out, int G, int B, int Cin, int Hin, int Cout, void* stream);

/* hidden[b][h][j] = LeakyReLU_0.2(b1[h][j] + feats[head_src[h]][b][:] . w1[h][j][:]);  feats [G][B][D], D % 4 == 0 */
int adaisp_policy_fc1(const float* feats, const int32_t* head_src, const float* w1, const float* b1, float* hidden,
                      int B, int D, int NH, int HID, void* stream);

/* Everything after the hidden layers (see struct). `args` is a HOST struct holding DEVICE pointers. */
int adaisp_policy_finish(const adaisp_policy_finish_args* args, int B, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Training-mode trunk of the policy / critic networks (FeatureExtractor, agent.py:26-60 / value.py:6-44, in train
 * mode as train.py:258,282-283 runs them): [Conv2d(k4 s2 p1) -> BatchNorm2d(batch statistics) -> LeakyReLU(0.2)] x 4
 * on a 64x64 input, forward and backward, fp32. One call serves G <= 2 trunk INSTANCES:
 *   the agent's two trunks (feature_extractor + action_selection: same input, two parameter sets), or
 *   the critic's two calls of an iteration (V(imgs, states), V(retouch, new_states): two inputs, ONE parameter set —
 *   pass the same `adaisp_trunk_params` twice and share_params = 1: statistics stay per instance, the running
 *   statistics are updated in instance order and the parameter gradients are the sum over instances, in that order).
 * Layer-1 input of instance g: the 3 planes img[g] [B,3,64,64] followed by svec[g] [B,n_state] as constant planes
 * (enrich_image_input, util.py:58-63; for the critic the state vector with its three hand statistics, value.py:65-80).
 * Channels: C[0] = 3 + n_state, C[1..4] the four conv widths (C[1..4] % 16 == 0). Output feat [G][B][C[4]*16] =
 * the last activation flattened channel-major (agent.py:57 reshape). Every reduction (batch statistics, their
 * backward sums, weight gradients) runs in a fixed order: results are bit-reproducible run to run.
 * `workspace` keeps what backward needs (pre-BatchNorm outputs, activations, mean / rstd); `scratch` holds backward
 * temporaries. Sizes: adaisp_trunk_train_workspace_bytes / _scratch_bytes.
 * --------------------------------------------------------------------------------------------------------- */
#define ADAISP_TRUNK_MAX_G 2
#define ADAISP_TRUNK_LAYERS 4

typedef struct adaisp_trunk_params {
    const float* w[ADAISP_TRUNK_LAYERS];        /* [C[l+1]][C[l]][4][4]                              */
    const float* bias[ADAISP_TRUNK_LAYERS];     /* [C[l+1]]                                          */
    const float* gamma[ADAISP_TRUNK_LAYERS];    /* BatchNorm weight                                  */
    const float* beta[ADAISP_TRUNK_LAYERS];     /* BatchNorm bias                                    */
    float* running_mean[ADAISP_TRUNK_LAYERS];   /* updated in place by the forward (NULL: left alone) */
    float* running_var[ADAISP_TRUNK_LAYERS];
} adaisp_trunk_params;

typedef struct adaisp_trunk_grads {             /* every tensor written (not accumulated into)       */
    float* w[ADAISP_TRUNK_LAYERS];
    float* bias[ADAISP_TRUNK_LAYERS];
    float* gamma[ADAISP_TRUNK_LAYERS];
    float* beta[ADAISP_TRUNK_LAYERS];
} adaisp_trunk_grads;

typedef struct adaisp_trunk_args {
    int32_t G, B, n_state, share_params;
    int32_t C[ADAISP_TRUNK_LAYERS + 1];
    float momentum, eps, slope;                 /* BatchNorm momentum / eps, LeakyReLU slope          */
    const float* img[ADAISP_TRUNK_MAX_G];
    const float* svec[ADAISP_TRUNK_MAX_G];
    adaisp_trunk_params p[ADAISP_TRUNK_MAX_G];
    float* feat;                                /* forward output [G][B][C[4]*16]                     */
    float* workspace;
    size_t workspace_bytes;
    /* backward only */
    const float* dfeat;                         /* [G][B][C[4]*16]                                    */
    adaisp_trunk_grads g[ADAISP_TRUNK_MAX_G];   /* share_params: g[0] only                            */
    float* dimg[ADAISP_TRUNK_MAX_G];            /* [B,3,64,64] or NULL (input gradient not wanted)    */
    float* dsvec[ADAISP_TRUNK_MAX_G];           /* [B,n_state] or NULL (both or neither per instance) */
    float* scratch;
    size_t scratch_bytes;
} adaisp_trunk_args;

size_t adaisp_trunk_train_workspace_bytes(const adaisp_trunk_args* args);
size_t adaisp_trunk_train_scratch_bytes(const adaisp_trunk_args* args);
/* `args` is a HOST struct holding DEVICE pointers. 8 launches. */
int adaisp_trunk_train_fwd(const adaisp_trunk_args* args, void* stream);
/* After adaisp_trunk_train_fwd with the same args / workspace. 11-15 launches. */
int adaisp_trunk_train_bwd(const adaisp_trunk_args* args, void* stream);

/*
 * The critic's hand statistics (value.py:65-80) of G <= 2 batches of 64x64 planes `small` [B,3,64,64]:
 *   svec[b] = [states[b][0..n_state), mean luminance, luminance variance (unbiased, torch.var), mean saturation]
 * i.e. the state vector the critic's trunk reads as constant planes. Backward: dsmall = dsmall_in (or 0) + the gradient
 * of the three statistics given dsvec [B,n_state+3] (torch's rules: clip passes on the closed interval, max / min over
 * channels give the first index on ties, minimum splits a tie); dsmall may alias dsmall_in; instances with
 * dsmall[g] == NULL are skipped. One launch each.
 */
typedef struct adaisp_critic_planes_args {
    int32_t G, B, n_state;
    const float* small[ADAISP_TRUNK_MAX_G];
    const float* states[ADAISP_TRUNK_MAX_G];      /* [B,n_state] (NULL when n_state == 0) */
    float* svec[ADAISP_TRUNK_MAX_G];              /* [B,n_state+3]                        */
    /* backward only */
    const float* dsvec[ADAISP_TRUNK_MAX_G];
    const float* dsmall_in[ADAISP_TRUNK_MAX_G];
    float* dsmall[ADAISP_TRUNK_MAX_G];
} adaisp_critic_planes_args;
int adaisp_critic_planes_fwd(const adaisp_critic_planes_args* args, void* stream);
int adaisp_critic_planes_bwd(const adaisp_critic_planes_args* args, void* stream);

/*
 * Reward / TD target / losses of one RL iteration (train.py:262-305): per-sample vectors of B floats in, the per-sample
 * reward / q_value / advantage and losses = [value_loss, agent_loss] out; backward from dlosses [2] to the five inputs
 * that carry gradients. The same operation order as the element-wise reference arithmetic. One launch each.
 */
typedef struct adaisp_td_args {
    int32_t B, state_dim;                          /* new_states [B,state_dim]: [., stopped, step, usage...] */
    int32_t use_penalty, use_truncated, use_td;
    float detect_loss_weight, all_reward, critic_logit_multiplier, discount_factor, parameter_lr_mul,
          maximum_trajectory_length, max_bri;
    const float *l_in, *l_re, *penalty, *surrogate, *new_states, *old_value, *new_value, *retouch_mean;
    float *reward, *q_value, *advantage, *losses;
    /* backward only */
    const float* dlosses;
    float *d_l_re, *d_penalty, *d_surrogate, *d_old_value, *d_new_value;
} adaisp_td_args;
int adaisp_td_fwd(const adaisp_td_args* args, void* stream);
int adaisp_td_bwd(const adaisp_td_args* args, void* stream);

/*
 * The policy's tail in TRAINING mode (agent.py:103-149, 234-280): from the heads' pre-activations x [B][F][param_width]
 * (fc_filter outputs, zero-padded slots) and the selector's logits [B][F] to everything Agent.forward hands on — the regressed
 * parameter table, pdf, sampled / forced selection, surrogate, packed row + op code for adaisp_forward, state update, penalty —
 * with the arithmetic of adaisp_policy_finish; and its backward: d_x (non-zero on the selected filter's row only: nothing else
 * reaches the pixels) and d_logits from d_packed [B][param_width], d_surrogate [B], d_penalty [B] (any may be NULL = zero).
 * One launch each.
 */
typedef struct adaisp_policy_tail_args {
    int32_t B, num_filters, param_width, noise_stride;
    int32_t sample;            /* 1: pdf_sample with noise[b][0] (training), 0: argmax                 */
    int32_t forced_id;         /* >= 0: teacher-forced selected_filter_id                              */
    float one_minus_exploration, exploration_over_f, entropy_coef, log_num_filters, test_steps, filter_usage_penalty,
          early_stop_penalty, runtime_lambda;
    adaisp_regressor reg[ADAISP_POLICY_MAX_FILTERS];
    const float *x, *logits, *noise, *states, *runtime;       /* runtime [F] or NULL                   */
    float* table;              /* [B][F][param_width] regressed parameters of every filter              */
    float* packed;             /* [B][param_width]                                                      */
    int32_t* op_ids;           /* [B]                                                                   */
    long long* selected;       /* [B]  (read by the backward)                                           */
    float* pdf;                /* [B][F] (read by the backward)                                         */
    float *surrogate, *new_states, *penalty;                  /* [B], [B][3+F], [B]                    */
    /* backward only */
    const float *d_packed, *d_surrogate, *d_penalty;
    float *d_x, *d_logits;
    /* non-NULL: the entropy coefficient is read from this DEVICE float at run time instead of `entropy_coef` (a launch captured
     * in a hipGraph is replayed with the coefficient of its iteration: train.py:258 feeds `progress` anew every iteration) */
    const float* entropy_coef_dev;
} adaisp_policy_tail_args;
int adaisp_policy_tail_fwd(const adaisp_policy_tail_args* args, void* stream);
int adaisp_policy_tail_bwd(const adaisp_policy_tail_args* args, void* stream);

/*
 * stats[b] = (mean, number of non-finite values) of image b of a batch of B images of n floats each — what the TD target's
 * brightness test (train.py:287-291) and the replay guard (train.py:374-381) read, in one pass over the batch. `workspace`:
 * B * 128 floats. Fixed summation order. Two launches.
 */
int adaisp_image_stats(const float* img, float* stats, float* workspace, int B, long n, void* stream);

/*
 * torch.nn.utils.clip_grad_norm_(params, max_norm) followed by torch.optim.Adam.step() (train.py:341-351; amsgrad off, no weight
 * decay) over a table of fp32 tensors in three launches. `table` is a DEVICE array of `ntensors` entries: parameter, gradient,
 * first / second moment (updated in place), the tensor's step count AFTER this step (a device float, as torch's fused Adam keeps
 * it), element count, and `chunk0` = the number of 4096-element chunks of the tensors before it (ascending); `nchunks` their total.
 * `workspace`: nchunks + 2 floats; afterwards workspace[nchunks] = the clip coefficient, workspace[nchunks + 1] = the total norm.
 * max_norm <= 0: no clipping. Fixed summation order; the update arithmetic of torch's fused Adam.
 */
typedef struct adaisp_adam_tensor {
    float* p;
    const float* g;
    float* m;
    float* v;
    const float* step;
    long n, chunk0;
} adaisp_adam_tensor;
int adaisp_clip_adam_step(const adaisp_adam_tensor* table, int ntensors, long nchunks, float* workspace, float max_norm, double lr,
                          double beta1, double beta2, double eps, void* stream);
/* The same with the learning rate read from a DEVICE double at run time (`lr_dev`): the launches of a captured iteration follow
 * the LambdaLR schedule (train.py:206-218, 350-351) without a new kernel argument. Same arithmetic, same three launches. */
int adaisp_clip_adam_step_dev(const adaisp_adam_tensor* table, int ntensors, long nchunks, float* workspace, float max_norm,
                              const double* lr_dev, double beta1, double beta2, double eps, void* stream);

/*
 * The policy's parameter heads in TRAINING mode, forward and backward on the filters' own parameter tensors (agent.py:103-116: per
 * filter `fc1` D -> hid, LeakyReLU 0.2, `fc_filter` hid -> n_f; agent.py:117-121: the selector's `fc1` D -> hid on ITS trunk's features,
 * LeakyReLU, `fc2` hid -> F). Forward (2 launches): hidden [B][F+1][hid] = the pre-activations of every fc1 (group F = the selector),
 * x [B][F][pw] = every fc_filter's output, zero in the slots >= n_f (what adaisp_policy_tail_fwd reads), logits [B][F]. Backward
 * (4 launches) from dx / dlogits: the gradient of every weight and bias into the caller's tensors (written, not accumulated) and of
 * both feature rows; `dhid` [B][F+1][hid] and `part` [(F+1)][B][D] are scratch. Every sum in a fixed order. B <= 8, hid a multiple of
 * 8, D a multiple of 1024.
 */
#define ADAISP_HEADS_MAX_B 8
typedef struct adaisp_heads_args {
    int32_t B, F, D, hid, pw;
    int32_t n[ADAISP_POLICY_MAX_FILTERS];
    const float *feat_f, *feat_s;                              /* [B][D] each                                     */
    const float *w1[ADAISP_POLICY_MAX_FILTERS], *b1[ADAISP_POLICY_MAX_FILTERS];   /* [hid][D], [hid]            */
    const float *wf[ADAISP_POLICY_MAX_FILTERS], *bf[ADAISP_POLICY_MAX_FILTERS];   /* [n_f][hid], [n_f]          */
    const float *ws1, *bs1, *ws2, *bs2;                        /* selector: [hid][D], [hid], [F][hid], [F]       */
    float *hidden, *x, *logits;
    /* backward only */
    const float *dx, *dlogits;
    float *dhid, *part;
    float *dw1[ADAISP_POLICY_MAX_FILTERS], *db1[ADAISP_POLICY_MAX_FILTERS], *dwf[ADAISP_POLICY_MAX_FILTERS], *dbf[ADAISP_POLICY_MAX_FILTERS];
    float *dws1, *dbs1, *dws2, *dbs2, *dfeat_f, *dfeat_s;
} adaisp_heads_args;
int adaisp_heads_fwd(const adaisp_heads_args* args, void* stream);
int adaisp_heads_bwd(const adaisp_heads_args* args, void* stream);

/* Number of regressed parameters an op reads per image (0 for ADAISP_OP_ZERO, -1 if unknown). */
int adaisp_num_params(int op);

const char* adaisp_strerror(int code);
int adaisp_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ADAISP_H_ */
