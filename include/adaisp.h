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

#define ADAISP_ABI_VERSION 2

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
 * `filter_id` lives on the device (no host sync is needed to pick the work); -1 writes zeros.
 * If `pooled64_next` != NULL it receives AdaptiveAvgPool2d((64,64)) of `out`
 * ([B,3,64,64]; agent.py:97 / value.py:63) for the next step's policy input.
 * `out` must not alias `img`.
 */
int adaisp_forward(const float* img, float* out, float* pooled64_next,
                   const int32_t* filter_id, const float* params, int param_stride,
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

/* AdaptiveAvgPool2d((64,64)) of a [B,3,H,W] image: agent.py:85,97, value.py:61,63. */
int adaisp_pool64(const float* img, float* pooled, int B, int H, int W, void* stream);

/* Number of regressed parameters an op reads per image (0 for ADAISP_OP_ZERO, -1 if unknown). */
int adaisp_num_params(int op);

const char* adaisp_strerror(int code);
int adaisp_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ADAISP_H_ */
