// Internal launcher declarations shared by the translation units of libadaisp.so.
// Everything here is gfx950 (CDNA4, wave64) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/adaisp.h"

namespace adaisp {

// `ids` == nullptr  ->  every image uses `uniform_op` (host-known op);
// `ids` != nullptr  ->  image b uses ids[b]; a kernel family skips images whose op it does not own.
struct Batch {
    const float* img;
    float* out;
    const int32_t* ids;
    int uniform_op;
    const float* params;
    int pstride;
    int B, H, W;
    unsigned flags;
};

__host__ __device__ inline bool op_is_pointwise(int op) {
    return op == ADAISP_OP_ZERO || op == ADAISP_OP_EXPOSURE || op == ADAISP_OP_GAMMA ||
           op == ADAISP_OP_CCM || op == ADAISP_OP_TONE || op == ADAISP_OP_CONTRAST ||
           op == ADAISP_OP_SATPLUS || op == ADAISP_OP_WNB || op == ADAISP_OP_WB ||
           op == ADAISP_OP_COLOR;
}
__host__ __device__ inline bool op_is_conv(int op) {
    return op == ADAISP_OP_SHARPEN || op == ADAISP_OP_SHARPEN_V2 || op == ADAISP_OP_USM;
}

hipError_t launch_pointwise(const Batch& a, hipStream_t s);
hipError_t launch_conv(const Batch& a, hipStream_t s);     // 3x3 sharpen, 3x3 sharpness, 5x5 USM
hipError_t launch_nlm(const Batch& a, hipStream_t s);
hipError_t launch_pool64(const float* img, float* pooled, int B, int H, int W, hipStream_t s);
hipError_t launch_pool64_bwd(const float* grad_pooled, float* grad_img, int B, int H, int W, hipStream_t s);
hipError_t launch_demosaic(const uint16_t* raw, float* out, int B, int H, int W, int pattern, float black, float white,
                           hipStream_t s);
hipError_t launch_backward_params(const float* img, const float* grad_out, const int32_t* ids,
                                  const float* params, int pstride, float* grad_params,
                                  int B, int H, int W, unsigned flags, hipStream_t s);

hipError_t launch_policy_conv(const float* in, const float* states, int n_state, const float* w, const float* bias,
                              float* out, int G, int B, int Cin, int Hin, int Cout, hipStream_t s);
hipError_t launch_policy_fc1(const float* feats, const int32_t* head_src, const float* w1, const float* b1,
                             float* hidden, int B, int D, int NH, int HID, hipStream_t s);
hipError_t launch_policy_finish(const adaisp_policy_finish_args& a, int B, hipStream_t s);

// Output clamp of Filter.forward (isp/filters.py:125). With `clip` false the bounds are +-inf.
struct Clip {
    float lo, hi;
    __device__ explicit Clip(bool clip)
        : lo(clip ? 0.0f : -__builtin_huge_valf()), hi(clip ? 1.0f : __builtin_huge_valf()) {}
    __device__ __forceinline__ float operator()(float v) const { return fminf(fmaxf(v, lo), hi); }
};

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

}  // namespace adaisp
