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

// ---- AdaptiveAvgPool2d((64,64)) window rule (PyTorch): cell o covers [floor(o*n/64), ceil((o+1)*n/64)) -----------
__host__ __device__ inline int win_lo(int o, int n) { return (int)(((long)o * n) / 64); }
__host__ __device__ inline int win_hi(int o, int n) { return (int)((((long)(o + 1)) * n + 63) / 64); }

// Fused pooling geometry: the pixel kernels that also produce the next step's 64x64 pooled planes are cut along the
// pool windows. A WAVE owns a "strip" = `cps` consecutive pool columns (its quad-aligned pixel span <= 256 = 64 lanes x
// 4 px), a WORKGROUP the `strips` strips of one pool row `oy` (rows win_lo(oy,H) .. win_hi(oy,H)). Rows / quads that
// belong to two windows are READ (and computed) by both owners and WRITTEN by the first. Column sums run down the
// window's rows in a lane's registers (ascending y from 0.0f), the x-window sums ascending x from 0.0f through LDS —
// exactly k_pool64's order, so fused and stand-alone pooling are bit-identical.
struct PoolGeom {
    int cps, strips;
    bool ok;
};
__host__ __device__ inline int strip_cell0(int s, int cps) { return s * cps < 64 ? s * cps : 64; }
__host__ __device__ inline int strip_x_lo(int s, int cps, int W) {               // first pixel a strip reads AND owns
    return strip_cell0(s, cps) >= 64 ? W : (win_lo(strip_cell0(s, cps), W) & ~3);
}
__host__ __device__ inline int strip_x_end(int s, int cps, int W) {              // one past the last pixel it reads
    return (win_hi(strip_cell0(s + 1, cps) - 1, W) + 3) & ~3;
}
inline PoolGeom pool_geom(int H, int W, const void* img, const void* out) {
    PoolGeom g{0, 0, false};
    // W <= 8192: the pointwise family keeps a whole row's column sums in LDS (3 x W floats, isp_pointwise.hip)
    if (H < 64 || W < 64 || W > 8192 || (W & 3) || (reinterpret_cast<uintptr_t>(img) & 15) || (reinterpret_cast<uintptr_t>(out) & 15))
        return g;
    for (int strips = 1; strips <= 16; ++strips) {
        const int cps = (64 + strips - 1) / strips;
        if ((64 + cps - 1) / cps != strips) continue;
        bool fits = true;
        for (int s = 0; s < strips && fits; ++s) fits = strip_x_end(s, cps, W) - strip_x_lo(s, cps, W) <= 256;
        if (fits) { g.cps = cps; g.strips = strips; g.ok = true; return g; }
    }
    return g;
}

hipError_t launch_pointwise(const Batch& a, hipStream_t s);
hipError_t launch_pointwise_pool(const Batch& a, float* pooled, const PoolGeom& g, hipStream_t s);
hipError_t launch_conv_pool(const Batch& a, float* pooled, const PoolGeom& g, hipStream_t s);
// pooling of `img` for the images the fused kernels did not serve: with `only_unfused` an image is skipped unless its op
// is NLM (ids on the device, or `uniform_op`)
hipError_t launch_pool64_sel(const float* img, float* pooled, const int32_t* ids, int uniform_op, bool only_unfused,
                             unsigned flags, int B, int H, int W, hipStream_t s);
hipError_t launch_conv(const Batch& a, hipStream_t s);     // 3x3 sharpen, 3x3 sharpness, 5x5 USM
hipError_t launch_nlm(const Batch& a, hipStream_t s);
hipError_t launch_nlm_general(const float* img, float* out, const float* h, int hstride, float* workspace, int B, int H, int W,
                              int search, int patch, hipStream_t s);
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

// x-window sum of a pool cell from column sums in LDS, ascending x from 0.0f (k_pool64's order). The additions are a dependent
// chain either way; when the window is a whole number of quads (W % 256 == 0: xs and the width are multiples of 4) its values
// are fetched with 16-byte reads issued together instead of one 4-byte read per addition (a chain of ~20 LDS round trips at
// the tail of every workgroup of the fused-pooling kernels).
__device__ __forceinline__ float window_sum(const float* col, int xs, int xe) {
    float a = 0.f;
    if (((xs | xe) & 3) == 0 && xe - xs <= 64) {
        float4 v[16];
        const int nq = (xe - xs) >> 2;
#pragma unroll
        for (int q = 0; q < 16; ++q) if (q < nq) v[q] = *reinterpret_cast<const float4*>(col + xs + 4 * q);
#pragma unroll
        for (int q = 0; q < 16; ++q) if (q < nq) { a += v[q].x; a += v[q].y; a += v[q].z; a += v[q].w; }
        return a;
    }
    for (int xx = xs; xx < xe; ++xx) a += col[xx];
    return a;
}

// Streaming access policy (tools/stream_ceiling.hip, profiles/round4_stream_ceiling.txt): on tensors that do not stay in the
// 256 MB Infinity Cache a grid-stride float4 copy moves 6.2 TB/s with non-temporal loads AND stores, 5.6 with plain loads +
// non-temporal stores, 5.2-5.4 with plain / plain (the vendor's copy: 5.1-5.2). Every pixel of the pointwise filters is read
// once and written once: ld4 / st4. The stencil filters re-read halo rows that a neighbouring wave fetched moments ago (L2
// hits with plain loads): they use st4 and plain loads (ISP_CONV_NT_LD measures the other choice).
#ifndef ISP_NT_LD
#define ISP_NT_LD 1
#endif
#ifndef ISP_NT_ST
#define ISP_NT_ST 1
#endif
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_nt(const float4* p) {
    const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 ld4(const float4* p) {
#if ISP_NT_LD
    return ld4_nt(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ void st4(float4* p, const float4& v) {
#if ISP_NT_ST
    __builtin_nontemporal_store(f32x4_t{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4_t*>(p));
#else
    *p = v;
#endif
}

}  // namespace adaisp
