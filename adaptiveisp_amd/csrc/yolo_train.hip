// Training-side kernels of the detector (the reward model is frozen, train.py:239-243: only the gradient with respect
// to its INPUT image is needed). The convolutions of the backward pass are the forward implicit-GEMM kernels run on
// transposed / flipped weights (data gradient of a stride-1 conv is a stride-1 conv; a stride-2 conv's data gradient
// is the stride-1 conv of the zero-inserted output gradient); what is left is element-wise and lives here:
//
//   k_silu_fwd       y = silu(p) (+ residual)                  train forward keeps the pre-activation p
//   k_silu_bwd       gp = gy * silu'(p); gres (+)= gy          silu'(p) = s + p s (1 - s), s = sigmoid(p)
//   k_zero_insert    u[b,2y,2x] = g[b,y,x], 0 elsewhere        data gradient of the stride-2 convs
//   k_upsample_bwd   gx[b,y,x] (+)= sum of the 2x2 block of gy  nn.Upsample(2,'nearest') backward
//   k_image_grad     NHWC bf16 (3 of 8 channels) -> planar fp32 [B,3,H,W] without the letterbox rows
//   k_stem (act)     the stem without SiLU, so its pre-activation can be kept
// All tensors NHWC bf16 with explicit channel strides (channel slices of concat buffers), 16-byte vectors.
#include "yolo_internal.h"

namespace adayolo {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ float lo_f(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float hi_f(unsigned v) { return __uint_as_float(v & 0xFFFF0000u); }
__device__ __forceinline__ unsigned pk(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
__device__ __forceinline__ float sigmoidf_(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}

__global__ __launch_bounds__(256) void k_silu_fwd(const unsigned short* __restrict__ pre, int pre_cs,
                                                  const unsigned short* __restrict__ res, int res_cs,
                                                  unsigned short* __restrict__ out, int out_cs, long npix, int C) {
    const int cpp = C / 8;
    const long total = npix * cpp;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long pix = i / cpp;
        const int ch = (int)(i - pix * cpp) * 8;
        const u32x4 p = *reinterpret_cast<const u32x4*>(pre + pix * pre_cs + ch);
        u32x4 r = {0u, 0u, 0u, 0u};
        if (res) r = *reinterpret_cast<const u32x4*>(res + pix * res_cs + ch);
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = lo_f(p[j]), b = hi_f(p[j]);
            float ya = a * sigmoidf_(a), yb = b * sigmoidf_(b);
            if (res) {                       // forward conv kernels round to bf16 before the shortcut add: same here
                ya = lo_f(pk(ya, 0.f)) + lo_f(r[j]);
                yb = lo_f(pk(yb, 0.f)) + hi_f(r[j]);
            }
            o[j] = pk(ya, yb);
        }
        *reinterpret_cast<u32x4*>(out + pix * out_cs + ch) = o;
    }
}

__global__ __launch_bounds__(256) void k_silu_bwd(const unsigned short* __restrict__ gy, int gy_cs,
                                                  const unsigned short* __restrict__ pre, int pre_cs,
                                                  unsigned short* __restrict__ gp, int gp_cs,
                                                  unsigned short* __restrict__ gres, int gres_cs, int accumulate,
                                                  long npix, int C) {
    const int cpp = C / 8;
    const long total = npix * cpp;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long pix = i / cpp;
        const int ch = (int)(i - pix * cpp) * 8;
        const u32x4 g = *reinterpret_cast<const u32x4*>(gy + pix * gy_cs + ch);
        if (gp) {
            const u32x4 p = *reinterpret_cast<const u32x4*>(pre + pix * pre_cs + ch);
            u32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = dsilu_bf16x2(g[j], p[j]);
            *reinterpret_cast<u32x4*>(gp + pix * gp_cs + ch) = o;
        }
        if (gres) {
            unsigned short* dst = gres + pix * gres_cs + ch;
            u32x4 o = g;
            if (accumulate) {
                const u32x4 old = *reinterpret_cast<const u32x4*>(dst);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = pk(lo_f(g[j]) + lo_f(old[j]), hi_f(g[j]) + hi_f(old[j]));
            }
            *reinterpret_cast<u32x4*>(dst) = o;
        }
    }
}

// out: [B, H, W, C] (H = 2*Ho or 2*Ho-1 rows of the conv INPUT), out[b, 2y, 2x] = in[b, y, x], zero elsewhere
__global__ __launch_bounds__(256) void k_zero_insert(const unsigned short* __restrict__ in, int in_cs,
                                                     unsigned short* __restrict__ out, int out_cs, int B, int Ho, int Wo,
                                                     int H, int W, int C) {
    const int cpp = C / 8;
    const long total = (long)B * H * W * cpp;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % cpp) * 8;
        const long pix = i / cpp;
        const int x = (int)(pix % W);
        const long by = pix / W;
        const int y = (int)(by % H);
        const long b = by / H;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (!(x & 1) && !(y & 1) && (y >> 1) < Ho && (x >> 1) < Wo)
            v = *reinterpret_cast<const u32x4*>(in + ((b * Ho + (y >> 1)) * Wo + (x >> 1)) * in_cs + ch);
        *reinterpret_cast<u32x4*>(out + pix * out_cs + ch) = v;
    }
}

__global__ __launch_bounds__(256) void k_upsample_bwd(const unsigned short* __restrict__ gy, int gy_cs,
                                                      unsigned short* __restrict__ gx, int gx_cs, int accumulate, int B,
                                                      int H, int W, int C) {
    const int cpp = C / 8;
    const long total = (long)B * H * W * cpp;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % cpp) * 8;
        const long pix = i / cpp;
        const int x = (int)(pix % W);
        const long by = pix / W;
        const int y = (int)(by % H);
        const long b = by / H;
        const unsigned short* s = gy + ((b * 2 * H + 2 * y) * (2L * W) + 2 * x) * gy_cs + ch;
        const u32x4 a = *reinterpret_cast<const u32x4*>(s), c = *reinterpret_cast<const u32x4*>(s + gy_cs);
        const u32x4 d = *reinterpret_cast<const u32x4*>(s + 2L * W * gy_cs);
        const u32x4 e = *reinterpret_cast<const u32x4*>(s + (2L * W + 1) * gy_cs);
        unsigned short* dst = gx + pix * gx_cs + ch;
        u32x4 old = {0u, 0u, 0u, 0u};
        if (accumulate) old = *reinterpret_cast<const u32x4*>(dst);
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = pk(((lo_f(a[j]) + lo_f(c[j])) + (lo_f(d[j]) + lo_f(e[j]))) + lo_f(old[j]),
                      ((hi_f(a[j]) + hi_f(c[j])) + (hi_f(d[j]) + hi_f(e[j]))) + hi_f(old[j]));
        *reinterpret_cast<u32x4*>(dst) = o;
    }
}

__global__ __launch_bounds__(256) void k_image_grad(const unsigned short* __restrict__ g, int g_cs,
                                                    float* __restrict__ grad_img, int B, int H, int W, int Hp, int pad_top) {
    const long plane = (long)H * W, total = (long)B * plane;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long b = i / plane, r = i - b * plane;
        const int y = (int)(r / W), x = (int)(r - (long)y * W);
        const unsigned short* s = g + ((b * Hp + y + pad_top) * (long)W + x) * g_cs;
        const unsigned v01 = *reinterpret_cast<const unsigned*>(s);
        const unsigned v2 = s[2];
        float* o = grad_img + b * 3 * plane + r;
        o[0] = lo_f(v01); o[plane] = hi_f(v01); o[2 * plane] = lo_f(v2);
    }
}

static inline unsigned grid_for(long total) {
    long blocks = (total + 255) / 256;
    return (unsigned)(blocks > 8192 ? 8192 : (blocks < 1 ? 1 : blocks));
}

hipError_t launch_silu_fwd(const void* pre, int pre_cs, const void* res, int res_cs, void* out, int out_cs, long npix,
                           int C, hipStream_t s) {
    hipLaunchKernelGGL(k_silu_fwd, dim3(grid_for(npix * (C / 8))), dim3(256), 0, s, static_cast<const unsigned short*>(pre),
                       pre_cs, static_cast<const unsigned short*>(res), res_cs, static_cast<unsigned short*>(out), out_cs, npix, C);
    return hipGetLastError();
}
hipError_t launch_silu_bwd(const void* gy, int gy_cs, const void* pre, int pre_cs, void* gp, int gp_cs, void* gres,
                           int gres_cs, int accumulate, long npix, int C, hipStream_t s) {
    hipLaunchKernelGGL(k_silu_bwd, dim3(grid_for(npix * (C / 8))), dim3(256), 0, s, static_cast<const unsigned short*>(gy),
                       gy_cs, static_cast<const unsigned short*>(pre), pre_cs, static_cast<unsigned short*>(gp), gp_cs,
                       static_cast<unsigned short*>(gres), gres_cs, accumulate, npix, C);
    return hipGetLastError();
}
hipError_t launch_zero_insert(const void* in, int in_cs, void* out, int out_cs, int B, int Ho, int Wo, int H, int W, int C,
                              hipStream_t s) {
    hipLaunchKernelGGL(k_zero_insert, dim3(grid_for((long)B * H * W * (C / 8))), dim3(256), 0, s,
                       static_cast<const unsigned short*>(in), in_cs, static_cast<unsigned short*>(out), out_cs, B, Ho, Wo, H, W, C);
    return hipGetLastError();
}
hipError_t launch_upsample_bwd(const void* gy, int gy_cs, void* gx, int gx_cs, int accumulate, int B, int H, int W, int C,
                               hipStream_t s) {
    hipLaunchKernelGGL(k_upsample_bwd, dim3(grid_for((long)B * H * W * (C / 8))), dim3(256), 0, s,
                       static_cast<const unsigned short*>(gy), gy_cs, static_cast<unsigned short*>(gx), gx_cs, accumulate, B, H, W, C);
    return hipGetLastError();
}
hipError_t launch_image_grad(const void* g, int g_cs, float* grad_img, int B, int H, int W, int Hp, int pad_top,
                             hipStream_t s) {
    hipLaunchKernelGGL(k_image_grad, dim3(grid_for((long)B * H * W)), dim3(256), 0, s, static_cast<const unsigned short*>(g),
                       g_cs, grad_img, B, H, W, Hp, pad_top);
    return hipGetLastError();
}

}  // namespace adayolo
