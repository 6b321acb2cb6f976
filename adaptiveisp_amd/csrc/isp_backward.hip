// Parameter gradients of the ISP filters for gfx950:
//   grad_params[b][k] = sum_{c,y,x} grad_out[b][c][y][x] * gate * d f_c / d params[b][k]
// where f is the selected filter's `process` and gate is the pass-through mask of the output clip
// (torch.clip backward: 1 where 0 <= f <= 1). This is the only gradient the reference's training
// needs (train.py:341-342: the image is a constant leaf, the heads learn through the parameters).
//
// Each lane accumulates its pixels' contributions in registers (grid-stride), waves reduce with
// shuffles, the workgroup through LDS, then ONE float atomic per parameter per workgroup.
// The stencil ops re-read their neighbourhood through L1/L2 (this is not the headline path);
// NLM reuses the forward kernel's tile machinery (isp_nlm.hip, GRAD instantiation).
#include "isp_internal.h"

namespace adaisp {
namespace {

constexpr int kThreads = 256;

template <int NG>
__device__ __forceinline__ void block_reduce_atomic(float (&g)[NG], float* __restrict__ dst) {
    __shared__ float red[4][NG];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        float v = g[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < NG) {
        const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (v != 0.0f) atomicAdd(dst + threadIdx.x, v);
    }
}

__device__ __forceinline__ float gate01(float f, bool clip) { return (!clip || (f >= 0.0f && f <= 1.0f)) ? 1.0f : 0.0f; }
__device__ __forceinline__ float lum276(float r, float g, float b) { return (0.27f * r + 0.67f * g) + 0.06f * b; }
__device__ __forceinline__ float py_mod(float a, float m) {
    float r = fmodf(a, m);
    if (r != 0.0f && r < 0.0f) r += m;
    return r;
}

// full-colour term of SaturationPlus (same arithmetic as the forward kernel)
__device__ __forceinline__ void sat_full(float r, float g, float b, float& fr, float& fg, float& fb) {
    const float mx = fmaxf(fmaxf(r, g), b), mn = fminf(fminf(r, g), b);
    const float d = (mx - mn) + 1e-8f;
    float hue = 0.0f;
    if (b == mx) hue = 4.0f + (r - g) / d;
    if (g == mx) hue = 2.0f + (b - r) / d;
    if (r == mx) hue = py_mod((g - b) / d, 6.0f);
    if (mn == mx) hue = 0.0f;
    hue = hue / 6.0f;
    float s = (mx - mn) / (mx + 1e-8f);
    if (mx == 0.0f) s = 0.0f;
    const float es = s + (1.0f - s) * (0.5f - fabsf(0.5f - mx)) * 0.8f;
    const float h = py_mod(hue, 1.0f), s2 = clamp01(es), v2 = clamp01(mx);
    const float h6 = h * 6.0f, hi = floorf(h6), f = h6 - hi;
    const float pp = v2 * (1.0f - s2), qq = v2 * (1.0f - (f * s2)), tt = v2 * (1.0f - ((1.0f - f) * s2));
    fr = fg = fb = 0.0f;
    if (hi == 0.0f) { fr = v2; fg = tt; fb = pp; }
    else if (hi == 1.0f) { fr = qq; fg = v2; fb = pp; }
    else if (hi == 2.0f) { fr = pp; fg = v2; fb = tt; }
    else if (hi == 3.0f) { fr = pp; fg = qq; fb = v2; }
    else if (hi == 4.0f) { fr = tt; fg = pp; fb = v2; }
    else if (hi == 5.0f) { fr = v2; fg = pp; fb = qq; }
}

// ---- pointwise ops ---------------------------------------------------------------------------------
template <int OP, int NG>
__device__ void bwd_pointwise(const float* __restrict__ in, const float* __restrict__ go, const float* __restrict__ p,
                              float* __restrict__ gp, long plane, bool clip) {
    float g[NG];
#pragma unroll
    for (int k = 0; k < NG; ++k) g[k] = 0.0f;
    // per-image constants
    float c0 = 0.f, rs[3] = {1.f, 1.f, 1.f}, S[3] = {1.f, 1.f, 1.f};
    if (OP == ADAISP_OP_EXPOSURE) c0 = expf(p[0] * 0.6931471805599453f);
    if (OP == ADAISP_OP_CCM)
        for (int i = 0; i < 3; ++i) rs[i] = (p[3 * i] + p[3 * i + 1]) + p[3 * i + 2];
    if (OP == ADAISP_OP_TONE) { float s = 0.f; for (int i = 0; i < 8; ++i) s += p[i]; S[0] = S[1] = S[2] = s + 1e-30f; }
    if (OP == ADAISP_OP_COLOR)
        for (int ch = 0; ch < 3; ++ch) { float s = 0.f; for (int i = 0; i < 8; ++i) s += p[3 * i + ch]; S[ch] = s + 1e-30f; }

    const long stride = (long)gridDim.x * kThreads;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < plane; i += stride) {
        const float x[3] = {in[i], in[i + plane], in[i + 2 * plane]};
        const float go3[3] = {go[i], go[i + plane], go[i + 2 * plane]};
        if (OP == ADAISP_OP_EXPOSURE) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { const float f = x[c] * c0; g[0] += go3[c] * gate01(f, clip) * f * 0.6931471805599453f; }
        } else if (OP == ADAISP_OP_GAMMA) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float m = fmaxf(x[c], 0.001f), f = powf(m, p[0]);
                g[0] += go3[c] * gate01(f, clip) * f * logf(m);
            }
        } else if (OP == ADAISP_OP_WB) {
#pragma unroll
            for (int c = 0; c < 3; ++c) g[c] += go3[c] * gate01(x[c] * p[c], clip) * x[c];
        } else if (OP == ADAISP_OP_CCM) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float f = (x[0] * (p[3 * c] / rs[c]) + x[1] * (p[3 * c + 1] / rs[c])) + x[2] * (p[3 * c + 2] / rs[c]);
                const float gg = go3[c] * gate01(f, clip) / rs[c];
#pragma unroll
                for (int j = 0; j < 3; ++j) g[3 * c + j] += gg * (x[j] - f);
            }
        } else if (OP == ADAISP_OP_TONE || OP == ADAISP_OP_COLOR) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float t[8], acc = 0.0f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    t[j] = fminf(fmaxf(x[c] - 0.125f * (float)j, 0.0f), 0.125f);
                    acc += t[j] * p[OP == ADAISP_OP_TONE ? j : 3 * j + c];
                }
                const float sc = 8.0f / S[c], f = acc * sc;
                const float gg = go3[c] * gate01(f, clip) * sc;
#pragma unroll
                for (int j = 0; j < 8; ++j) g[OP == ADAISP_OP_TONE ? j : 3 * j + c] += gg * (t[j] - acc / S[c]);
            }
        } else if (OP == ADAISP_OP_CONTRAST) {
            const float L = clamp01(lum276(x[0], x[1], x[2]));
            const float cl = -cosf(3.14159274101257324f * L) * 0.5f + 0.5f, den = L + 1e-6f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float ci = x[c] / den * cl, f = (1.0f - p[0]) * x[c] + p[0] * ci;
                g[0] += go3[c] * gate01(f, clip) * (ci - x[c]);
            }
        } else if (OP == ADAISP_OP_WNB) {
            const float L = lum276(x[0], x[1], x[2]);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float f = (1.0f - p[0]) * x[c] + p[0] * L;
                g[0] += go3[c] * gate01(f, clip) * (L - x[c]);
            }
        } else if (OP == ADAISP_OP_SATPLUS) {
            const float xc[3] = {clamp01(x[0]), clamp01(x[1]), clamp01(x[2])};
            float fc[3];
            sat_full(xc[0], xc[1], xc[2], fc[0], fc[1], fc[2]);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float f = xc[c] * (1.0f - p[0]) + fc[c] * p[0];
                g[0] += go3[c] * gate01(f, clip) * (fc[c] - xc[c]);
            }
        }
    }
    block_reduce_atomic<NG>(g, gp);
}

__global__ __launch_bounds__(kThreads) void k_bwd_pointwise(const float* __restrict__ img, const float* __restrict__ go,
                                                            const int32_t* __restrict__ ids,
                                                            const float* __restrict__ params, int pstride,
                                                            float* __restrict__ gparams, long plane, unsigned flags) {
    const int b = blockIdx.y;
    const float* in = img + (long)b * 3 * plane;
    const float* g = go + (long)b * 3 * plane;
    const float* p = params + (long)b * pstride;
    float* gp = gparams + (long)b * pstride;
    const bool clip = (flags & ADAISP_CLIP01) != 0;
    switch (ids[b]) {
        case ADAISP_OP_EXPOSURE: bwd_pointwise<ADAISP_OP_EXPOSURE, 1>(in, g, p, gp, plane, clip); break;
        case ADAISP_OP_GAMMA:    bwd_pointwise<ADAISP_OP_GAMMA, 1>(in, g, p, gp, plane, clip); break;
        case ADAISP_OP_WB:       bwd_pointwise<ADAISP_OP_WB, 3>(in, g, p, gp, plane, clip); break;
        case ADAISP_OP_CCM:      bwd_pointwise<ADAISP_OP_CCM, 9>(in, g, p, gp, plane, clip); break;
        case ADAISP_OP_TONE:     bwd_pointwise<ADAISP_OP_TONE, 8>(in, g, p, gp, plane, clip); break;
        case ADAISP_OP_COLOR:    bwd_pointwise<ADAISP_OP_COLOR, 24>(in, g, p, gp, plane, clip); break;
        case ADAISP_OP_CONTRAST: bwd_pointwise<ADAISP_OP_CONTRAST, 1>(in, g, p, gp, plane, clip); break;
        case ADAISP_OP_WNB:      bwd_pointwise<ADAISP_OP_WNB, 1>(in, g, p, gp, plane, clip); break;
        case ADAISP_OP_SATPLUS:  bwd_pointwise<ADAISP_OP_SATPLUS, 1>(in, g, p, gp, plane, clip); break;
        default: break;
    }
}

// ---- 3x3 sharpen pair and 5x5 unsharp mask ------------------------------------------------------------
__device__ __forceinline__ int reflect(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

__global__ __launch_bounds__(kThreads) void k_bwd_conv(const float* __restrict__ img, const float* __restrict__ go,
                                                       const int32_t* __restrict__ ids, const float* __restrict__ params,
                                                       int pstride, float* __restrict__ gparams, int H, int W) {
    const int b = blockIdx.y;
    const int op = ids[b];
    if (op != ADAISP_OP_SHARPEN && op != ADAISP_OP_SHARPEN_V2 && op != ADAISP_OP_USM) return;
    const long plane = (long)H * W;
    const float* in = img + (long)b * 3 * plane;
    const float* g = go + (long)b * 3 * plane;
    const float* p = params + (long)b * pstride;
    float acc[2] = {0.0f, 0.0f};
    // USM weights and their sigma-derivative: w_ij = g_i g_j / S^2, g_i = exp(-x_i^2 / (2 sigma^2))
    float w5[5][5], dw5[5][5];
    if (op == ADAISP_OP_USM) {
        const float sg = p[0];
        float g1[5], dg[5], S = 0.0f, dS = 0.0f;
        for (int i = 0; i < 5; ++i) {
            const float xi = (float)(i - 2), t = xi / sg;
            g1[i] = expf(-0.5f * (t * t));
            dg[i] = g1[i] * xi * xi / (sg * sg * sg);
            S += g1[i]; dS += dg[i];
        }
        for (int i = 0; i < 5; ++i)
            for (int j = 0; j < 5; ++j) {
                w5[i][j] = (g1[i] / S) * (g1[j] / S);
                dw5[i][j] = (dg[i] * g1[j] + g1[i] * dg[j]) / (S * S) - 2.0f * g1[i] * g1[j] * dS / (S * S * S);
            }
    }
    const float a13 = 1.0f / 13.0f, c13 = 5.0f / 13.0f;
    const long total = 3 * plane, stride = (long)gridDim.x * kThreads;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i / plane);
        const long r = i - c * plane;
        const int y = (int)(r / W), x = (int)(r - (long)y * W);
        const float* src = in + c * plane;
        const float ctr = src[r], gout = g[i];
        if (op == ADAISP_OP_USM) {
            float blur = 0.0f, dblur = 0.0f;
            for (int ii = 0; ii < 5; ++ii)
                for (int jj = 0; jj < 5; ++jj) {
                    const float v = src[(long)reflect(y + ii - 2, H) * W + reflect(x + jj - 2, W)];
                    blur = fmaf(w5[ii][jj], v, blur);
                    dblur = fmaf(dw5[ii][jj], v, dblur);
                }
            const float f = ctr + (ctr - blur) * p[1];
            const float gt = gout * ((f >= 0.0f && f <= 1.0f) ? 1.0f : 0.0f);
            acc[0] += gt * (-p[1] * dblur);
            acc[1] += gt * (ctr - blur);
        } else {
            float blur = ctr;
            if (y > 0 && y < H - 1 && x > 0 && x < W - 1) {
                blur = 0.0f;
                for (int ii = 0; ii < 3; ++ii)
                    for (int jj = 0; jj < 3; ++jj)
                        blur = fmaf((ii == 1 && jj == 1) ? c13 : a13, src[(long)(y + ii - 1) * W + (x + jj - 1)], blur);
            }
            const float f = (op == ADAISP_OP_SHARPEN) ? ctr * p[0] + blur * (1.0f - p[0]) : ctr + (ctr - blur) * p[0];
            acc[0] += gout * ((f >= 0.0f && f <= 1.0f) ? 1.0f : 0.0f) * (ctr - blur);
        }
    }
    block_reduce_atomic<2>(acc, gparams + (long)b * pstride);
}

}  // namespace

hipError_t launch_nlm_backward(const float* img, const float* grad_out, const int32_t* ids, const float* params,
                               int pstride, float* grad_params, int B, int H, int W, unsigned flags, hipStream_t s);

hipError_t launch_backward_params(const float* img, const float* grad_out, const int32_t* ids, const float* params,
                                  int pstride, float* grad_params, int B, int H, int W, unsigned flags, hipStream_t s) {
    hipError_t e = hipMemsetAsync(grad_params, 0, sizeof(float) * (size_t)B * pstride, s);
    if (e != hipSuccess) return e;
    const long plane = (long)H * W;
    long bx = (plane + kThreads * 8 - 1) / (kThreads * 8);
    if (bx > 512) bx = 512;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(k_bwd_pointwise, dim3((unsigned)bx, B), dim3(kThreads), 0, s, img, grad_out, ids, params, pstride,
                       grad_params, plane, flags);
    if (H >= 3 && W >= 3)
        hipLaunchKernelGGL(k_bwd_conv, dim3((unsigned)bx, B), dim3(kThreads), 0, s, img, grad_out, ids, params, pstride,
                           grad_params, H, W);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_nlm_backward(img, grad_out, ids, params, pstride, grad_params, B, H, W, flags, s);
}

}  // namespace adaisp
