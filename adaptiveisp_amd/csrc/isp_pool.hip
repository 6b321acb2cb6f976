// AdaptiveAvgPool2d((64,64)) of a planar [B,3,H,W] fp32 image for gfx950 — the policy/critic input
// (agent.py:85,97; value.py:61,63). PyTorch window rule: output cell o covers input
// [floor(o*n/64), ceil((o+1)*n/64)) along each axis (windows overlap when n % 64 != 0, and repeat
// pixels when n < 64).
//
// One workgroup per (image, output row): lanes sweep the row window column-wise with 16-B loads
// (each input row is read once per window it belongs to, ~1.07x the image at 720 rows), column
// sums go to LDS, then 64 lanes reduce their column windows. Deterministic (no atomics).
#include "isp_internal.h"

#ifndef ISP_POOL_NT_LD
#define ISP_POOL_NT_LD 0
#endif

namespace adaisp {
namespace {

constexpr int kThreads = 256;

// grid: (64 output rows, B, XCH column chunks of 64/XCH output cells); rows of the window are unrolled 4-deep so
// several 16-B loads per lane are in flight.
constexpr int XCH = 4;

// `only_nlm`: pool only the images whose op is NLM (the pointwise and stencil kernels of the same RL step have already
// written the pooled planes of theirs, fused: isp_internal.h PoolGeom); ids on the device or one host-known op.
template <bool VEC>
__global__ __launch_bounds__(kThreads) void k_pool64(const float* __restrict__ img, float* __restrict__ pooled,
                                                     int H, int W, const int32_t* __restrict__ ids, int uniform_op,
                                                     int only_nlm) {
    extern __shared__ __attribute__((aligned(16))) float colsum[];   // [3][span] column sums of this chunk
    const int oy = blockIdx.x, b = blockIdx.y, ch = blockIdx.z;
    if (only_nlm && (ids ? ids[b] : uniform_op) != ADAISP_OP_NLM) return;
    const int ox0 = ch * (64 / XCH), ox1 = ox0 + 64 / XCH;
    const int ys = win_lo(oy, H), ye = win_hi(oy, H);
    const int x_lo = win_lo(ox0, W) & ~3, x_hi = win_hi(ox1 - 1, W);  // quad-aligned start of the chunk's columns
    const int span = x_hi - x_lo;
    const float kh = (float)(ye - ys);
    const long plane = (long)H * W;
    const float* __restrict__ base = img + (long)b * 3 * plane;
    if (VEC) {
        const int nq = (span + 3) >> 2;
        for (int i = threadIdx.x; i < 3 * nq; i += kThreads) {
            const int c = i / nq, q = i - c * nq;
            const int x = x_lo + 4 * q;
            const float* src = base + c * plane + x;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (x + 3 < W) {
#pragma unroll 4
                for (int y = ys; y < ye; ++y) {
#if ISP_POOL_NT_LD
                    const float4 v = ld4_nt(reinterpret_cast<const float4*>(src + (long)y * W));
#else
                    const float4 v = *reinterpret_cast<const float4*>(src + (long)y * W);
#endif
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
            }
            *reinterpret_cast<float4*>(colsum + c * ((span + 7) & ~3) + 4 * q) = acc;
        }
    } else {
        for (int i = threadIdx.x; i < 3 * span; i += kThreads) {
            const int c = i / span, xo = i - c * span;
            const float* src = base + c * plane + x_lo + xo;
            float acc = 0.f;
            for (int y = ys; y < ye; ++y) acc += src[(long)y * W];
            colsum[c * ((span + 7) & ~3) + xo] = acc;
        }
    }
    __syncthreads();
    if (threadIdx.x < 3 * (64 / XCH)) {
        const int c = threadIdx.x / (64 / XCH), ox = ox0 + threadIdx.x % (64 / XCH);
        const int xs = win_lo(ox, W), xe = win_hi(ox, W);
        const float* cs = colsum + c * ((span + 7) & ~3) - x_lo;
        float acc = 0.f;
        for (int x = xs; x < xe; ++x) acc += cs[x];
        pooled[(((long)b * 3 + c) * 64 + oy) * 64 + ox] = acc / kh / (float)(xe - xs);
    }
}

// Backward of the pooling: grad_img[y][x] = sum over the (1..2, more when n < 64) windows that contain the pixel of
// grad_pooled[oy][ox] / kh / kw, accumulated oy-major in ascending order (ATen's adaptive_avg_pool2d_backward order).
// The critic reads the retouched image through this pooling and back-propagates into the filter parameters
// (train.py:281-305: agent_loss contains -q_value = -(reward + gamma * V(retouch, new_states)) when cfg.use_TD).
// One thread = one column x over kBwdRows rows: the column's windows (1..2, more when W < 64) and their widths are found once,
// the rows' are uniform over the workgroup (scalar unit); the per-pixel work is the two divisions and the store.
constexpr int kBwdRows = 8;
__global__ __launch_bounds__(kThreads) void k_pool64_bwd(const float* __restrict__ gp, float* __restrict__ gimg, int H,
                                                         int W) {
    const int x = blockIdx.x * kThreads + threadIdx.x, bc = blockIdx.z;
    if (x >= W) return;
    int ox0 = (int)(((unsigned)x * 64u) / (unsigned)W), ox1 = ox0;
    while (ox0 > 0 && win_hi(ox0 - 1, W) > x) --ox0;
    while (ox1 < 63 && win_lo(ox1 + 1, W) <= x) ++ox1;
    const float* g = gp + (long)bc * 64 * 64;
    const int y_end = min((int)(blockIdx.y + 1) * kBwdRows, H);
    for (int y = blockIdx.y * kBwdRows; y < y_end; ++y) {
        int oy0 = (int)(((unsigned)y * 64u) / (unsigned)H), oy1 = oy0;
        while (oy0 > 0 && win_hi(oy0 - 1, H) > y) --oy0;
        while (oy1 < 63 && win_lo(oy1 + 1, H) <= y) ++oy1;
        float acc = 0.f;
        for (int oy = oy0; oy <= oy1; ++oy) {
            const float kh = (float)(win_hi(oy, H) - win_lo(oy, H));
            for (int ox = ox0; ox <= ox1; ++ox)
                acc += g[oy * 64 + ox] / kh / (float)(win_hi(ox, W) - win_lo(ox, W));
        }
        gimg[((long)bc * H + y) * W + x] = acc;
    }
}

}  // namespace

hipError_t launch_pool64_bwd(const float* grad_pooled, float* grad_img, int B, int H, int W, hipStream_t s) {
    if ((long)W * 64 >= (1L << 32) || (long)H * 64 >= (1L << 32)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_pool64_bwd, dim3((W + kThreads - 1) / kThreads, (H + kBwdRows - 1) / kBwdRows, B * 3), dim3(kThreads), 0, s, grad_pooled,
                       grad_img, H, W);
    return hipGetLastError();
}

hipError_t launch_pool64_sel(const float* img, float* pooled, const int32_t* ids, int uniform_op, bool only_unfused,
                             unsigned flags, int B, int H, int W, hipStream_t s) {
    (void)flags;
    const bool vec = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(img) & 15) == 0);
    dim3 grid(64, B, XCH);
    const size_t smem = 3 * ((size_t)(W + XCH - 1) / XCH + 16 + W / 64) * sizeof(float);   // >= 3 x padded chunk span
    if (vec) hipLaunchKernelGGL(k_pool64<true>, grid, dim3(kThreads), smem, s, img, pooled, H, W, ids, uniform_op, only_unfused ? 1 : 0);
    else hipLaunchKernelGGL(k_pool64<false>, grid, dim3(kThreads), smem, s, img, pooled, H, W, ids, uniform_op, only_unfused ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_pool64(const float* img, float* pooled, int B, int H, int W, hipStream_t s) {
    return launch_pool64_sel(img, pooled, nullptr, 0, false, 0u, B, H, W, s);
}

}  // namespace adaisp
