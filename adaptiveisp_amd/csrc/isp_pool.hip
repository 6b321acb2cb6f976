// AdaptiveAvgPool2d((64,64)) of a planar [B,3,H,W] fp32 image for gfx950 — the policy/critic input
// (agent.py:85,97; value.py:61,63). PyTorch window rule: output cell o covers input
// [floor(o*n/64), ceil((o+1)*n/64)) along each axis (windows overlap when n % 64 != 0, and repeat
// pixels when n < 64).
//
// One workgroup per (image, output row): lanes sweep the row window column-wise with 16-B loads
// (each input row is read once per window it belongs to, ~1.07x the image at 720 rows), column
// sums go to LDS, then 64 lanes reduce their column windows. Deterministic (no atomics).
#include "isp_internal.h"

namespace adaisp {
namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ int win_lo(int o, int n) { return (int)(((long)o * n) / 64); }
__device__ __forceinline__ int win_hi(int o, int n) { return (int)((((long)(o + 1)) * n + 63) / 64); }

template <bool VEC>
__global__ __launch_bounds__(kThreads) void k_pool64(const float* __restrict__ img, float* __restrict__ pooled,
                                                     int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float colsum[];   // [W]
    const int oy = blockIdx.x, b = blockIdx.y;
    const int ys = win_lo(oy, H), ye = win_hi(oy, H);
    const float kh = (float)(ye - ys);
    const long plane = (long)H * W;
    for (int c = 0; c < 3; ++c) {
        const float* __restrict__ src = img + ((long)b * 3 + c) * plane;
        if (VEC) {
            for (int x = 4 * threadIdx.x; x < W; x += 4 * kThreads) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int y = ys; y < ye; ++y) {
                    const float4 v = *reinterpret_cast<const float4*>(src + (long)y * W + x);
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
                *reinterpret_cast<float4*>(colsum + x) = acc;
            }
        } else {
            for (int x = threadIdx.x; x < W; x += kThreads) {
                float acc = 0.f;
                for (int y = ys; y < ye; ++y) acc += src[(long)y * W + x];
                colsum[x] = acc;
            }
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const int ox = threadIdx.x;
            const int xs = win_lo(ox, W), xe = win_hi(ox, W);
            float acc = 0.f;
            for (int x = xs; x < xe; ++x) acc += colsum[x];
            pooled[(((long)b * 3 + c) * 64 + oy) * 64 + ox] = acc / kh / (float)(xe - xs);
        }
        __syncthreads();
    }
}

}  // namespace

hipError_t launch_pool64(const float* img, float* pooled, int B, int H, int W, hipStream_t s) {
    const bool vec = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(img) & 15) == 0);
    dim3 grid(64, B);
    const size_t smem = (size_t)W * sizeof(float);
    if (vec) hipLaunchKernelGGL(k_pool64<true>, grid, dim3(kThreads), smem, s, img, pooled, H, W);
    else hipLaunchKernelGGL(k_pool64<false>, grid, dim3(kThreads), smem, s, img, pooled, H, W);
    return hipGetLastError();
}

}  // namespace adaisp
