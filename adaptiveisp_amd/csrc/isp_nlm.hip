// Non-local-means denoise (gray-weighted, 11x11 search, 5x5 patch) for gfx950.
//
// Reference: NonLocalMeansGray(search_window_size=11, patch_size=5), isp/denoise.py:93-119, with
// BoxFilter :46-65 and rgb_to_luminance :11-17; called from DenoiseFilter.process, isp/filters.py:583-586.
// The reference evaluates 121 full-image passes of ~65 ATen ops each (torch.roll = circular wrap);
// here one launch does all of it from LDS:
//
//   * a 256-thread workgroup owns a 64x32 output tile; the clamped RGB tile (+5 halo) and its
//     luminance (+7 halo = 5 search + 2 patch) are staged once into LDS with circular addressing;
//   * each lane owns one column x 8 rows. The 5x12 centre luminances it needs stay in registers for
//     the whole kernel; per shift it reads the 5x12 shifted luminances from LDS (consecutive lanes ->
//     consecutive banks, conflict-free), forms the squared differences and adds the 25 patch terms
//     of each pixel IN THE REFERENCE'S ORDER (patch column outer, patch row inner, starting from 0),
//     so the patch distance is bit-identical to the roll-based chain;
//   * weights/accumulators (3 colour sums + 1 weight sum per pixel) never leave registers; shifts are
//     visited x-shift outer / y-shift inner like the reference, so num/den round the same way.
//
// No barriers inside the 121-shift loop. Compute-bound on the fp32 VALU (≈3.3 kflop/px + 363
// sqrt/div/exp per px), not on HBM (24 B/px).
#include "isp_internal.h"

namespace adaisp {
namespace {

constexpr int kThreads = 256;
constexpr int TW = 64, TH = 32, RPT = 8;      // tile, rows per thread
constexpr int SR = 5, PR = 2;                 // search radius, patch radius
constexpr int HY = SR + PR;                   // luminance halo (7)
constexpr int YP = TW + 2 * HY;               // 78
constexpr int YROWS = TH + 2 * HY;            // 46
constexpr int CP = TW + 2 * SR;               // 74
constexpr int CROWS = TH + 2 * SR;            // 42
constexpr int NK = RPT + 2 * PR;              // 12 rows of squared differences per column

__device__ __forceinline__ int wrap(int v, int n) {
    v %= n;
    return v < 0 ? v + n : v;
}

// GRAD = false: forward (writes `out`). GRAD = true: d/dh of the forward, contracted with grad_out and summed over
// the image: with w_s = exp(-d_s/hh), d w_s/dh = w_s d_s / hh^2 (h > 0), so
//   d out_c/dh = (A_c - out_c * Bsum) / den,  A_c = sum_s x_sc w_s d_s / hh^2,  Bsum = sum_s w_s d_s / hh^2.
template <bool GRAD>
__global__ __launch_bounds__(kThreads) void k_nlm(const float* __restrict__ img, float* __restrict__ out,
                                                  const int32_t* __restrict__ ids, int uniform_op,
                                                  const float* __restrict__ params, int pstride, int H, int W,
                                                  const float* __restrict__ grad_out, float* __restrict__ grad_params) {
    __shared__ float ylds[YROWS * YP];
    __shared__ float clds[3 * CROWS * CP];

    const int b = blockIdx.z;
    const int op = ids ? ids[b] : uniform_op;
    if (op != ADAISP_OP_NLM) return;
    const long plane = (long)H * W;
    const float* __restrict__ in = img + (long)b * 3 * plane;
    float* __restrict__ o = out + (long)b * 3 * plane;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int tid = threadIdx.x;

    // ---- stage clamp(rgb) and its luminance with circular addressing (torch.roll semantics) --------
    for (int q = tid; q < YROWS * YP; q += kThreads) {
        const int ly = q / YP, lx = q - ly * YP;
        const int gy = wrap(y0 + ly - HY, H), gx = wrap(x0 + lx - HY, W);
        const long g = (long)gy * W + gx;
        const float r = clamp01(in[g]), gg = clamp01(in[g + plane]), bb = clamp01(in[g + 2 * plane]);
        ylds[q] = (0.299f * r + 0.587f * gg) + 0.114f * bb;     // denoise.py:17
        const int cy = ly - PR, cx = lx - PR;
        if (cy >= 0 && cy < CROWS && cx >= 0 && cx < CP) {
            clds[(0 * CROWS + cy) * CP + cx] = r;
            clds[(1 * CROWS + cy) * CP + cx] = gg;
            clds[(2 * CROWS + cy) * CP + cx] = bb;
        }
    }
    __syncthreads();

    const int tx = tid & 63, ty = tid >> 6;
    const int rb = ty * RPT;                       // first output row of this lane inside the tile
    const float hh = fmaxf(params[(long)b * pstride], 0.0f) + 1e-8f;   // relu(h) + EPS, denoise.py:113

    // centre luminances: 5 patch columns x 12 rows, fixed for all 121 shifts
    float yc[2 * PR + 1][NK];
#pragma unroll
    for (int bi = 0; bi < 2 * PR + 1; ++bi)        // bi = bx + PR
#pragma unroll
        for (int k = 0; k < NK; ++k)
            yc[bi][k] = ylds[(rb + HY - PR + k) * YP + tx + HY - (bi - PR)];

    float num[3][RPT], den[RPT];
    float ga[GRAD ? 3 : 1][GRAD ? RPT : 1], gb[GRAD ? RPT : 1];
#pragma unroll
    for (int r = 0; r < RPT; ++r) { num[0][r] = num[1][r] = num[2][r] = 0.0f; den[r] = 0.0f; }
    if (GRAD) {
#pragma unroll
        for (int r = 0; r < RPT; ++r) { ga[0][r] = ga[1][r] = ga[2][r] = 0.0f; gb[r] = 0.0f; }
    }
    const float inv_hh2 = 1.0f / (hh * hh);

    for (int dx = -SR; dx <= SR; ++dx) {           // x_shift outer   (denoise.py:104)
        for (int dy = -SR; dy <= SR; ++dy) {       // y_shift inner   (denoise.py:105)
            // shifted value at (i,j) is the source at (i-dy, j-dx)
            const float* ys = ylds + (rb + HY - PR - dy) * YP + tx + HY - dx;
            float D[RPT];
#pragma unroll
            for (int r = 0; r < RPT; ++r) D[r] = 0.0f;
#pragma unroll
            for (int bi = 0; bi < 2 * PR + 1; ++bi) {      // patch column bx = bi - PR, outer (denoise.py:60)
                float sq[NK];
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const float d = yc[bi][k] - ys[k * YP - (bi - PR)];
                    sq[k] = d * d;
                }
#pragma unroll
                for (int byi = 0; byi < 2 * PR + 1; ++byi)  // patch row by = byi - PR, inner (denoise.py:61)
#pragma unroll
                    for (int r = 0; r < RPT; ++r) D[r] += sq[r + PR - (byi - PR)];
            }
            const float* cs = clds + (rb + SR - dy) * CP + tx + SR - dx;
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const float dist = __fsqrt_rn(fmaxf(D[r], 0.0f));
                const float wgt = expf(-dist / hh);
                const float c0 = cs[(0 * CROWS + r) * CP], c1 = cs[(1 * CROWS + r) * CP], c2 = cs[(2 * CROWS + r) * CP];
                num[0][r] += c0 * wgt;
                num[1][r] += c1 * wgt;
                num[2][r] += c2 * wgt;
                den[r] += wgt;
                if (GRAD) {
                    const float dw = wgt * dist * inv_hh2;
                    ga[0][r] = fmaf(c0, dw, ga[0][r]);
                    ga[1][r] = fmaf(c1, dw, ga[1][r]);
                    ga[2][r] = fmaf(c2, dw, ga[2][r]);
                    gb[r] += dw;
                }
            }
        }
    }

    const int gx = x0 + tx;
    if (!GRAD) {
        if (gx < W) {
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const int gy = y0 + rb + r;
                if (gy < H) {
                    const long g = (long)gy * W + gx;
                    o[g] = clamp01(num[0][r] / den[r]);
                    o[g + plane] = clamp01(num[1][r] / den[r]);
                    o[g + 2 * plane] = clamp01(num[2][r] / den[r]);
                }
            }
        }
    } else {
        float acc = 0.0f;
        const float* go = grad_out + (long)b * 3 * plane;
        if (gx < W && params[(long)b * pstride] > 0.0f) {          // relu(h): no gradient for h <= 0
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const int gy = y0 + rb + r;
                if (gy < H) {
                    const long g = (long)gy * W + gx;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float oc = num[c][r] / den[r];
                        if (oc >= 0.0f && oc <= 1.0f) acc += go[g + c * plane] * (ga[c][r] - oc * gb[r]) / den[r];
                    }
                }
            }
        }
        __shared__ float red[4];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if ((tid & 63) == 0) red[tid >> 6] = acc;
        __syncthreads();
        if (tid == 0) atomicAdd(grad_params + (long)b * pstride, (red[0] + red[1]) + (red[2] + red[3]));
    }
}

}  // namespace

hipError_t launch_nlm(const Batch& a, hipStream_t s) {
    dim3 grid((a.W + TW - 1) / TW, (a.H + TH - 1) / TH, a.B);
    hipLaunchKernelGGL(k_nlm<false>, grid, dim3(kThreads), 0, s, a.img, a.out, a.ids, a.uniform_op, a.params, a.pstride,
                       a.H, a.W, static_cast<const float*>(nullptr), static_cast<float*>(nullptr));
    return hipGetLastError();
}

hipError_t launch_nlm_backward(const float* img, const float* grad_out, const int32_t* ids, const float* params,
                               int pstride, float* grad_params, int B, int H, int W, hipStream_t s) {
    dim3 grid((W + TW - 1) / TW, (H + TH - 1) / TH, B);
    hipLaunchKernelGGL(k_nlm<true>, grid, dim3(kThreads), 0, s, img, static_cast<float*>(nullptr), ids, 0, params,
                       pstride, H, W, grad_out, grad_params);
    return hipGetLastError();
}

}  // namespace adaisp
