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
//   * consecutive y-shifts (dy, dy+1) are processed as packed fp32 pairs (v_pk_add_f32 / v_pk_mul_f32), which halves
//     the instruction count of the patch sums; the weight is exp2(v_sqrt(D) * c) on the hardware units.
//
// No barriers inside the 121-shift loop. Compute-bound on the fp32 VALU (≈3.3 kflop/px + 242
// transcendentals per px), not on HBM (24 B/px).
#include "isp_internal.h"
#ifndef NLM_FMA
#define NLM_FMA 1             // k_nlm_fwd: column sums as fused multiply-add chains (round 4: 453.7 -> 438.4 us, -3.4 %; 0 = mul + add)
#endif
#ifndef NLM_ABL
#define NLM_ABL 0             // measurement builds of k_nlm_fwd: 1 no transcendentals, 2 no row sums, 3 no colour sums, 4 exp2 only
#endif

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
constexpr int kYBytes = ((YROWS * YP * 8 + 15) / 16) * 16;
constexpr int kSmemBytes = kYBytes + 3 * CROWS * CP * 4;

typedef float v2f __attribute__((ext_vector_type(2)));

// (a.lo - b.lo, a.lo - b.hi) and (a.hi - b.lo, a.hi - b.hi): packed subtract with one half of `a` broadcast
// through the VOP3P operand selects (no v_mov to build the splat).
__device__ __forceinline__ v2f pk_sub_bcast_lo(v2f a, v2f b) {
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ v2f pk_sub_bcast_hi(v2f a, v2f b) {     // plain: see the note on high-half selects in k_nlm_fwd
    return v2f{a.y, a.y} - b;
}

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
    // LDS (dynamic, 66 KB): luminance as (y[row], y[row-1]) pairs so that the operands of two consecutive y-shifts
    // are ONE aligned ds_read_b64, then the three clamped colour planes
    extern __shared__ __attribute__((aligned(16))) unsigned char nlm_smem[];
    v2f* ylds2 = reinterpret_cast<v2f*>(nlm_smem);                           // [YROWS][YP]
    float* clds = reinterpret_cast<float*>(nlm_smem + kYBytes);              // [3][CROWS][CP]

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
        const float yv = (0.299f * r + 0.587f * gg) + 0.114f * bb;     // denoise.py:17
        reinterpret_cast<float*>(ylds2)[2 * q] = yv;                       // .x of row ly
        if (ly + 1 < YROWS) reinterpret_cast<float*>(ylds2)[2 * (q + YP) + 1] = yv;   // .y of row ly+1
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

    // centre luminances: 5 patch columns x 12 rows, fixed for all 121 shifts; kept as register pairs (rows 2j, 2j+1)
    v2f ycp[2 * PR + 1][NK / 2];
#pragma unroll
    for (int bi = 0; bi < 2 * PR + 1; ++bi)        // bi = bx + PR
#pragma unroll
        for (int j = 0; j < NK / 2; ++j) {
            const v2f* q = ylds2 + (rb + HY - PR + 2 * j) * YP + tx + HY - (bi - PR);
            ycp[bi][j] = v2f{q[0].x, q[YP].x};
        }

    float num[3][RPT], den[RPT];
    float ga[GRAD ? 3 : 1][GRAD ? RPT : 1], gb[GRAD ? RPT : 1];
#pragma unroll
    for (int r = 0; r < RPT; ++r) { num[0][r] = num[1][r] = num[2][r] = 0.0f; den[r] = 0.0f; }
    if (GRAD) {
#pragma unroll
        for (int r = 0; r < RPT; ++r) { ga[0][r] = ga[1][r] = ga[2][r] = 0.0f; gb[r] = 0.0f; }
    }
    const float inv_hh2 = 1.0f / (hh * hh);

    // exp(-dist/hh) = exp2(dist * nc): one multiply + the hardware exp2 (the relative error |x|*2^-24 only matters
    // for weights far too small to contribute); v_sqrt_f32 is within 1 ulp.
    const float nc = -1.44269504088896341f / hh;

    // One shift: patch distance for the 8 pixels in the reference's summation order, then the weighted sums.
    auto accumulate = [&](int r, float Dr, const float* cs) {
        const float dist = __builtin_amdgcn_sqrtf(fmaxf(Dr, 0.0f));
        const float wgt = __builtin_amdgcn_exp2f(dist * nc);
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
    };

    for (int dx = -SR; dx <= SR; ++dx) {           // x_shift outer   (denoise.py:104)
        // y shifts are visited in ascending order like the reference (denoise.py:105); two consecutive shifts
        // (dy, dy+1) share every centre luminance, so their squared differences and patch sums run as packed
        // fp32 pairs (v_pk_add/mul_f32: two IEEE results per instruction, same rounding as the scalar ops).
        for (int dy = -SR; dy + 1 <= SR; dy += 2) {
            const v2f* ys = ylds2 + (rb + HY - PR - dy) * YP + tx + HY - dx;      // .x: rows for shift dy, .y: one row up (dy+1)
            v2f D2[RPT];
#pragma unroll
            for (int r = 0; r < RPT; ++r) D2[r] = v2f{0.0f, 0.0f};
#pragma unroll
            for (int bi = 0; bi < 2 * PR + 1; ++bi) {      // patch column bx = bi - PR, outer (denoise.py:60)
                v2f sq2[NK];
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const v2f sh = ys[k * YP - (bi - PR)];                // (shift dy, shift dy+1) in one ds_read_b64
                    const v2f d = (k & 1) ? pk_sub_bcast_hi(ycp[bi][k >> 1], sh) : pk_sub_bcast_lo(ycp[bi][k >> 1], sh);
                    sq2[k] = d * d;
                }
#pragma unroll
                for (int byi = 0; byi < 2 * PR + 1; ++byi)  // patch row by = byi - PR, inner (denoise.py:61)
#pragma unroll
                    for (int r = 0; r < RPT; ++r) D2[r] += sq2[r + PR - (byi - PR)];
            }
            const float* cs0 = clds + (rb + SR - dy) * CP + tx + SR - dx;
#pragma unroll
            for (int r = 0; r < RPT; ++r) accumulate(r, D2[r].x, cs0);
#pragma unroll
            for (int r = 0; r < RPT; ++r) accumulate(r, D2[r].y, cs0 - CP);
        }
        {   // the eleventh shift (dy = +SR) has no partner
            const int dy = SR;
            const v2f* ys = ylds2 + (rb + HY - PR - dy) * YP + tx + HY - dx;
            float D[RPT];
#pragma unroll
            for (int r = 0; r < RPT; ++r) D[r] = 0.0f;
#pragma unroll
            for (int bi = 0; bi < 2 * PR + 1; ++bi) {
                float sq[NK];
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const float d = ycp[bi][k >> 1][k & 1] - ys[k * YP - (bi - PR)].x;
                    sq[k] = d * d;
                }
#pragma unroll
                for (int byi = 0; byi < 2 * PR + 1; ++byi)
#pragma unroll
                    for (int r = 0; r < RPT; ++r) D[r] += sq[r + PR - (byi - PR)];
            }
            const float* cs = clds + (rb + SR - dy) * CP + tx + SR - dx;
#pragma unroll
            for (int r = 0; r < RPT; ++r) accumulate(r, D[r], cs);
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


// ---------------------------------------------------------------------------------------------------------
// Separable variant (default). The 5x5 patch sum is a column sum (5 rows, in-lane, in the reference's patch-row
// order) followed by a row sum over the 5 neighbouring columns, which live in the neighbouring LANES: they are
// fetched with DPP wave shifts folded into the adds (v_add_f32_dpp wave_shl/shr), no LDS traffic. Per pixel and
// shift that is 2 + 4 + 4 flops instead of 50, and the lane only keeps its own 12 centre luminances (12 VGPRs
// instead of 60 -> 3+ waves/SIMD). The association of the 25-term sum differs from the reference's single running
// sum (5 partial sums of 5), i.e. ~1 ulp of the patch distance; measured against the oracle in the parity tests
// (rtol 1e-5). The reference-order kernel above stays available through ADAISP_NLM_EXACT.
// A wave covers 64 columns but only its inner 60 are complete (2 on each side lack neighbours): tiles advance by 60.
// ---------------------------------------------------------------------------------------------------------
constexpr int SW = 60;                          // valid columns per wave
constexpr int SP = 64 + 2 * SR;                 // staged columns: x0-2-5 .. x0+61+5  (74)
constexpr int kSepBytes = (YROWS * SP + 3 * CROWS * SP) * 4;

// (the wave's end lanes keep their own value: they belong to the 2+2 columns that are discarded anyway)
__device__ __forceinline__ float lane_up(float v) {     // value of lane+1
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x130, 0xF, 0xF, false));
}
__device__ __forceinline__ float lane_down(float v) {   // value of lane-1
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138, 0xF, 0xF, false));
}

template <bool GRAD>
__global__ __launch_bounds__(kThreads) void k_nlm_sep(const float* __restrict__ img, float* __restrict__ out,
                                                      const int32_t* __restrict__ ids, int uniform_op,
                                                      const float* __restrict__ params, int pstride, int H, int W,
                                                      const float* __restrict__ grad_out,
                                                      float* __restrict__ grad_params) {
    __shared__ float ylds[YROWS * SP];
    __shared__ float clds[3 * CROWS * SP];
    const int b = blockIdx.z;
    const int op = ids ? ids[b] : uniform_op;
    if (op != ADAISP_OP_NLM) return;
    const long plane = (long)H * W;
    const float* __restrict__ in = img + (long)b * 3 * plane;
    const int x0 = blockIdx.x * SW, y0 = blockIdx.y * TH;
    const int tid = threadIdx.x;

    // staged column c <-> image column x0 - 2 - SR + c ; y rows start at y0 - HY, colour rows at y0 - SR
    for (int q = tid; q < YROWS * SP; q += kThreads) {
        const int ly = q / SP, lx = q - ly * SP;
        const int gy = wrap(y0 + ly - HY, H), gx = wrap(x0 - 2 - SR + lx, W);
        const long g = (long)gy * W + gx;
        const float r = clamp01(in[g]), gg = clamp01(in[g + plane]), bb = clamp01(in[g + 2 * plane]);
        ylds[q] = (0.299f * r + 0.587f * gg) + 0.114f * bb;
        const int cy = ly - PR;
        if (cy >= 0 && cy < CROWS) {
            clds[(0 * CROWS + cy) * SP + lx] = r;
            clds[(1 * CROWS + cy) * SP + lx] = gg;
            clds[(2 * CROWS + cy) * SP + lx] = bb;
        }
    }
    __syncthreads();

    const int lane = tid & 63, ty = tid >> 6;
    const int rb = ty * RPT;
    const float hh = fmaxf(params[(long)b * pstride], 0.0f) + 1e-8f;
    const float nc = -1.44269504088896341f / hh;
    const float inv_hh2 = 1.0f / (hh * hh);

    float yc[NK];                                  // this lane's own column, rows rb-2 .. rb+9
#pragma unroll
    for (int k = 0; k < NK; ++k) yc[k] = ylds[(rb + HY - PR + k) * SP + lane + SR];

    v2f num2[3][RPT / 2], den2[RPT / 2];
    v2f ga2[GRAD ? 3 : 1][GRAD ? RPT / 2 : 1], gb2[GRAD ? RPT / 2 : 1];
#pragma unroll
    for (int r = 0; r < RPT / 2; ++r) { num2[0][r] = num2[1][r] = num2[2][r] = den2[r] = v2f{0.0f, 0.0f}; }
    if (GRAD) {
#pragma unroll
        for (int r = 0; r < RPT / 2; ++r) { ga2[0][r] = ga2[1][r] = ga2[2][r] = gb2[r] = v2f{0.0f, 0.0f}; }
    }

    for (int dx = -SR; dx <= SR; ++dx) {
        for (int dy = -SR; dy <= SR; ++dy) {
            const float* ys = ylds + (rb + HY - PR - dy) * SP + lane + SR - dx;
            float sq[NK];
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const float d = yc[k] - ys[k * SP];
                sq[k] = d * d;
            }
            const float* cs = clds + (rb + SR - dy) * SP + lane + SR - dx;
            // pixels are finished in row pairs so that the weighting runs on packed fp32 (v_pk_mul/fma_f32)
#pragma unroll
            for (int rp = 0; rp < RPT / 2; ++rp) {
                v2f D2;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int r = 2 * rp + e;
                    // column sum, patch rows by = -2..2  <->  sq[r+4] .. sq[r]
                    const float c = (((sq[r + 4] + sq[r + 3]) + sq[r + 2]) + sq[r + 1]) + sq[r];
                    // row sum, patch columns bx = -2..2  <->  lanes +2 .. -2
                    const float u1 = lane_up(c), u2 = lane_up(u1), d1 = lane_down(c), d2 = lane_down(d1);
                    D2[e] = (((u2 + u1) + c) + d1) + d2;
                }
                D2 = __builtin_elementwise_max(D2, v2f{0.0f, 0.0f});
                const v2f dist = v2f{__builtin_amdgcn_sqrtf(D2.x), __builtin_amdgcn_sqrtf(D2.y)};
                const v2f ex = dist * nc;
                const v2f wgt = v2f{__builtin_amdgcn_exp2f(ex.x), __builtin_amdgcn_exp2f(ex.y)};
                const int r = 2 * rp;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const v2f cv = v2f{cs[(c * CROWS + r) * SP], cs[(c * CROWS + r + 1) * SP]};
                    num2[c][rp] = __builtin_elementwise_fma(cv, wgt, num2[c][rp]);
                    if (GRAD) ga2[c][rp] = __builtin_elementwise_fma(cv, wgt * dist * inv_hh2, ga2[c][rp]);
                }
                den2[rp] += wgt;
                if (GRAD) gb2[rp] += wgt * dist * inv_hh2;
            }
        }
    }
    float num[3][RPT], den[RPT];
    float ga[GRAD ? 3 : 1][GRAD ? RPT : 1], gb[GRAD ? RPT : 1];
#pragma unroll
    for (int rp = 0; rp < RPT / 2; ++rp)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            den[2 * rp + e] = den2[rp][e];
#pragma unroll
            for (int c = 0; c < 3; ++c) num[c][2 * rp + e] = num2[c][rp][e];
            if (GRAD) {
                gb[2 * rp + e] = gb2[rp][e];
#pragma unroll
                for (int c = 0; c < 3; ++c) ga[c][2 * rp + e] = ga2[c][rp][e];
            }
        }

    const int gx = x0 - 2 + lane;
    const bool mine = lane >= 2 && lane < 2 + SW && gx < W;
    if (!GRAD) {
        float* __restrict__ o = out + (long)b * 3 * plane;
        if (mine) {
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
        if (mine && params[(long)b * pstride] > 0.0f) {
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

// ---------------------------------------------------------------------------------------------------------
// Forward kernel (default): the separable scheme above with the instruction stream laid out by hand. The compiled
// k_nlm_sep<false> loop spends 74 of its ~200 VALU instructions per shift on moves (v_mov_b32_dpp for every
// neighbour fetch, v_mov to assemble the operands of the packed adds). Here:
//   * a lane owns rows (r, r+16) of its column, r = 4*wave + 0..3, and LDS holds the luminance (and blue) of rows
//     a and a+16 as one 8-byte pair: ONE ds_read_b64 delivers the operands of a packed instruction, so the squared
//     differences, the 5-row column sums, the weights and the blue sums are packed fp32 from the LDS read on —
//     8 sub + 4 mul + 16 fma/add for 8 pixels, no operand assembly, no overlap between the two halves;
//   * the 5-lane row sum is four v_add_f32_dpp per value (the shift is an operand modifier of the add, no move):
//     t1 = c[i+1] + c[i], t2 = t1[i+1] + c[i], P = c[i-1] + c[i], D = P[i-1] + t2 — issued as one block of 32 so that
//     no DPP operand was written by the two preceding instructions (the DPP read hazard needs no s_nop);
//   * red and green are interleaved in LDS: one ds_read_b64 per pixel, accumulated as (R,G) x (w,w); every LDS read
//     uses an immediate offset from one base register.
// ~100 VALU + 20 LDS instructions per shift for 8 pixels. Same association of the 25-term sum up to the order of the
// five column sums (~1 ulp of the patch distance; parity rtol 1e-5 against the oracle).
// ---------------------------------------------------------------------------------------------------------
// RQ = row pairs per lane: 4 -> 32-row tile, 58 KB of LDS, 2 workgroups per CU; 3 -> 24-row tile, 48.5 KB, THREE
// workgroups per CU (the kernel issues ~65 % of the time with two waves per SIMD; the third wave is worth more than the
// 5-row column sums it re-computes: 7 squared differences per 3 outputs instead of 8 per 4). 720 = 30 x 24 and
// 2160 = 90 x 24: no ragged tile row at the BASELINE sizes. Measured in one process (tools/nlm_ab.py), bit-identical
// outputs: 8x720x1280 0.488 vs 0.538 ms, 4x2160x3840 1.904 vs 2.154 ms; a 16-row tile (4 per CU) 0.478 / 2.005 ms: no better.
template <int RQ>
__global__ __launch_bounds__(kThreads) void k_nlm_fwd(const float* __restrict__ img, float* __restrict__ out,
                                                      const int32_t* __restrict__ ids, int uniform_op,
                                                      const float* __restrict__ params, int pstride, int H, int W) {
    constexpr int TH = 8 * RQ, HALF = TH / 2;       // HALF: distance between the two rows of a lane's pair
    constexpr int YROWS = TH + 2 * HY, CROWS = TH + 2 * SR;
    constexpr int Y2ROWS = YROWS - HALF;            // luminance pairs (a, a + HALF)
    constexpr int B2ROWS = CROWS - HALF;            // blue pairs
    __shared__ v2f y2[Y2ROWS * SP];
    __shared__ v2f crg[CROWS * SP];
    __shared__ v2f cb2[B2ROWS * SP];
    const int b = blockIdx.z;
    const int op = ids ? ids[b] : uniform_op;
    if (op != ADAISP_OP_NLM) return;
    const long plane = (long)H * W;
    const float* __restrict__ in = img + (long)b * 3 * plane;
    const int x0 = blockIdx.x * SW, y0 = blockIdx.y * TH;
    const int tid = threadIdx.x;

    float* y2f = reinterpret_cast<float*>(y2);
    float* cb2f = reinterpret_cast<float*>(cb2);
    // staging: 4 positions = 12 loads in flight per thread (one position per trip is 14 serial memory round trips)
    for (int q0 = tid; q0 < YROWS * SP; q0 += 4 * kThreads) {
        float rr[4], gg4[4], bb4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = min(q0 + i * kThreads, YROWS * SP - 1);
            const int ly = q / SP, lx = q - ly * SP;
            const int gy = wrap(y0 + ly - HY, H), gx = wrap(x0 - 2 - SR + lx, W);
            const long g = (long)gy * W + gx;
            rr[i] = in[g]; gg4[i] = in[g + plane]; bb4[i] = in[g + 2 * plane];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = q0 + i * kThreads;
            if (q >= YROWS * SP) break;
            const int ly = q / SP, lx = q - ly * SP;
            const float r = clamp01(rr[i]), gg = clamp01(gg4[i]), bb = clamp01(bb4[i]);
            const float yv = (0.299f * r + 0.587f * gg) + 0.114f * bb;
            if (ly < Y2ROWS) y2f[(ly * SP + lx) * 2] = yv;
            if (ly >= HALF) y2f[((ly - HALF) * SP + lx) * 2 + 1] = yv;
            const int cy = ly - PR;
            if (cy >= 0 && cy < CROWS) {
                crg[cy * SP + lx] = v2f{r, gg};
                if (cy < B2ROWS) cb2f[(cy * SP + lx) * 2] = bb;
                if (cy >= HALF) cb2f[((cy - HALF) * SP + lx) * 2 + 1] = bb;
            }
        }
    }
    __syncthreads();

    const int lane = tid & 63, ty = tid >> 6;
    const int rb = ty * RQ;                         // rows rb .. rb+3 and rb+16 .. rb+19 of the tile
    const float hh = fmaxf(params[(long)b * pstride], 0.0f) + 1e-8f;
    const float nc = -1.44269504088896341f / hh;

    v2f yc2[RQ + 2 * PR];                           // own column, rows rb-2 .. rb+5 (and +16)
#pragma unroll
    for (int j = 0; j < RQ + 2 * PR; ++j) yc2[j] = y2[(rb + HY - PR + j) * SP + lane + SR];

    v2f nrgA[RQ], nrgB[RQ], nb2[RQ], den2[RQ];
#pragma unroll
    for (int i = 0; i < RQ; ++i) nrgA[i] = nrgB[i] = nb2[i] = den2[i] = v2f{0.0f, 0.0f};

    for (int dx = -SR; dx <= SR; ++dx) {
        for (int dy = -SR; dy <= SR; ++dy) {        // (unrolling the 11 y-shifts was measured: no gain)
            const v2f* ys = y2 + (rb + HY - PR - dy) * SP + lane + SR - dx;
            float c[2 * RQ];
#if NLM_FMA
            // 5-row column sums as fused chains: d4^2, then fma(d3, d3, .) ... fma(d0, d0, .) — the same order of the five terms as
            // the mul + add form below, one rounding per term instead of two: 7 sub + 3 x (1 mul + 4 fma) = 22 packed instructions
            // per shift instead of 7 + 7 + 12 = 26 (the squares are recomputed instead of shared)
            v2f dd[RQ + 2 * PR];
#pragma unroll
            for (int j = 0; j < RQ + 2 * PR; ++j) dd[j] = yc2[j] - ys[j * SP];
#pragma unroll
            for (int i = 0; i < RQ; ++i) {
                v2f cc = dd[i + 4] * dd[i + 4];
                cc = __builtin_elementwise_fma(dd[i + 3], dd[i + 3], cc);
                cc = __builtin_elementwise_fma(dd[i + 2], dd[i + 2], cc);
                cc = __builtin_elementwise_fma(dd[i + 1], dd[i + 1], cc);
                cc = __builtin_elementwise_fma(dd[i], dd[i], cc);
                c[2 * i] = cc.x;
                c[2 * i + 1] = cc.y;
            }
#else
            v2f S[RQ + 2 * PR];
#pragma unroll
            for (int j = 0; j < RQ + 2 * PR; ++j) {
                const v2f d = yc2[j] - ys[j * SP];
                S[j] = d * d;
            }
#pragma unroll
            for (int i = 0; i < RQ; ++i) {
                const v2f cc = (((S[i + 4] + S[i + 3]) + S[i + 2]) + S[i + 1]) + S[i];
                c[2 * i] = cc.x;
                c[2 * i + 1] = cc.y;
            }
#endif
            float D[2 * RQ], t1[2 * RQ], t2[2 * RQ], P[2 * RQ];
#if NLM_ABL == 2          // measurement build: no 5-lane row sums
#pragma unroll
            for (int i = 0; i < 2 * RQ; ++i) D[i] = c[i];
            if (false)
#endif
            // t1 = c[i+1] + c[i], P = c[i-1] + c[i], t2 = t1[i+1] + c[i], D = P[i-1] + t2 (see above); one block per RQ
            if constexpr (RQ == 4) {
                asm("s_nop 1\n\t"
                    "v_add_f32_dpp %8, %32, %32 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %9, %33, %33 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %10, %34, %34 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %11, %35, %35 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %12, %36, %36 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %13, %37, %37 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %14, %38, %38 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %15, %39, %39 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %24, %32, %32 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %25, %33, %33 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %26, %34, %34 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %27, %35, %35 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %28, %36, %36 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %29, %37, %37 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %30, %38, %38 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %31, %39, %39 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %16, %8, %32 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %17, %9, %33 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %18, %10, %34 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %19, %11, %35 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %20, %12, %36 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %21, %13, %37 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %22, %14, %38 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %23, %15, %39 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %0, %24, %16 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %1, %25, %17 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %2, %26, %18 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %3, %27, %19 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %4, %28, %20 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %5, %29, %21 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %6, %30, %22 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %7, %31, %23 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
                    : "=&v"(D[0]), "=&v"(D[1]), "=&v"(D[2]), "=&v"(D[3]), "=&v"(D[4]), "=&v"(D[5]), "=&v"(D[6]), "=&v"(D[7]),
                      "=&v"(t1[0]), "=&v"(t1[1]), "=&v"(t1[2]), "=&v"(t1[3]), "=&v"(t1[4]), "=&v"(t1[5]), "=&v"(t1[6]), "=&v"(t1[7]),
                      "=&v"(t2[0]), "=&v"(t2[1]), "=&v"(t2[2]), "=&v"(t2[3]), "=&v"(t2[4]), "=&v"(t2[5]), "=&v"(t2[6]), "=&v"(t2[7]),
                      "=&v"(P[0]), "=&v"(P[1]), "=&v"(P[2]), "=&v"(P[3]), "=&v"(P[4]), "=&v"(P[5]), "=&v"(P[6]), "=&v"(P[7])
                    : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]), "v"(c[7]));
            } else {
                asm("s_nop 1\n\t"
                    "v_add_f32_dpp %6, %24, %24 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %7, %25, %25 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %8, %26, %26 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %9, %27, %27 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %10, %28, %28 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %11, %29, %29 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %18, %24, %24 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %19, %25, %25 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %20, %26, %26 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %21, %27, %27 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %22, %28, %28 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %23, %29, %29 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %12, %6, %24 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %13, %7, %25 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %14, %8, %26 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %15, %9, %27 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %16, %10, %28 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %17, %11, %29 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %0, %18, %12 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %1, %19, %13 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %2, %20, %14 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %3, %21, %15 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %4, %22, %16 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32_dpp %5, %23, %17 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
                    : "=&v"(D[0]), "=&v"(D[1]), "=&v"(D[2]), "=&v"(D[3]), "=&v"(D[4]), "=&v"(D[5]),
                      "=&v"(t1[0]), "=&v"(t1[1]), "=&v"(t1[2]), "=&v"(t1[3]), "=&v"(t1[4]), "=&v"(t1[5]),
                      "=&v"(t2[0]), "=&v"(t2[1]), "=&v"(t2[2]), "=&v"(t2[3]), "=&v"(t2[4]), "=&v"(t2[5]),
                      "=&v"(P[0]), "=&v"(P[1]), "=&v"(P[2]), "=&v"(P[3]), "=&v"(P[4]), "=&v"(P[5])
                    : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]));
            }
            const v2f* rg = crg + (rb + SR - dy) * SP + lane + SR - dx;
            const v2f* bl = cb2 + (rb + SR - dy) * SP + lane + SR - dx;
#pragma unroll
            for (int i = 0; i < RQ; ++i) {
                // D >= 0 (a sum of squares): no clamp before the square root
#if NLM_ABL == 1          // measurement build: no transcendentals
                const v2f wgt = v2f{D[2 * i], D[2 * i + 1]} * nc + 1.0f;
#elif NLM_ABL == 4        // measurement build: exp2 only
                const v2f ex = v2f{D[2 * i], D[2 * i + 1]} * nc;
                const v2f wgt = v2f{__builtin_amdgcn_exp2f(ex.x), __builtin_amdgcn_exp2f(ex.y)};
#else
                const v2f dist = v2f{__builtin_amdgcn_sqrtf(D[2 * i]), __builtin_amdgcn_sqrtf(D[2 * i + 1])};
                const v2f ex = dist * nc;
                const v2f wgt = v2f{__builtin_amdgcn_exp2f(ex.x), __builtin_amdgcn_exp2f(ex.y)};
#endif
#if NLM_ABL == 3          // measurement build: no colour accumulation (weights summed only)
                den2[i] += wgt;
                continue;
#endif
                // (R,G) x (w,w): left to the compiler, which broadcasts the LOW half through op_sel_hi and moves a
                // high half down first. A hand-written high-half broadcast (op_sel:[0,1,0]) on v_pk_fma_f32 passed every
                // stand-alone test and gave run-to-run different sums beside the detector's workgroups
                // (tools/trunk_stress.py) — not used anywhere in this library.
                nrgA[i] = __builtin_elementwise_fma(rg[i * SP], v2f{wgt.x, wgt.x}, nrgA[i]);
                nrgB[i] = __builtin_elementwise_fma(rg[(i + HALF) * SP], v2f{wgt.y, wgt.y}, nrgB[i]);
                nb2[i] = __builtin_elementwise_fma(bl[i * SP], wgt, nb2[i]);
                den2[i] += wgt;
            }
        }
    }

    const int gx = x0 - 2 + lane;
    const bool mine = lane >= 2 && lane < 2 + SW && gx < W;
    float* __restrict__ o = out + (long)b * 3 * plane;
    if (mine) {
#pragma unroll
        for (int i = 0; i < RQ; ++i)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int gy = y0 + rb + i + e * HALF;
                if (gy < H) {
                    const long g = (long)gy * W + gx;
                    const float den = den2[i][e];
                    const v2f n2 = e ? nrgB[i] : nrgA[i];
                    o[g] = clamp01(n2.x / den);
                    o[g + plane] = clamp01(n2.y / den);
                    o[g + 2 * plane] = clamp01(nb2[i][e] / den);
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------
// NonLocalMeansGray(search_window_size, patch_size) for ANY odd sizes (isp/denoise.py:93-119; the class default is
// 21 / 7, the ISP's DenoiseFilter uses 11 / 5 and has the tuned kernels above). A plain gather form, one pixel per
// lane: the clipped luminance comes from a plane computed once (k_nlm_luma, denoise.py:11-17), the patch distance is summed
// in the reference's order (patch column outer, patch row inner, from 0), shifts are visited x outer / y inner, the
// colours are the UNCLIPPED input (the class clips only inside rgb_to_luminance), the result is clamped to [0, 1].
// O(search^2 * patch^2) cached loads per pixel: a correctness path for configurations the ISP does not use.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_nlm_luma(const float* __restrict__ img, float* __restrict__ y, long plane, long total) {
    const long i = (long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= total) return;
    const long b = i / plane, r = i - b * plane;
    const float* in = img + b * 3 * plane + r;
    const float rr = clamp01(in[0]), gg = clamp01(in[plane]), bb = clamp01(in[2 * plane]);
    y[i] = (0.299f * rr + 0.587f * gg) + 0.114f * bb;
}

__global__ __launch_bounds__(kThreads) void k_nlm_general(const float* __restrict__ img, const float* __restrict__ ylum,
                                                          float* __restrict__ out, const float* __restrict__ h, int hstride,
                                                          int H, int W, int SRg, int PRg) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), yy = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
    if (x >= W || yy >= H) return;
    const long plane = (long)H * W;
    const float* in = img + (long)b * 3 * plane;
    const float* yl = ylum + (long)b * plane;
    const float hh = fmaxf(h[(long)b * hstride], 0.0f) + 1e-8f;
    float n0 = 0.f, n1 = 0.f, n2 = 0.f, den = 0.f;
    for (int dx = -SRg; dx <= SRg; ++dx)
        for (int dy = -SRg; dy <= SRg; ++dy) {
            float D = 0.f;
            for (int bx = -PRg; bx <= PRg; ++bx) {
                const int jj = wrap(x - bx, W), js = wrap(x - bx - dx, W);
                for (int by = -PRg; by <= PRg; ++by) {
                    const int ii = wrap(yy - by, H), is = wrap(yy - by - dy, H);
                    const float d = yl[(long)ii * W + jj] - yl[(long)is * W + js];
                    D += d * d;
                }
            }
            const float wgt = expf(-sqrtf(fmaxf(D, 0.0f)) / hh);
            const long sidx = (long)wrap(yy - dy, H) * W + wrap(x - dx, W);
            n0 += in[sidx] * wgt; n1 += in[sidx + plane] * wgt; n2 += in[sidx + 2 * plane] * wgt;
            den += wgt;
        }
    float* o = out + (long)b * 3 * plane + (long)yy * W + x;
    o[0] = clamp01(n0 / den); o[plane] = clamp01(n1 / den); o[2 * plane] = clamp01(n2 / den);
}

}  // namespace

hipError_t launch_nlm_general(const float* img, float* out, const float* h, int hstride, float* workspace, int B, int H, int W,
                              int search, int patch, hipStream_t s) {
    const long plane = (long)H * W, total = plane * B;
    hipLaunchKernelGGL(k_nlm_luma, dim3((unsigned)((total + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, img, workspace, plane, total);
    hipLaunchKernelGGL(k_nlm_general, dim3((W + 63) / 64, (H + 3) / 4, B), dim3(kThreads), 0, s, img, workspace, out, h, hstride, H, W,
                       search / 2, patch / 2);
    return hipGetLastError();
}

hipError_t launch_nlm(const Batch& a, hipStream_t s) {
    if (!(a.flags & ADAISP_NLM_EXACT)) {
        dim3 g((a.W + SW - 1) / SW, (a.H + TH - 1) / TH, a.B);
        if (a.flags & ADAISP_NLM_SEP_V1)                 // the compiler-scheduled form of the same scheme (A/B, tests)
            hipLaunchKernelGGL(k_nlm_sep<false>, g, dim3(kThreads), 0, s, a.img, a.out, a.ids, a.uniform_op, a.params,
                               a.pstride, a.H, a.W, static_cast<const float*>(nullptr), static_cast<float*>(nullptr));
        else if (a.flags & ADAISP_NLM_TILE32)            // the 32-row tile (2 workgroups per CU): A/B, tests
            hipLaunchKernelGGL(k_nlm_fwd<4>, g, dim3(kThreads), 0, s, a.img, a.out, a.ids, a.uniform_op, a.params, a.pstride,
                               a.H, a.W);
        else
            hipLaunchKernelGGL(k_nlm_fwd<3>, dim3(g.x, (a.H + 23) / 24, a.B), dim3(kThreads), 0, s, a.img, a.out, a.ids,
                               a.uniform_op, a.params, a.pstride, a.H, a.W);
        return hipGetLastError();
    }
    dim3 grid((a.W + TW - 1) / TW, (a.H + TH - 1) / TH, a.B);
    static bool configured = false;          // dynamic LDS above the 64 KB default (idempotent)
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_nlm<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kSmemBytes);
        if (e != hipSuccess) return e;
        configured = true;
    }
    hipLaunchKernelGGL(k_nlm<false>, grid, dim3(kThreads), kSmemBytes, s, a.img, a.out, a.ids, a.uniform_op, a.params, a.pstride,
                       a.H, a.W, static_cast<const float*>(nullptr), static_cast<float*>(nullptr));
    return hipGetLastError();
}

hipError_t launch_nlm_backward(const float* img, const float* grad_out, const int32_t* ids, const float* params,
                               int pstride, float* grad_params, int B, int H, int W, unsigned flags, hipStream_t s) {
    if (!(flags & ADAISP_NLM_EXACT)) {
        dim3 g((W + SW - 1) / SW, (H + TH - 1) / TH, B);
        hipLaunchKernelGGL(k_nlm_sep<true>, g, dim3(kThreads), 0, s, img, static_cast<float*>(nullptr), ids, 0, params,
                           pstride, H, W, grad_out, grad_params);
        return hipGetLastError();
    }
    dim3 grid((W + TW - 1) / TW, (H + TH - 1) / TH, B);
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_nlm<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kSmemBytes);
        if (e != hipSuccess) return e;
        configured = true;
    }
    hipLaunchKernelGGL(k_nlm<true>, grid, dim3(kThreads), kSmemBytes, s, img, static_cast<float*>(nullptr), ids, 0, params,
                       pstride, H, W, grad_out, grad_params);
    return hipGetLastError();
}

}  // namespace adaisp
