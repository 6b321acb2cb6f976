// Fused eval path of the policy step (Agent.forward, agent.py:88-285) for gfx950.
//
// In the reference one RL step of the policy is ~250 tiny ATen launches per batch (two 4-layer CNN trunks
// with BatchNorm, 10 filters x (fc1, LeakyReLU, fc_filter, fc_mask, regressor), selector head, softmax,
// sampling, one-hot, state update, penalties). At 64x64 inputs that is pure launch latency. Here:
//
//   k_trunk_conv  x4  Conv2d(k4,s2,p1) with the BatchNorm folded in + LeakyReLU(0.2), fp32. Both trunks
//                     (parameter features / action selection) run in one launch (blockIdx.z). The first
//                     layer assembles its input on the fly: 3 pooled image planes + the state vector as
//                     constant planes (enrich_image_input, util.py:58-63) — no concat tensor.
//                     Weights are wave-uniform (scalar loads), each lane owns one output pixel x 8 channels.
//   k_fc1         x1  the eleven 4096->128 hidden layers (10 filter heads + selector), one wave per neuron,
//                     weights streamed once for the whole batch.
//   k_finish      x1  per image: fc_filter of every head + its regressor (tanh_range / exp / sigmoid / WB
//                     normalisation), selector fc2 + softmax + exploration mix + entropy, pdf_sample / argmax /
//                     forced id, one-hot, surrogate, state update, penalty, and the packed parameter row + op
//                     code that adaisp_forward consumes. Integer stages follow agent.py:12-23,138-149,234-259.
//
// All arithmetic is fp32 with fmaf accumulation (the reference's fp32 GEMM/conv libraries use other
// summation orders; parity is to 1e-5, selection is exact on the golden vectors).
#include "isp_internal.h"

namespace adaisp {
namespace {

__device__ __forceinline__ float lrelu02(float v) { return v > 0.0f ? v : 0.2f * v; }

// ---- trunk conv: out[g][b][co][oy][ox] = lrelu(bias + sum_{ci,kh,kw} in[.., 2oy-1+kh, 2ox-1+kw] * w[g][co][ci][kh][kw])
// Workgroup = KS waves: 64 output pixels x 8 output channels, the input channels split into KS slices (one
// wave per slice, so the slice's weights stay wave-uniform scalar loads); partial sums meet in LDS.
// The deep layers have few pixels (4x4 maps) and long reductions (K = 2048): without the split they run
// on a handful of waves, each grinding a 16k-FMA serial chain.
constexpr int CO_PER = 8;
constexpr int PX_PER = 64;

__global__ __launch_bounds__(1024) void k_trunk_conv(const float* __restrict__ in, const float* __restrict__ states,
                                                     int n_state, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ out, int B,
                                                     int Cin, int Hin, int Cout, int KS) {
    // LDS: this workgroup's weights [Cin][CO_PER][16] (staged once with coalesced loads, then read back as
    // wave-uniform broadcasts), later reused for the partial sums [KS][CO_PER][PX_PER]
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int Ho = Hin >> 1;
    const int g = blockIdx.z, co0 = blockIdx.y * CO_PER;
    const int lane = threadIdx.x & 63, ks = threadIdx.x >> 6;
    const int idx = blockIdx.x * PX_PER + lane;
    const int npix = B * Ho * Ho;
    const bool live = idx < npix;
    const int pid = live ? idx : 0;
    const int b = pid / (Ho * Ho), r = pid - b * Ho * Ho, oy = r / Ho, ox = r - oy * Ho;
    const float* wg = w + ((long)g * Cout + co0) * Cin * 16;
    for (int i = threadIdx.x; i < Cin * CO_PER * 16; i += blockDim.x) {
        const int c = i / (Cin * 16), rem = i - c * (Cin * 16);      // global order [c][ci][t] -> LDS [ci][c][t]
        const int ci = rem >> 4, t = rem & 15;
        lds[(ci * CO_PER + c) * 16 + t] = wg[i];
    }
    __syncthreads();
    float acc[CO_PER];
#pragma unroll
    for (int c = 0; c < CO_PER; ++c) acc[c] = 0.0f;
    bool vy[4], vx[4];                                          // zero padding 1
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        vy[t] = (2 * oy - 1 + t) >= 0 && (2 * oy - 1 + t) < Hin;
        vx[t] = (2 * ox - 1 + t) >= 0 && (2 * ox - 1 + t) < Hin;
    }
    const int n_img = states ? 3 : Cin;                         // channels that come from a tensor
    const float* ib = states ? in + (long)b * 3 * Hin * Hin     // first layer: shared pooled image [B,3,H,H]
                             : in + ((long)g * B + b) * Cin * Hin * Hin;
    const int cps = (Cin + KS - 1) / KS;                        // channels per slice
    const int c_lo = ks * cps, c_hi = min(Cin, c_lo + cps);
    // four input channels per trip: all 64 window loads are issued before the first FMA needs one, so a wave pays the
    // global-memory latency twice (8 channels per slice) instead of eight times
    constexpr int CU = 2;
    for (int cb = c_lo; cb < c_hi; cb += CU) {
        float v[CU][16];
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            const int ci = cb + u;
            const bool cok = ci < c_hi;
            if (ci < n_img) {
                const float* p = ib + (long)(cok ? ci : c_lo) * Hin * Hin + (2 * oy - 1) * Hin + (2 * ox - 1);
#pragma unroll
                for (int kh = 0; kh < 4; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 4; ++kw) v[u][kh * 4 + kw] = (cok && vy[kh] && vx[kw]) ? p[kh * Hin + kw] : 0.0f;
            } else {
                const float s = cok ? states[b * n_state + (ci - 3)] : 0.0f;   // constant plane, zero outside the frame
#pragma unroll
                for (int kh = 0; kh < 4; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 4; ++kw) v[u][kh * 4 + kw] = (vy[kh] && vx[kw]) ? s : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            const int ci = min(cb + u, Cin - 1);                  // (values are zero when cb+u is past the slice)
            const float4* wl = reinterpret_cast<const float4*>(lds + ci * CO_PER * 16);
#pragma unroll
            for (int c = 0; c < CO_PER; ++c)
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    const float4 wv = wl[c * 4 + t4];             // same address in every lane: LDS broadcast
                    acc[c] = fmaf(v[u][4 * t4 + 0], wv.x, acc[c]);
                    acc[c] = fmaf(v[u][4 * t4 + 1], wv.y, acc[c]);
                    acc[c] = fmaf(v[u][4 * t4 + 2], wv.z, acc[c]);
                    acc[c] = fmaf(v[u][4 * t4 + 3], wv.w, acc[c]);
                }
        }
    }
    __syncthreads();                                            // weights no longer needed: reuse LDS for the partials
    float* part = lds;
#pragma unroll
    for (int c = 0; c < CO_PER; ++c) part[(ks * CO_PER + c) * PX_PER + lane] = acc[c];
    __syncthreads();
    // wave c of the first CO_PER waves finishes channel c (KS < CO_PER: the remaining channels loop)
    for (int c = ks; c < CO_PER; c += KS) {
        float v = bias[g * Cout + co0 + c];
        for (int k = 0; k < KS; ++k) v += part[(k * CO_PER + c) * PX_PER + lane];
        if (live) out[(((long)g * B + b) * Cout + co0 + c) * Ho * Ho + oy * Ho + ox] = lrelu02(v);
    }
}

// ---- trunk conv on the fp32 matrix cores ----------------------------------------------------------------------
// The same layer as an implicit GEMM D[co][px] = W[co][k] . X[k][px], k = ci*16 + kh*4 + kw, on
// v_mfma_f32_16x16x4_f32 (fp32 products, fp32 accumulation). A wave owns a 16 co x 16 px tile and a slice of the
// input channels; a lane feeds one weight row (A) and one pixel column (B). One input channel = 16 k = four MFMAs:
// lane quarter q = lane >> 4 owns kernel row kh = q and loads its four taps — a float4 of the weight row and four
// neighbouring input pixels — and MFMA j pairs tap kw = j of all four quarters, so neither operand passes through
// LDS. Up to 16 waves split the input channels (8 per wave per round) and meet in LDS, summed in slice order
// (deterministic). The 16x16 tile is what spreads the deep layers over the chip: the last layer is 128 px x 256 co
// = 256 tiles (64 with 32x32 tiles, whose K = 2048 then serialises 16k matrix cycles on one CU); every layer of the
// trunk comes to ~4k matrix cycles per CU. Weights and, between two RL steps, the activations come from HBM or
// another XCD's writes (~1-2 us per dependent access), so a wave issues ALL loads of a round before its first MFMA.
// (k_trunk_conv — 64 x 8 scalar-weight FMAs per lane, 64 KB of staged weights per workgroup — is kept for the shapes
// this form does not take.)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
constexpr int TM_U = 8;                                         // input channels per round (all loads before the first MFMA)

__global__ __launch_bounds__(1024) void k_trunk_mfma(const float* __restrict__ in, const float* __restrict__ states,
                                                     int n_state, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ out, int B,
                                                     int Cin, int Hin, int Cout, int KS) {
    __shared__ float part[15 * 4 * 64];                         // slices 1..15 (slice 0 keeps its registers)
    const int Ho = Hin >> 1, HW = Hin * Hin;
    const int g = blockIdx.z, co0 = blockIdx.y * 16;
    const int lane = threadIdx.x & 63, ks = threadIdx.x >> 6;
    const int col = lane & 15, q = lane >> 4;
    const int npix = B * Ho * Ho;
    const int idx = blockIdx.x * 16 + col;
    const bool live = idx < npix;
    const int pid = live ? idx : npix - 1;
    const int b = pid / (Ho * Ho), r = pid - b * Ho * Ho, oy = r / Ho, ox = r - oy * Ho;
    const int n_img = states ? 3 : Cin;
    const float* ib = states ? in + (long)b * 3 * HW : in + ((long)g * B + b) * Cin * HW;
    const float* wrow = w + ((long)g * Cout + co0 + col) * Cin * 16 + 4 * q;
    // this lane's kernel row kh = q and its four taps: clamped addresses (every load is legal and unconditional), the
    // zero padding is a multiply by 0/1 afterwards (a select would be turned into a branch around the load + vmcnt(0))
    const int iy = 2 * oy - 1 + q, ix0 = 2 * ox - 1;
    const float fy = (iy >= 0 && iy < Hin) ? 1.0f : 0.0f;
    const int rowoff = min(max(iy, 0), Hin - 1) * Hin;
    int cx[4];
    float fm[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        cx[t] = rowoff + min(max(ix0 + t, 0), Hin - 1);
        fm[t] = (ix0 + t >= 0 && ix0 + t < Hin) ? fy : 0.0f;
    }
    // a state channel (first layer, ci >= 3) is a constant plane: the same scalar for every tap inside the frame
    // (enrich_image_input, util.py:58-63)
    const float* sb = states ? states + b * n_state - 3 : in;
    const int cpw = Cin / KS;                                   // input channels per wave, a multiple of TM_U (launcher)
    f32x4_t acc = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int c0 = ks * cpw; c0 < (ks + 1) * cpw; c0 += TM_U) {
        float4 wa[TM_U];
        float xb[TM_U][4];
#pragma unroll
        for (int u = 0; u < TM_U; ++u) {
            const int ci = c0 + u;
            wa[u] = *reinterpret_cast<const float4*>(wrow + ci * 16);
            const bool plane = ci < n_img;
            const float* p = plane ? ib + (long)ci * HW : sb + ci;
#pragma unroll
            for (int t = 0; t < 4; ++t) xb[u][t] = p[plane ? cx[t] : 0] * fm[t];
        }
#pragma unroll
        for (int u = 0; u < TM_U; ++u) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[u].x, xb[u][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[u].y, xb[u][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[u].z, xb[u][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[u].w, xb[u][3], acc, 0, 0, 0);
        }
    }
    if (KS > 1) {
        if (ks != 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) part[((ks - 1) * 4 + e) * 64 + lane] = acc[e];
        }
        __syncthreads();
        if (ks != 0) return;
        for (int k = 1; k < KS; ++k)                            // fixed order: deterministic
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += part[((k - 1) * 4 + e) * 64 + lane];
    }
    if (!live) return;
    // D[row = co][col = px]: lane holds px = col and rows 4 * q + e
    float* ob = out + (((long)g * B + b) * Cout + co0 + 4 * q) * Ho * Ho + r;
#pragma unroll
    for (int e = 0; e < 4; ++e) ob[(long)e * Ho * Ho] = lrelu02(acc[e] + bias[g * Cout + co0 + 4 * q + e]);
}

// ---- fc1 of every head: hidden[b][h][j] = lrelu(b1[h][j] + feats[src(h)][b][:] . w1[h][j][:]) ------------
// Workgroup = 4 neurons of one head x the whole batch, ONE WAVE PER NEURON: a lane strides the 4096-long dot product
// (16 float4 of the weight row, re-used for every image of the batch), the wave reduces with shuffles and lane 0
// writes the result. No LDS and no workgroup barrier: an earlier form that split one dot product over the four
// waves and met in LDS gave run-to-run different sums when its workgroups started beside large-LDS workgroups of
// another stream (tools/chain_stress5.py); this form keeps every sum inside one wave. The four waves of a block read
// the same feature rows (L1 hits).
constexpr int FC_MAXB = 8;
constexpr int FC_NPB = 4;

__global__ __launch_bounds__(256) void k_fc1(const float* __restrict__ feats, const int32_t* __restrict__ head_src,
                                             const float* __restrict__ w1, const float* __restrict__ b1,
                                             float* __restrict__ hidden, int B, int D, int NH, int HID) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int neuron = blockIdx.x * FC_NPB + wave;             // over NH*HID; HID % FC_NPB == 0
    const int h = neuron / HID;
    const float4* fb = reinterpret_cast<const float4*>(feats + (long)head_src[h] * B * D);
    const float4* wr = reinterpret_cast<const float4*>(w1 + (long)neuron * D);
    const int nq = D / 4;                                      // float4 per row
    for (int b0 = 0; b0 < B; b0 += FC_MAXB) {
        const int nb = min(FC_MAXB, B - b0);
        float acc[FC_MAXB];
#pragma unroll
        for (int i = 0; i < FC_MAXB; ++i) acc[i] = 0.0f;
        auto dot = [&](int k, const float4 wv) {
#pragma unroll
            for (int i = 0; i < FC_MAXB; ++i) {
                const int row = b0 + (i < nb ? i : nb - 1);       // always a valid row: no divergent loads; extra sums unused
                const float4 f = fb[(long)row * nq + k];
                acc[i] = fmaf(wv.x, f.x, fmaf(wv.y, f.y, fmaf(wv.z, f.z, fmaf(wv.w, f.w, acc[i]))));
            }
        };
        if (nq == 16 * 64) {
            // the trunk's 4096 features: the 16 weight loads of the row (HBM misses, ~2 us each when taken one by one)
            // are issued before the first FMA
            float4 wv[16];
#pragma unroll
            for (int it = 0; it < 16; ++it) wv[it] = wr[lane + 64 * it];
#pragma unroll
            for (int it = 0; it < 16; ++it) dot(lane + 64 * it, wv[it]);
        } else {
            for (int k = lane; k < nq; k += 64) dot(k, wr[k]);
        }
#pragma unroll
        for (int i = 0; i < FC_MAXB; ++i) {
            float v = acc[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            if (lane == 0 && i < nb) hidden[((long)(b0 + i) * NH + h) * HID + neuron - h * HID] = lrelu02(v + b1[neuron]);
        }
    }
}

// ---- finish ------------------------------------------------------------------------------------------
__device__ __forceinline__ float tanh01f(float x) { return tanhf(x) * 0.5f + 0.5f; }

__global__ __launch_bounds__(1024) void k_finish(adaisp_policy_finish_args a) {
    const int b = blockIdx.x, t = threadIdx.x;
    const int F = a.num_filters, HID = a.hid, PW = a.param_width;
    __shared__ float raw[ADAISP_POLICY_MAX_FILTERS * ADAISP_MAX_PARAMS];   // fc_filter outputs
    __shared__ float logit[ADAISP_POLICY_MAX_FILTERS];
    __shared__ float pdf[ADAISP_POLICY_MAX_FILTERS];
    __shared__ int sel_sh;
    const float* hb = a.hidden + (long)b * (F + 1) * HID;

    // fc_filter rows of every head (row -> filter via a.row_filter) and the selector's fc2: one wave per row,
    // lanes stride the hidden dimension, shuffle reduction
    {
        const int lane = t & 63, wv = t >> 6;
        const int nrows = a.num_rows + F;
        constexpr int RPW = 6, KPL = 4;                 // rows per wave, hidden elements per lane held in registers
        if (nrows <= 16 * RPW && HID <= 64 * KPL) {
            // every load of the wave's rows is issued before the first use: two dependent rounds (row -> filter index,
            // then weights and hidden activations) instead of two per row
            int fidx[RPW];
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                const int r = min(wv + 16 * i, nrows - 1);
                fidx[i] = r >= a.num_rows ? F : a.row_filter[r];
            }
            float wv_[RPW][KPL], hv_[RPW][KPL];
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                const int r = min(wv + 16 * i, nrows - 1);
                const bool is_sel = r >= a.num_rows;
                const float* wrow = is_sel ? a.w_sel + (long)(r - a.num_rows) * HID : a.w_filter + (long)r * HID;
                const float* h = hb + (long)fidx[i] * HID;
#pragma unroll
                for (int j = 0; j < KPL; ++j) {
                    const int k = min(lane + 64 * j, HID - 1);
                    wv_[i][j] = wrow[k];
                    hv_[i][j] = h[k];
                }
            }
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                const int r = wv + 16 * i;
                float acc = 0.0f;
#pragma unroll
                for (int j = 0; j < KPL; ++j)
                    if (lane + 64 * j < HID) acc = fmaf(wv_[i][j], hv_[i][j], acc);     // same order as the loop form
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
                if (lane == 0 && r < nrows) {
                    const bool is_sel = r >= a.num_rows;
                    const int rr = is_sel ? r - a.num_rows : r;
                    acc += is_sel ? a.b_sel[rr] : a.b_filter[rr];
                    if (is_sel) logit[rr] = acc;
                    else raw[fidx[i] * ADAISP_MAX_PARAMS + a.row_slot[rr]] = acc;
                }
            }
        } else
        for (int r = wv; r < nrows; r += 16) {
            const bool is_sel = r >= a.num_rows;
            const int rr = is_sel ? r - a.num_rows : r;
            const int f = is_sel ? F : a.row_filter[rr];
            const float* wrow = is_sel ? a.w_sel + (long)rr * HID : a.w_filter + (long)rr * HID;
            const float* h = hb + (long)f * HID;
            float acc = 0.0f;
            for (int k = lane; k < HID; k += 64) acc = fmaf(wrow[k], h[k], acc);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
            if (lane == 0) {
                acc += is_sel ? a.b_sel[rr] : a.b_filter[rr];
                if (is_sel) logit[rr] = acc;
                else raw[f * ADAISP_MAX_PARAMS + a.row_slot[rr]] = acc;
            }
        }
    }
    __syncthreads();

    // regressors (isp/filters.py: filter_param_regressor of each class) -> params_all[b][f][slot]
    for (int r = t; r < a.num_rows; r += 1024) {
        const int f = a.row_filter[r], s = a.row_slot[r];
        const adaisp_regressor rg = a.reg[f];
        const float x = raw[f * ADAISP_MAX_PARAMS + s];
        float v;
        switch (rg.kind) {
            case ADAISP_REG_TANH_RANGE: v = tanh01f(x + rg.bias) * rg.scale + rg.lo; break;
            case ADAISP_REG_EXP_TANH_RANGE: v = expf(tanh01f(x + rg.bias) * rg.scale + rg.lo); break;
            case ADAISP_REG_SIGMOID: v = 1.0f / (1.0f + expf(-x)); break;
            case ADAISP_REG_TANH: v = tanhf(x); break;
            default: {  // ADAISP_REG_WB: exp(tanh_range(-.5,.5)(x * [0,1,1])) / (1e-5 + lum of the three gains)
                float gsc[3];
                for (int c = 0; c < 3; ++c) {
                    const float xc = raw[f * ADAISP_MAX_PARAMS + c] * (c == 0 ? 0.0f : 1.0f);
                    gsc[c] = expf(tanh01f(xc + rg.bias) * rg.scale + rg.lo);
                }
                const float lum = ((1e-5f + 0.27f * gsc[0]) + 0.67f * gsc[1]) + 0.06f * gsc[2];
                v = gsc[s] * (1.0f / lum);
            } break;
        }
        a.params_all[((long)b * F + f) * PW + s] = v;
    }

    // selector: softmax + 1e-37, exploration mix, renormalise, entropy, sample / argmax / forced (agent.py:126-149)
    // The transcendental parts (10 expf, 10 logf, 20 divisions: ~3k dependent instructions when one thread does them)
    // run one filter per lane; every SUM stays a sequential loop of one thread in the reference's order, so the values
    // are bit-identical to the single-thread form.
    __shared__ float sc[4];
    __shared__ float entl[ADAISP_POLICY_MAX_FILTERS];
    if (t < F) {
        float mx = logit[0];
        for (int k = 1; k < F; ++k) mx = fmaxf(mx, logit[k]);
        pdf[t] = expf(logit[t] - mx);
    }
    __syncthreads();
    if (t == 0) {
        float sum = 0.0f;
        for (int k = 0; k < F; ++k) sum += pdf[k];
        sc[0] = sum;
    }
    __syncthreads();
    if (t < F) {
        float p = pdf[t] / sc[0] + 1e-37f;
        pdf[t] = p * a.one_minus_exploration + a.exploration_over_f;
    }
    __syncthreads();
    if (t == 0) {
        float tot = 0.0f;
        for (int k = 0; k < F; ++k) tot += pdf[k];
        sc[1] = tot + 1e-30f;
    }
    __syncthreads();
    if (t < F) {
        const float p = pdf[t] / sc[1];
        pdf[t] = p;
        entl[t] = -p * logf(p);
    }
    __syncthreads();
    if (t == 0) {
        float ent = 0.0f;
        for (int k = 0; k < F; ++k) ent += entl[k];
        // pdf_sample: pdf / (sum + 1e-36); index = #{k : cdf_exclusive_k < u} - 1
        float s2 = 0.0f;
        for (int k = 0; k < F; ++k) s2 += pdf[k];
        s2 += 1e-36f;
        const float u = a.noise[(long)b * a.noise_stride];
        int cnt = 0, amax = 0;
        float run = 0.0f;
        for (int k = 0; k < F; ++k) {
            const float pk = pdf[k] / s2;
            run += pk;
            if (run - pk < u) ++cnt;
            if (pdf[k] > pdf[amax]) amax = k;
        }
        const int sel = a.forced_id >= 0 ? a.forced_id : (a.train_mode ? cnt - 1 : amax);
        sel_sh = sel;
        a.selected[b] = (long long)sel;
        a.op_ids[b] = (sel >= 0 && sel < F) ? a.reg[sel].op : ADAISP_OP_ZERO;
        for (int k = 0; k < F; ++k) a.pdf_out[(long)b * F + k] = pdf[k];
        a.surrogate[b] = (sel >= 0 && sel < F) ? logf(pdf[sel] + 1e-10f) : 0.0f;
        // state update + penalty (agent.py:234-280); mean(clip(x-1,0)^2) is 0 because x is clipped to [0,1]
        const int S = 3 + F;
        const float* st = a.states + (long)b * S;
        float* ns = a.new_states + (long)b * S;
        const float last = fabsf(st[2] + 1.0f - a.test_steps) < 1e-4f ? 1.0f : 0.0f;
        ns[0] = last; ns[1] = last; ns[2] = st[2] + 1.0f;
        float usage_pen = 0.0f;
        for (int k = 0; k < F; ++k) {
            const float oh = (k == sel) ? 1.0f : 0.0f;
            usage_pen += st[3 + k] * oh;
            ns[3 + k] = fmaxf(st[3 + k], oh);
        }
        const float entropy_pen = a.entropy_coef * (-ent + a.log_num_filters);
        const float early = (1.0f - last) * last * a.early_stop_penalty;
        float runtime_pen = 0.0f;
        if (a.runtime && sel >= 0 && sel < F) runtime_pen = a.runtime_lambda * a.runtime[sel];
        a.penalty[b] = 0.0f + entropy_pen + usage_pen * a.filter_usage_penalty + early + runtime_pen;
    }
    __syncthreads();
    // packed parameter row of the selected filter (zeros for the all-zero one-hot)
    const int sel = sel_sh;
    for (int s = t; s < PW; s += 1024) {
        float v = 0.0f;
        if (sel >= 0 && sel < F && s < a.reg[sel].n) v = a.params_all[((long)b * F + sel) * PW + s];
        a.packed[(long)b * PW + s] = v;
    }
}

}  // namespace

hipError_t launch_policy_conv(const float* in, const float* states, int n_state, const float* w, const float* bias,
                              float* out, int G, int B, int Cin, int Hin, int Cout, hipStream_t s) {
    const int Ho = Hin / 2;
    {   // matrix-core form: one round of TM_U input channels per wave where 16 waves suffice
        int ks = Cin / TM_U;
        if (ks > 16) ks = 16;
        if (ks >= 1 && Cout % 16 == 0 && Cin % (ks * TM_U) == 0 && (!states || Cin >= 3)) {
            dim3 grid((B * Ho * Ho + 15) / 16, Cout / 16, G);
            hipLaunchKernelGGL(k_trunk_mfma, grid, dim3(64 * ks), 0, s, in, states, n_state, w, bias, out, B, Cin, Hin, Cout, ks);
            return hipGetLastError();
        }
    }
    int KS = Cin / 8;                                   // ~8 input channels per wave
    if (KS < 1) KS = 1;
    if (KS > 16) KS = 16;
    dim3 grid((B * Ho * Ho + PX_PER - 1) / PX_PER, Cout / CO_PER, G);
    size_t smem = (size_t)KS * CO_PER * PX_PER * sizeof(float);            // partial sums
    const size_t wbytes = (size_t)Cin * CO_PER * 16 * sizeof(float);      // staged weights (Cin <= 128: 64 KB)
    if (wbytes > smem) smem = wbytes;
    hipLaunchKernelGGL(k_trunk_conv, grid, dim3(64 * KS), smem, s, in, states, n_state, w, bias, out, B, Cin, Hin, Cout,
                       KS);
    return hipGetLastError();
}

hipError_t launch_policy_fc1(const float* feats, const int32_t* head_src, const float* w1, const float* b1,
                             float* hidden, int B, int D, int NH, int HID, hipStream_t s) {
    hipLaunchKernelGGL(k_fc1, dim3(NH * HID / FC_NPB), dim3(256), 0, s, feats, head_src, w1, b1, hidden, B, D, NH, HID);
    return hipGetLastError();
}

hipError_t launch_policy_finish(const adaisp_policy_finish_args& a, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_finish, dim3(B), dim3(1024), 0, s, a);
    return hipGetLastError();
}

}  // namespace adaisp
