// The policy's parameter heads in training mode (agent.py:103-116, 234-259: per filter fc1 4096 -> 128, LeakyReLU 0.2, fc_filter
// 128 -> n; the selector's fc1 4096 -> 128, LeakyReLU, fc2 128 -> F) forward and backward on the filters' OWN parameter tensors
// (gfx950). Through ATen the batched form of these heads is 19 launches forward and ~25 backward per iteration — four
// concatenations and two zero-padded scatters that rebuild a 23 MB weight layout, four rocBLAS GEMMs with M = B = 8, their
// activations and the mirror of all that — 0.215 + 0.12 ms of a 6.5 ms iteration in which the chip idles (tools/train_timeline.sh).
// Here: 2 launches forward, 4 backward, every weight read once per pass through a pointer table, every sum in a fixed order.
#include "isp_internal.h"

namespace adaisp {
namespace {

constexpr int kRows = 8;                    // fc1 rows per workgroup (two per wave)
constexpr int kChunk = 1024;                // features staged per round: B x kChunk floats in LDS
constexpr int kMaxB = ADAISP_HEADS_MAX_B;

__device__ __forceinline__ float lrelu(float v) { return v > 0.0f ? v : 0.2f * v; }
__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// row r of the stacked fc1 matrices: filter r / hid, or the selector (group F)
__device__ __forceinline__ const float* fc1_row(const adaisp_heads_args& a, int group, int j) {
    return (group < a.F ? a.w1[group] : a.ws1) + (long)j * a.D;
}

// hidden[b][g][j] = b1[g][j] + sum_k W1[g][j][k] * feat_g[b][k]          (pre-activation; g = F: the selector on ITS features)
// grid ((F + 1) * hid / kRows), 256 threads: the workgroup's features chunk by chunk through LDS, two rows per wave
__global__ __launch_bounds__(256) void k_heads_fc1(const adaisp_heads_args a) {
    __shared__ float sf[kMaxB * kChunk];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * kRows, group = row0 / a.hid, j0 = row0 - group * a.hid + 2 * wave;
    const float* feat = group < a.F ? a.feat_f : a.feat_s;
    const float* w0 = fc1_row(a, group, j0);
    const float* w1 = w0 + a.D;
    float acc0[kMaxB], acc1[kMaxB];
#pragma unroll
    for (int b = 0; b < kMaxB; ++b) acc0[b] = acc1[b] = 0.0f;
    for (int k0 = 0; k0 < a.D; k0 += kChunk) {
        // this chunk's weights first: 8 loads per lane in flight while the features are staged
        float4 u[kChunk / 256], v[kChunk / 256];
#pragma unroll
        for (int i = 0; i < kChunk / 256; ++i) {
            u[i] = *reinterpret_cast<const float4*>(w0 + k0 + 4 * lane + 256 * i);
            v[i] = *reinterpret_cast<const float4*>(w1 + k0 + 4 * lane + 256 * i);
        }
        __syncthreads();
        for (int t = tid; t < a.B * (kChunk / 4); t += 256) {
            const int b = t / (kChunk / 4), q = t - b * (kChunk / 4);
            reinterpret_cast<float4*>(sf)[b * (kChunk / 4) + q] = *reinterpret_cast<const float4*>(feat + (long)b * a.D + k0 + 4 * q);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kChunk / 256; ++i) {
            const int k = 4 * lane + 256 * i;
#pragma unroll
            for (int b = 0; b < kMaxB; ++b) {
                if (b < a.B) {
                    const float4 f = *reinterpret_cast<const float4*>(sf + b * kChunk + k);
                    acc0[b] = fmaf(u[i].w, f.w, fmaf(u[i].z, f.z, fmaf(u[i].y, f.y, fmaf(u[i].x, f.x, acc0[b]))));
                    acc1[b] = fmaf(v[i].w, f.w, fmaf(v[i].z, f.z, fmaf(v[i].y, f.y, fmaf(v[i].x, f.x, acc1[b]))));
                }
            }
        }
    }
    const float* bias = group < a.F ? a.b1[group] : a.bs1;
#pragma unroll
    for (int b = 0; b < kMaxB; ++b) {
        if (b < a.B) {
            const float s0 = wave_sum64(acc0[b]), s1 = wave_sum64(acc1[b]);
            if (lane == 0) {
                float* h = a.hidden + ((long)b * (a.F + 1) + group) * a.hid + j0;
                h[0] = s0 + bias[j0];
                h[1] = s1 + bias[j0 + 1];
            }
        }
    }
}

// x[b][f][r] = bf[f][r] + sum_h Wf[f][r][h] * lrelu(hidden[b][f][h])  (r < n_f; 0 beyond), logits[b][f] likewise from the selector's row
// grid (B, F + 1), 256 threads: one output row per wave and round, the lanes over hid
__global__ __launch_bounds__(256) void k_heads_out(const adaisp_heads_args a) {
    const int b = blockIdx.x, g = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, G = a.F + 1;
    const bool sel = g == a.F;
    const float* hid = a.hidden + ((long)b * G + g) * a.hid;
    const float* w2 = sel ? a.ws2 : a.wf[g];
    const float* b2 = sel ? a.bs2 : a.bf[g];
    const int rows = sel ? a.F : a.n[g], total = sel ? a.F : a.pw;
    float* out = sel ? a.logits + (long)b * a.F : a.x + ((long)b * a.F + g) * a.pw;
    for (int r = wave; r < total; r += 4) {
        float s = 0.0f;
        if (r < rows)
            for (int h = lane; h < a.hid; h += 64) s = fmaf(w2[(long)r * a.hid + h], lrelu(hid[h]), s);
        s = wave_sum64(s);
        if (lane == 0) out[r] = r < rows ? s + b2[r] : 0.0f;
    }
}

// Backward, stage 1 — grid (F + 1), 256 threads, one group each: d pre-activation dhid[b][g][h] for every image, and the gradients of
// the group's second layer (dWf / dbf, or dWs2 / dbs2) summed over the images in image order.
__global__ __launch_bounds__(256) void k_heads_dhid(const adaisp_heads_args a) {
    extern __shared__ float sh[];                           // [B][hid] activations, then [B][rows] upstream gradients
    const int g = blockIdx.x, tid = threadIdx.x, G = a.F + 1;
    const bool sel = g == a.F;
    const int rows = sel ? a.F : a.n[g];
    float* act = sh;
    float* up = sh + a.B * a.hid;
    for (int t = tid; t < a.B * a.hid; t += 256) {
        const int b = t / a.hid, h = t - b * a.hid;
        act[t] = a.hidden[((long)b * G + g) * a.hid + h];    // pre-activation (its sign decides the slope)
    }
    for (int t = tid; t < a.B * rows; t += 256) {
        const int b = t / rows, r = t - b * rows;
        up[t] = sel ? a.dlogits[(long)b * a.F + r] : a.dx[((long)b * a.F + g) * a.pw + r];
    }
    __syncthreads();
    const float* w2 = sel ? a.ws2 : a.wf[g];
    for (int t = tid; t < a.B * a.hid; t += 256) {
        const int b = t / a.hid, h = t - b * a.hid;
        float s = 0.0f;
        for (int r = 0; r < rows; ++r) s = fmaf(w2[(long)r * a.hid + h], up[b * rows + r], s);
        a.dhid[((long)b * G + g) * a.hid + h] = act[t] > 0.0f ? s : 0.2f * s;
    }
    float* dw2 = sel ? a.dws2 : a.dwf[g];
    float* db2 = sel ? a.dbs2 : a.dbf[g];
    for (int t = tid; t < rows * a.hid; t += 256) {
        const int r = t / a.hid, h = t - r * a.hid;
        float s = 0.0f;
        for (int b = 0; b < a.B; ++b) s = fmaf(up[b * rows + r], lrelu(act[b * a.hid + h]), s);
        dw2[t] = s;
    }
    for (int r = tid; r < rows; r += 256) {
        float s = 0.0f;
        for (int b = 0; b < a.B; ++b) s += up[b * rows + r];
        db2[r] = s;
    }
}

// Backward, stage 2 — dW1[g][j][k] = sum_b dhid[b][g][j] * feat_g[b][k], db1[g][j] = sum_b dhid[b][g][j]; the geometry of k_heads_fc1
__global__ __launch_bounds__(256) void k_heads_dw1(const adaisp_heads_args a) {
    __shared__ float sf[kMaxB * kChunk];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, G = a.F + 1;
    const int row0 = blockIdx.x * kRows, group = row0 / a.hid, j0 = row0 - group * a.hid + 2 * wave;
    const float* feat = group < a.F ? a.feat_f : a.feat_s;
    float* d0 = (group < a.F ? a.dw1[group] : a.dws1) + (long)j0 * a.D;
    float* d1 = d0 + a.D;
    float g0[kMaxB], g1[kMaxB];
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int b = 0; b < kMaxB; ++b) {
        g0[b] = g1[b] = 0.0f;
        if (b < a.B) {
            g0[b] = a.dhid[((long)b * G + group) * a.hid + j0];
            g1[b] = a.dhid[((long)b * G + group) * a.hid + j0 + 1];
            s0 += g0[b];
            s1 += g1[b];
        }
    }
    if (lane == 0) {
        float* db = group < a.F ? a.db1[group] : a.dbs1;
        db[j0] = s0;
        db[j0 + 1] = s1;
    }
    for (int k0 = 0; k0 < a.D; k0 += kChunk) {
        __syncthreads();
        for (int t = tid; t < a.B * (kChunk / 4); t += 256) {
            const int b = t / (kChunk / 4), q = t - b * (kChunk / 4);
            reinterpret_cast<float4*>(sf)[b * (kChunk / 4) + q] = *reinterpret_cast<const float4*>(feat + (long)b * a.D + k0 + 4 * q);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kChunk / 256; ++i) {
            const int k = 4 * lane + 256 * i;
            float4 u = {0.0f, 0.0f, 0.0f, 0.0f}, v = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int b = 0; b < kMaxB; ++b) {
                if (b < a.B) {
                    const float4 f = *reinterpret_cast<const float4*>(sf + b * kChunk + k);
                    u.x = fmaf(g0[b], f.x, u.x); u.y = fmaf(g0[b], f.y, u.y); u.z = fmaf(g0[b], f.z, u.z); u.w = fmaf(g0[b], f.w, u.w);
                    v.x = fmaf(g1[b], f.x, v.x); v.y = fmaf(g1[b], f.y, v.y); v.z = fmaf(g1[b], f.z, v.z); v.w = fmaf(g1[b], f.w, v.w);
                }
            }
            *reinterpret_cast<float4*>(d0 + k0 + k) = u;
            *reinterpret_cast<float4*>(d1 + k0 + k) = v;
        }
    }
}

// Backward, stage 3 — the features' gradient: part[g][b][k] = sum_j dhid[b][g][j] * W1[g][j][k] per group (grid (D / 256, F + 1)),
// then dfeat_f[b][k] = sum over the filters' groups in group order, dfeat_s = the selector's part (grid (D / 256, B)).
__global__ __launch_bounds__(256) void k_heads_dfeat_part(const adaisp_heads_args a, float* __restrict__ part) {
    extern __shared__ float sg[];                           // [B][hid] dhid of this group, then [4][B][256] partials of the waves
    const int g = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, G = a.F + 1;
    const int k = blockIdx.x * 256 + 4 * lane;             // this lane's four columns
    float* red = sg + a.B * a.hid;
    for (int t = tid; t < a.B * a.hid; t += 256) {
        const int b = t / a.hid, h = t - b * a.hid;
        sg[t] = a.dhid[((long)b * G + g) * a.hid + h];
    }
    __syncthreads();
    const float* w = (g < a.F ? a.w1[g] : a.ws1) + k;
    float4 acc[kMaxB];
#pragma unroll
    for (int b = 0; b < kMaxB; ++b) acc[b] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int per = a.hid / 4;                             // rows of this wave: [wave * per, (wave + 1) * per), in row order
#pragma unroll 4
    for (int j = wave * per; j < (wave + 1) * per; ++j) {
        const float4 wv = *reinterpret_cast<const float4*>(w + (long)j * a.D);
#pragma unroll
        for (int b = 0; b < kMaxB; ++b) {
            if (b < a.B) {
                const float d = sg[b * a.hid + j];
                acc[b].x = fmaf(d, wv.x, acc[b].x); acc[b].y = fmaf(d, wv.y, acc[b].y);
                acc[b].z = fmaf(d, wv.z, acc[b].z); acc[b].w = fmaf(d, wv.w, acc[b].w);
            }
        }
    }
#pragma unroll
    for (int b = 0; b < kMaxB; ++b)
        if (b < a.B) *reinterpret_cast<float4*>(red + (wave * a.B + b) * 256 + 4 * lane) = acc[b];
    __syncthreads();
    for (int t = tid; t < a.B * 256; t += 256) {
        const int b = t >> 8, c = t & 255;
        const float s = ((red[(0 * a.B + b) * 256 + c] + red[(1 * a.B + b) * 256 + c]) + red[(2 * a.B + b) * 256 + c]) + red[(3 * a.B + b) * 256 + c];
        part[((long)g * a.B + b) * a.D + blockIdx.x * 256 + c] = s;
    }
}
__global__ __launch_bounds__(256) void k_heads_dfeat_sum(const adaisp_heads_args a, const float* __restrict__ part) {
    const int b = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    float s = 0.0f;
    for (int g = 0; g < a.F; ++g) s += part[((long)g * a.B + b) * a.D + k];
    a.dfeat_f[(long)b * a.D + k] = s;
    a.dfeat_s[(long)b * a.D + k] = part[((long)a.F * a.B + b) * a.D + k];
}

int heads_check(const adaisp_heads_args* a, bool bwd) {
    if (!a) return ADAISP_EINVAL;
    if (a->B < 1 || a->B > kMaxB || a->F < 1 || a->F > ADAISP_POLICY_MAX_FILTERS || a->hid < 2 || a->hid > 256 || (a->hid % kRows) ||
        a->D < kChunk || (a->D % kChunk) || a->pw < 1 || a->pw > ADAISP_MAX_PARAMS)
        return ADAISP_ESHAPE;
    if (!a->feat_f || !a->feat_s || !a->ws1 || !a->bs1 || !a->ws2 || !a->bs2 || !a->hidden || !a->x || !a->logits) return ADAISP_EINVAL;
    for (int f = 0; f < a->F; ++f) {
        if (a->n[f] < 1 || a->n[f] > a->pw) return ADAISP_ESHAPE;
        if (!a->w1[f] || !a->b1[f] || !a->wf[f] || !a->bf[f]) return ADAISP_EINVAL;
        if (bwd && (!a->dw1[f] || !a->db1[f] || !a->dwf[f] || !a->dbf[f])) return ADAISP_EINVAL;
    }
    if (bwd && (!a->dx || !a->dlogits || !a->dhid || !a->part || !a->dws1 || !a->dbs1 || !a->dws2 || !a->dbs2 || !a->dfeat_f || !a->dfeat_s))
        return ADAISP_EINVAL;
    return ADAISP_OK;
}

}  // namespace
}  // namespace adaisp

using namespace adaisp;

extern "C" {

int adaisp_heads_fwd(const adaisp_heads_args* a, void* stream) {
    const int rc = heads_check(a, false);
    if (rc != ADAISP_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int G = a->F + 1;
    hipLaunchKernelGGL(k_heads_fc1, dim3(G * a->hid / kRows), dim3(256), 0, s, *a);
    hipLaunchKernelGGL(k_heads_out, dim3(a->B, G), dim3(256), 0, s, *a);
    return hipGetLastError() == hipSuccess ? ADAISP_OK : ADAISP_ELAUNCH;
}

int adaisp_heads_bwd(const adaisp_heads_args* a, void* stream) {
    const int rc = heads_check(a, true);
    if (rc != ADAISP_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int G = a->F + 1;
    const int rows = a->pw > a->F ? a->pw : a->F;
    hipLaunchKernelGGL(k_heads_dhid, dim3(G), dim3(256), (size_t)a->B * (a->hid + rows) * sizeof(float), s, *a);
    hipLaunchKernelGGL(k_heads_dw1, dim3(G * a->hid / kRows), dim3(256), 0, s, *a);
    hipLaunchKernelGGL(k_heads_dfeat_part, dim3(a->D / 256, G), dim3(256), (size_t)a->B * (a->hid + 4 * 256) * sizeof(float), s, *a, a->part);
    hipLaunchKernelGGL(k_heads_dfeat_sum, dim3(a->D / 256, a->B), dim3(256), 0, s, *a, static_cast<const float*>(a->part));
    return hipGetLastError() == hipSuccess ? ADAISP_OK : ADAISP_ELAUNCH;
}

}  // extern "C"
