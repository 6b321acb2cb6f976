// Training-mode trunk of the policy / critic networks for gfx950: [Conv2d(k4 s2 p1) -> BatchNorm2d(batch statistics) ->
// LeakyReLU(0.2)] x 4 on a 64x64 input, forward and backward (FeatureExtractor, agent.py:26-60 / value.py:6-44, as
// train.py:258,282-283 runs them). Through the vendor libraries one pass of one trunk is ~150 launches (layout
// transposes, im2col-style helpers, per-layer BatchNorm kernels, element-wise backward nodes) of a few microseconds each on
// tensors of 0.1-1 MB: the RL iteration runs four such passes and is bound by the host's enqueue work. Here one C call
// enqueues the whole pass — 8 launches forward, 11-15 backward — for up to two trunk instances at once:
//
//   k_tconv_fwd  the conv as an implicit GEMM D[co][px] = W[co][k] . X[k][px] on v_mfma_f32_16x16x4_f32 (fp32 products and
//                sums), operands straight from global memory / L2 like the eval kernel (isp_policy.hip), input channels split
//                over up to 16 waves that meet in LDS in slice order; writes the pre-BatchNorm output y
//   k_tbn_fwd    one workgroup per channel: the channel's B x H x W values in registers, two-pass mean / variance, the
//                activation a = lrelu(gamma (y - mean) rstd + beta), running statistics (momentum, unbiased variance)
//   k_tbn_bwd    the same shape backward: LeakyReLU', the two BatchNorm sums, dy (the gradient at the conv output),
//                dgamma / dbeta / dbias
//   k_twgrad     dW[co][ci][tap] = sum_px dy[co][px] a_in[ci][px + tap]: a 16 co x 16 tap tile of ONE input channel per
//                workgroup, pixels split over up to 16 waves; the wide early layers also over up to 8 workgroups, whose
//                partial tiles k_twgrad_reduce adds in index order
//   k_tdgrad     da_in[ci][px] = sum_co,tap dy[co][..] W[co][ci][tap] per pixel-parity class (a stride-2 k4 conv reaches an
//                input pixel through exactly 2 x 2 taps), output channels split over up to 16 waves
//   k_tplane_sum gradient of the state vector = the sum over its constant planes (first layer, critic only)
//
// Every sum runs in an order fixed by the launch geometry: the results are bit-reproducible; against the module path they
// differ by fp32 rounding (other summation orders), which tests/test_gpu_trunk_train.py measures.
#include "isp_internal.h"

namespace adaisp {
namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
constexpr int kU = 8;                       // reduction steps whose loads are all issued before the first MFMA
constexpr int kMaxG = ADAISP_TRUNK_MAX_G;
constexpr int kMaxWSplit = 8;               // pixel-range chunks of a weight-gradient tile (times the instances that share it)

// sum over the workgroup, the same order every run: lane butterfly, then the waves in index order. `red` holds 16 floats.
__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    float t = 0.0f;
    for (int w = 0; w < nw; ++w) t += red[w];
    __syncthreads();
    return t;
}

// partial accumulators of the waves 1.. of a workgroup meet wave 0's in slice order
__device__ __forceinline__ bool meet_in_lds(f32x4_t& acc, float* part, int ks, int KS, int lane) {
    if (KS > 1) {
        if (ks != 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) part[((ks - 1) * 4 + e) * 64 + lane] = acc[e];
        }
        __syncthreads();
        if (ks != 0) return false;
        for (int k = 1; k < KS; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += part[((k - 1) * 4 + e) * 64 + lane];
    }
    return true;
}

// ---- forward conv --------------------------------------------------------------------------------------------------
struct ConvIO {
    const float* in[kMaxG];     // first layer: the 3 image planes [B,3,H,H]; else the previous activation [B,Cin,H,H]
    const float* svec[kMaxG];   // first layer: [B,n_state] constant planes (channels 3..), else null
    const float* w[kMaxG];
    const float* bias[kMaxG];
    float* out[kMaxG];
};

__global__ __launch_bounds__(1024) void k_tconv_fwd(ConvIO io, int n_state, int B, int Cin, int Hin, int Cout, int KS,
                                                    int cpw) {
    __shared__ float part[15 * 4 * 64];
    const int Ho = Hin >> 1, HW = Hin * Hin, HoHo = Ho * Ho;
    const int g = blockIdx.z, co0 = blockIdx.y * 16;
    const int lane = threadIdx.x & 63, ks = threadIdx.x >> 6;
    const int col = lane & 15, q = lane >> 4;
    const int pid = blockIdx.x * 16 + col;                      // B * Ho * Ho is a multiple of 16 (host check)
    const int b = pid / HoHo, r = pid - b * HoHo, oy = r / Ho, ox = r - oy * Ho;
    const float* sv = io.svec[g];
    const int n_img = sv ? 3 : Cin;
    const float* ib = io.in[g] + (long)b * n_img * HW;
    const float* wrow = io.w[g] + (long)(co0 + col) * Cin * 16 + 4 * q;
    // lane quarter q owns kernel row kh = q: clamped addresses, zero padding as a multiply (isp_policy.hip)
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
    const float* sb = sv ? sv + b * n_state - 3 : io.in[g];
    f32x4_t acc = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int c0 = ks * cpw; c0 < (ks + 1) * cpw; c0 += kU) {
        float4 wa[kU];
        float xb[kU][4];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int ci = min(c0 + u, Cin - 1);
            const float wm = (c0 + u < Cin) ? 1.0f : 0.0f;      // channels past Cin (the slices are padded to kU): weight 0
            wa[u] = *reinterpret_cast<const float4*>(wrow + ci * 16);
            wa[u].x *= wm; wa[u].y *= wm; wa[u].z *= wm; wa[u].w *= wm;
            const bool plane = ci < n_img;
            const float* p = plane ? ib + (long)ci * HW : sb + ci;
#pragma unroll
            for (int t = 0; t < 4; ++t) xb[u][t] = p[plane ? cx[t] : 0] * fm[t];
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[u].x, xb[u][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[u].y, xb[u][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[u].z, xb[u][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[u].w, xb[u][3], acc, 0, 0, 0);
        }
    }
    if (!meet_in_lds(acc, part, ks, KS, lane)) return;
    // D[row = co][col = px]: the lane holds px = col and rows 4 q + e
    float* ob = io.out[g] + ((long)b * Cout + co0 + 4 * q) * HoHo + r;
    const float* bs = io.bias[g] + co0 + 4 * q;
#pragma unroll
    for (int e = 0; e < 4; ++e) ob[(long)e * HoHo] = acc[e] + bs[e];
}

// ---- BatchNorm (batch statistics) + LeakyReLU, forward ---------------------------------------------------------------
struct BnIO {
    const float* y[kMaxG];
    float* a[kMaxG];
    float* mean[kMaxG];
    float* rstd[kMaxG];
    const float* gamma[kMaxG];
    const float* beta[kMaxG];
    float* rmean[kMaxG];
    float* rvar[kMaxG];
};
constexpr int kPerThread = 8;               // REG: a thread keeps its values of the channel (B * H * W <= 8 * 1024: batch 8)

// REG = false re-reads the channel (L2) in each of the three passes: any batch size
template <bool REG>
__global__ __launch_bounds__(1024) void k_tbn_fwd(BnIO io, int Gseq, int B, int C, int HW, float momentum, float eps,
                                                  float slope) {
    __shared__ float red[16];
    const int c = blockIdx.x, N = B * HW, nthr = blockDim.x, tid = threadIdx.x;
    const int trips = REG ? kPerThread : (N + nthr - 1) / nthr;
    for (int s = 0; s < Gseq; ++s) {
        const int g = blockIdx.y * Gseq + s;
        const float* y = io.y[g];
        float v[kPerThread];
        float sum = 0.0f;
#pragma unroll(REG ? kPerThread : 1)
        for (int k = 0; k < trips; ++k) {
            const int i = tid + k * nthr;
            float x = 0.0f;
            if (i < N) {
                const int b = i / HW, p = i - b * HW;
                x = y[((long)b * C + c) * HW + p];
                sum += x;
            }
            if (REG) v[k] = x;
        }
        const float mean = block_sum(sum, red) / (float)N;
        float ssd = 0.0f;
#pragma unroll(REG ? kPerThread : 1)
        for (int k = 0; k < trips; ++k) {
            const int i = tid + k * nthr;
            if (i < N) {
                const int b = i / HW, p = i - b * HW;
                const float d = (REG ? v[k] : y[((long)b * C + c) * HW + p]) - mean;
                ssd += d * d;
            }
        }
        ssd = block_sum(ssd, red);
        const float rstd = 1.0f / sqrtf(ssd / (float)N + eps);
        const float ga = io.gamma[g][c], be = io.beta[g][c];
        float* a = io.a[g];
#pragma unroll(REG ? kPerThread : 1)
        for (int k = 0; k < trips; ++k) {
            const int i = tid + k * nthr;
            if (i < N) {
                const int b = i / HW, p = i - b * HW;
                const long o = ((long)b * C + c) * HW + p;
                const float z = ((REG ? v[k] : y[o]) - mean) * rstd * ga + be;
                a[o] = z > 0.0f ? z : z * slope;
            }
        }
        if (tid == 0) {
            io.mean[g][c] = mean;
            io.rstd[g][c] = rstd;
            if (io.rmean[g]) {      // nn.BatchNorm2d: running = (1 - m) running + m batch, the variance unbiased
                io.rmean[g][c] = (1.0f - momentum) * io.rmean[g][c] + momentum * mean;
                io.rvar[g][c] = (1.0f - momentum) * io.rvar[g][c] + momentum * (ssd / (float)(N - 1));
            }
        }
    }
}

// ---- the same, backward: da (gradient at the activation) -> dy (gradient at the conv output) + parameter gradients -----
struct BnBwdIO {
    const float* da[kMaxG];
    const float* a[kMaxG];
    const float* y[kMaxG];
    const float* mean[kMaxG];
    const float* rstd[kMaxG];
    const float* gamma[kMaxG];
    float* dy[kMaxG];
    float* dgamma[kMaxG];
    float* dbeta[kMaxG];
    float* dbias[kMaxG];
};

template <bool REG>
__global__ __launch_bounds__(1024) void k_tbn_bwd(BnBwdIO io, int Gseq, int B, int C, int HW, float slope) {
    __shared__ float red[16];
    const int c = blockIdx.x, N = B * HW, nthr = blockDim.x, tid = threadIdx.x;
    const int trips = REG ? kPerThread : (N + nthr - 1) / nthr;
    float tg = 0.0f, tb = 0.0f, tbias = 0.0f;
    for (int s = 0; s < Gseq; ++s) {
        const int g = blockIdx.y * Gseq + s;
        const float mean = io.mean[g][c], rstd = io.rstd[g][c];
        float dh[kPerThread], xh[kPerThread];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll(REG ? kPerThread : 1)
        for (int k = 0; k < trips; ++k) {
            const int i = tid + k * nthr;
            float d = 0.0f, x = 0.0f;
            if (i < N) {
                const int b = i / HW, p = i - b * HW;
                const long o = ((long)b * C + c) * HW + p;
                d = io.da[g][o];
                d = io.a[g][o] > 0.0f ? d : d * slope;              // a > 0 <=> the LeakyReLU's input > 0
                x = (io.y[g][o] - mean) * rstd;
                s1 += d;
                s2 += d * x;
            }
            if (REG) { dh[k] = d; xh[k] = x; }
        }
        s1 = block_sum(s1, red);
        s2 = block_sum(s2, red);
        const float m1 = s1 / (float)N, m2 = s2 / (float)N, k0 = io.gamma[g][c] * rstd;
        float s3 = 0.0f;
#pragma unroll(REG ? kPerThread : 1)
        for (int k = 0; k < trips; ++k) {
            const int i = tid + k * nthr;
            if (i < N) {
                const int b = i / HW, p = i - b * HW;
                const long o = ((long)b * C + c) * HW + p;
                float d, x;
                if (REG) { d = dh[k]; x = xh[k]; }
                else {
                    d = io.da[g][o];
                    d = io.a[g][o] > 0.0f ? d : d * slope;
                    x = (io.y[g][o] - mean) * rstd;
                }
                d = (d - m1 - x * m2) * k0;
                io.dy[g][o] = d;
                s3 += d;
            }
        }
        s3 = block_sum(s3, red);
        tg += s2; tb += s1; tbias += s3;
    }
    if (tid == 0) {
        const int g0 = blockIdx.y * Gseq;
        io.dgamma[g0][c] = tg;
        io.dbeta[g0][c] = tb;
        io.dbias[g0][c] = tbias;
    }
}

// ---- weight gradient -----------------------------------------------------------------------------------------------------
struct WgIO {
    const float* dy[kMaxG];
    const float* ain[kMaxG];
    const float* svec[kMaxG];
    float* dw[kMaxG];
};

// A[row = co][k = px] = dy, B[k = px][col = tap] = the input pixel the tap sees. Lane (col, kq) fetches 4 consecutive
// pixels of its dy row (one 16-byte load) and the 4 input pixels its tap pairs with them; MFMA j takes element j of both.
template <int U>
__global__ __launch_bounds__(1024) void k_twgrad(WgIO io, int Gseq, int PS, float* part, int n_state, int B, int Cin, int Hin,
                                                 int Cout, int KS, int spw) {
    __shared__ float part_lds[15 * 4 * 64];
    const int Ho = Hin >> 1, HW = Hin * Hin, HoHo = Ho * Ho;
    // blockIdx.x = ci + Cin * split; split = (instance of a shared parameter set, chunk of the pixel range): with more than
    // one split the tile goes to `part` [split][Cout][Cin][16] and k_twgrad_reduce adds the splits in index order
    const int ci = blockIdx.x % Cin, split = blockIdx.x / Cin;
    const int gs = split / PS, chunk = split - gs * PS;
    const int co0 = blockIdx.y * 16;
    const int lane = threadIdx.x & 63, ks = threadIdx.x >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int kh = col >> 2, kw = col & 3;
    f32x4_t acc = {0.0f, 0.0f, 0.0f, 0.0f};
    // one instance per workgroup when the range is cut (PS > 1), else the instances of a shared parameter set in order
    for (int g = (PS > 1 ? gs : 0); g < (PS > 1 ? gs + 1 : Gseq); ++g) {
        const int gg = blockIdx.z * Gseq + g;
        const float* sv = io.svec[gg];
        const bool plane = !sv || ci < 3;
        const int n_img = sv ? 3 : Cin;
        const float* dy = io.dy[gg];
        const float* ain = io.ain[gg];
        const int st0 = (chunk * KS + ks) * spw;
        for (int st = st0; st < st0 + spw; st += U) {
            float4 av[U];
            float xb[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int pb = 16 * (st + u) + 4 * kq;
                const int b = pb / HoHo, r = pb - b * HoHo, oy = r / Ho, ox0 = r - oy * Ho;
                av[u] = *reinterpret_cast<const float4*>(dy + ((long)b * Cout + co0 + col) * HoHo + r);
                const int iy = 2 * oy - 1 + kh;
                const float fy = (iy >= 0 && iy < Hin) ? 1.0f : 0.0f;
                const int rowoff = min(max(iy, 0), Hin - 1) * Hin;
                const float* p = plane ? ain + ((long)b * n_img + ci) * HW : sv + b * n_state + (ci - 3);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ix = 2 * (ox0 + j) - 1 + kw;
                    const float fm = (ix >= 0 && ix < Hin) ? fy : 0.0f;
                    xb[u][j] = p[plane ? rowoff + min(max(ix, 0), Hin - 1) : 0] * fm;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, xb[u][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, xb[u][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, xb[u][2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, xb[u][3], acc, 0, 0, 0);
            }
        }
    }
    if (!meet_in_lds(acc, part_lds, ks, KS, lane)) return;
    // D[row = co][col = tap]
    float* dw = (PS > 1 ? part + ((long)blockIdx.z * Gseq * PS + split) * Cout * Cin * 16 : io.dw[blockIdx.z * Gseq]) +
                ((long)(co0 + 4 * kq) * Cin + ci) * 16 + col;
#pragma unroll
    for (int e = 0; e < 4; ++e) dw[(long)e * Cin * 16] = acc[e];
}

__global__ __launch_bounds__(256) void k_twgrad_reduce(WgIO io, const float* __restrict__ part, int nsplit, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* p = part + (long)blockIdx.z * nsplit * n + i;
    float t = p[0];
    for (int k = 1; k < nsplit; ++k) t += p[(long)k * n];
    io.dw[blockIdx.z][i] = t;
}

// ---- data gradient -------------------------------------------------------------------------------------------------------
struct DgIO {
    const float* dy[kMaxG];
    const float* w[kMaxG];
    float* dain[kMaxG];     // [B,Cin,H,H]; first layer with dimg: channels 3.. as [B,Cin-3,H,H]; null: instance skipped
    float* dimg[kMaxG];     // first layer: channels 0..2 [B,3,H,H], else null
};

// Input pixel (iy, ix) = (2u + ry, 2v + rx) receives from the taps kh = (ry + 1) % 2 + 2a, kw = (rx + 1) % 2 + 2c at output
// pixel ((iy + 1 - kh) / 2, (ix + 1 - kw) / 2): per parity class (ry, rx) a GEMM D[ci][px] = sum_co sum_(a,c) W . dy with the
// four (a, c) as the MFMA's k = lane quarter.
__global__ __launch_bounds__(1024) void k_tdgrad(DgIO io, int B, int Cin, int Hin, int Cout, int KS, int cpw) {
    __shared__ float part[15 * 4 * 64];
    const int g = blockIdx.z >> 2, cls = blockIdx.z & 3;
    if (!io.dain[g] && !io.dimg[g]) return;
    const int ry = cls >> 1, rx = cls & 1;
    const int Ho = Hin >> 1, HW = Hin * Hin, HoHo = Ho * Ho;
    const int ci0 = blockIdx.y * 16;
    const int lane = threadIdx.x & 63, ks = threadIdx.x >> 6;
    const int col = lane & 15, q = lane >> 4;
    const int e = blockIdx.x * 16 + col;
    const int b = e / HoHo, r = e - b * HoHo, u = r / Ho, v = r - u * Ho;
    const int iy = 2 * u + ry, ix = 2 * v + rx;
    const int kh = ((ry + 1) & 1) + 2 * (q >> 1), kw = ((rx + 1) & 1) + 2 * (q & 1);
    const int oy = (iy + 1 - kh) >> 1, ox = (ix + 1 - kw) >> 1;
    const float fm = (oy >= 0 && oy < Ho && ox >= 0 && ox < Ho) ? 1.0f : 0.0f;
    const float* dyp = io.dy[g] + (long)b * Cout * HoHo + min(max(oy, 0), Ho - 1) * Ho + min(max(ox, 0), Ho - 1);
    const int cia = min(ci0 + col, Cin - 1);
    const float wm = (ci0 + col < Cin) ? 1.0f : 0.0f;
    const float* wp = io.w[g] + (long)cia * 16 + kh * 4 + kw;
    f32x4_t acc = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int c0 = ks * cpw; c0 < (ks + 1) * cpw; c0 += kU) {
        float wa[kU], xb[kU];
#pragma unroll
        for (int t = 0; t < kU; ++t) {
            wa[t] = wp[(long)(c0 + t) * Cin * 16] * wm;
            xb[t] = dyp[(long)(c0 + t) * HoHo] * fm;
        }
#pragma unroll
        for (int t = 0; t < kU; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[t], xb[t], acc, 0, 0, 0);
    }
    if (!meet_in_lds(acc, part, ks, KS, lane)) return;
    // D[row = ci][col = px]
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ci = ci0 + 4 * q + k;
        if (ci >= Cin) continue;
        if (io.dimg[g]) {
            if (ci < 3) io.dimg[g][((long)b * 3 + ci) * HW + iy * Hin + ix] = acc[k];
            else io.dain[g][((long)b * (Cin - 3) + ci - 3) * HW + iy * Hin + ix] = acc[k];
        } else {
            io.dain[g][((long)b * Cin + ci) * HW + iy * Hin + ix] = acc[k];
        }
    }
}

// dsvec[b][s] = sum over the constant plane s of image b, fixed order
struct PlaneIO {
    const float* planes[kMaxG];
    float* dsvec[kMaxG];
};
__global__ __launch_bounds__(256) void k_tplane_sum(PlaneIO io, int n_state, int HW) {
    __shared__ float red[16];
    const int g = blockIdx.z;
    if (!io.dsvec[g]) return;
    const int s = blockIdx.x, b = blockIdx.y;
    const float* p = io.planes[g] + ((long)b * n_state + s) * HW;
    float t = 0.0f;
    for (int i = threadIdx.x; i < HW; i += 256) t += p[i];
    t = block_sum(t, red);
    if (threadIdx.x == 0) io.dsvec[g][b * n_state + s] = t;
}

// ---- what lives where ------------------------------------------------------------------------------------------------------
struct Plan {
    int H[ADAISP_TRUNK_LAYERS + 1];
    size_t y[ADAISP_TRUNK_LAYERS + 1], a[ADAISP_TRUNK_LAYERS + 1], mean[ADAISP_TRUNK_LAYERS + 1], rstd[ADAISP_TRUNK_LAYERS + 1];
    size_t ws_per_g;
    size_t dy[ADAISP_TRUNK_LAYERS + 1], da[ADAISP_TRUNK_LAYERS + 1];
    size_t scr_per_g;
    size_t wpart, scr_total;            // weight-gradient partial tiles (one region, after the per-instance blocks)
};

// chunks the pixel steps of a weight-gradient tile are cut into: waves split them 16 ways, more than 4 steps per wave -> cut
int wgrad_chunks(int steps) {
    int PS = 1;
    while (PS < kMaxWSplit && steps % (2 * PS) == 0 && steps / (2 * PS) >= 16 * 4) PS *= 2;
    return PS;
}

Plan make_plan(const adaisp_trunk_args& t) {
    Plan p{};
    size_t o = 0, s = 0;
    for (int l = 0; l <= ADAISP_TRUNK_LAYERS; ++l) p.H[l] = 64 >> l;
    for (int l = 1; l <= ADAISP_TRUNK_LAYERS; ++l) {
        const size_t n = (size_t)t.B * t.C[l] * p.H[l] * p.H[l];
        p.y[l] = o; o += n;
        if (l < ADAISP_TRUNK_LAYERS) { p.a[l] = o; o += n; }            // the last activation is `feat`
        p.mean[l] = o; o += (size_t)((t.C[l] + 3) & ~3);
        p.rstd[l] = o; o += (size_t)((t.C[l] + 3) & ~3);
        p.dy[l] = s; s += n;
        if (l < ADAISP_TRUNK_LAYERS) { p.da[l] = s; s += n; }
    }
    p.da[0] = s; s += (size_t)t.B * t.C[0] * 64 * 64;                      // state-plane gradients of the first layer
    p.ws_per_g = o;
    p.scr_per_g = s;
    p.wpart = s * t.G;
    size_t wp = 0;
    for (int l = 1; l <= ADAISP_TRUNK_LAYERS; ++l) {
        const int PS = wgrad_chunks(t.B * p.H[l] * p.H[l] / 16);
        const size_t n = PS > 1 ? (size_t)PS * t.G * t.C[l] * t.C[l - 1] * 16 : 0;
        wp = n > wp ? n : wp;
    }
    p.scr_total = p.wpart + wp;
    return p;
}

int largest_divisor_le(int n, int cap) {
    for (int k = cap; k > 1; --k)
        if (n % k == 0) return k;
    return 1;
}

int check(const adaisp_trunk_args* t, bool backward) {
    if (!t) return ADAISP_EINVAL;
    if (t->G < 1 || t->G > kMaxG || t->B < 1 || t->n_state < 0 || t->C[0] != 3 + t->n_state) return ADAISP_ESHAPE;
    if (t->B > 4096) return ADAISP_ESHAPE;
    for (int l = 1; l <= ADAISP_TRUNK_LAYERS; ++l)
        if (t->C[l] < 16 || t->C[l] % 16 || t->C[l] > 1024) return ADAISP_ESHAPE;
    if (!t->feat || !t->workspace) return ADAISP_EINVAL;
    for (int g = 0; g < t->G; ++g) {
        if (!t->img[g] || (t->n_state && !t->svec[g])) return ADAISP_EINVAL;
        for (int l = 0; l < ADAISP_TRUNK_LAYERS; ++l) {
            const adaisp_trunk_params& p = t->p[g];
            if (!p.w[l] || !p.bias[l] || !p.gamma[l] || !p.beta[l]) return ADAISP_EINVAL;
            if ((p.running_mean[l] == nullptr) != (p.running_var[l] == nullptr)) return ADAISP_EINVAL;
            if (t->share_params && g && p.w[l] != t->p[0].w[l]) return ADAISP_EINVAL;
        }
        if ((t->dimg[g] == nullptr) != (t->dsvec[g] == nullptr) && t->n_state && backward) return ADAISP_EINVAL;
    }
    if (t->workspace_bytes < adaisp_trunk_train_workspace_bytes(t)) return ADAISP_ESHAPE;
    if (backward) {
        if (!t->dfeat || !t->scratch || t->scratch_bytes < adaisp_trunk_train_scratch_bytes(t)) return ADAISP_EINVAL;
        const int ng = t->share_params ? 1 : t->G;
        for (int g = 0; g < ng; ++g)
            for (int l = 0; l < ADAISP_TRUNK_LAYERS; ++l)
                if (!t->g[g].w[l] || !t->g[g].bias[l] || !t->g[g].gamma[l] || !t->g[g].beta[l]) return ADAISP_EINVAL;
    }
    return ADAISP_OK;
}

}  // namespace
}  // namespace adaisp

using namespace adaisp;

extern "C" {

size_t adaisp_trunk_train_workspace_bytes(const adaisp_trunk_args* t) {
    if (!t || t->G < 1 || t->G > kMaxG) return 0;
    return make_plan(*t).ws_per_g * t->G * sizeof(float);
}

size_t adaisp_trunk_train_scratch_bytes(const adaisp_trunk_args* t) {
    if (!t || t->G < 1 || t->G > kMaxG) return 0;
    return make_plan(*t).scr_total * sizeof(float);
}

int adaisp_trunk_train_fwd(const adaisp_trunk_args* t, void* stream) {
    const int rc = check(t, false);
    if (rc != ADAISP_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Plan p = make_plan(*t);
    const int G = t->G, B = t->B;
    const int Gseq = t->share_params ? G : 1, Gpar = t->share_params ? 1 : G;
    for (int l = 1; l <= ADAISP_TRUNK_LAYERS; ++l) {
        const int Cin = t->C[l - 1], Cout = t->C[l], Hin = p.H[l - 1], Ho = p.H[l];
        ConvIO c{};
        BnIO n{};
        for (int g = 0; g < G; ++g) {
            float* ws = t->workspace + p.ws_per_g * g;
            c.in[g] = l == 1 ? t->img[g] : ws + p.a[l - 1];
            c.svec[g] = (l == 1 && t->n_state) ? t->svec[g] : nullptr;
            c.w[g] = t->p[g].w[l - 1];
            c.bias[g] = t->p[g].bias[l - 1];
            c.out[g] = ws + p.y[l];
            n.y[g] = ws + p.y[l];
            n.a[g] = l < ADAISP_TRUNK_LAYERS ? ws + p.a[l] : t->feat + (size_t)g * B * Cout * Ho * Ho;
            n.mean[g] = ws + p.mean[l];
            n.rstd[g] = ws + p.rstd[l];
            n.gamma[g] = t->p[g].gamma[l - 1];
            n.beta[g] = t->p[g].beta[l - 1];
            n.rmean[g] = t->p[g].running_mean[l - 1];
            n.rvar[g] = t->p[g].running_var[l - 1];
        }
        int KS = (Cin + kU - 1) / kU;
        if (KS > 16) KS = 16;
        const int cpw = ((Cin + KS * kU - 1) / (KS * kU)) * kU;
        hipLaunchKernelGGL(k_tconv_fwd, dim3(B * Ho * Ho / 16, Cout / 16, G), dim3(64 * KS), 0, s, c, t->n_state, B, Cin, Hin,
                           Cout, KS, cpw);
        const int N = B * Ho * Ho;
        int nthr = N < 1024 ? ((N + 63) & ~63) : 1024;
        if (N <= 1024 * kPerThread)
            hipLaunchKernelGGL(k_tbn_fwd<true>, dim3(Cout, Gpar), dim3(nthr), 0, s, n, Gseq, B, Cout, Ho * Ho, t->momentum,
                               t->eps, t->slope);
        else
            hipLaunchKernelGGL(k_tbn_fwd<false>, dim3(Cout, Gpar), dim3(nthr), 0, s, n, Gseq, B, Cout, Ho * Ho, t->momentum,
                               t->eps, t->slope);
    }
    return hipGetLastError() == hipSuccess ? ADAISP_OK : ADAISP_ELAUNCH;
}

int adaisp_trunk_train_bwd(const adaisp_trunk_args* t, void* stream) {
    const int rc = check(t, true);
    if (rc != ADAISP_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Plan p = make_plan(*t);
    const int G = t->G, B = t->B;
    const int Gseq = t->share_params ? G : 1, Gpar = t->share_params ? 1 : G;
    bool want_input = false;
    for (int g = 0; g < G; ++g) want_input = want_input || t->dimg[g];
    for (int l = ADAISP_TRUNK_LAYERS; l >= 1; --l) {
        const int Cin = t->C[l - 1], Cout = t->C[l], Hin = p.H[l - 1], Ho = p.H[l];
        BnBwdIO n{};
        WgIO wg{};
        DgIO dg{};
        for (int g = 0; g < G; ++g) {
            float* ws = t->workspace + p.ws_per_g * g;
            float* sc = t->scratch + p.scr_per_g * g;
            const int gp = t->share_params ? 0 : g;
            n.da[g] = l == ADAISP_TRUNK_LAYERS ? t->dfeat + (size_t)g * B * Cout * Ho * Ho : sc + p.da[l];
            n.a[g] = l == ADAISP_TRUNK_LAYERS ? t->feat + (size_t)g * B * Cout * Ho * Ho : ws + p.a[l];
            n.y[g] = ws + p.y[l];
            n.mean[g] = ws + p.mean[l];
            n.rstd[g] = ws + p.rstd[l];
            n.gamma[g] = t->p[g].gamma[l - 1];
            n.dy[g] = sc + p.dy[l];
            n.dgamma[g] = t->g[gp].gamma[l - 1];
            n.dbeta[g] = t->g[gp].beta[l - 1];
            n.dbias[g] = t->g[gp].bias[l - 1];
            wg.dy[g] = sc + p.dy[l];
            wg.ain[g] = l == 1 ? t->img[g] : ws + p.a[l - 1];
            wg.svec[g] = (l == 1 && t->n_state) ? t->svec[g] : nullptr;
            wg.dw[g] = t->g[gp].w[l - 1];
            dg.dy[g] = sc + p.dy[l];
            dg.w[g] = t->p[g].w[l - 1];
            if (l > 1) {
                dg.dain[g] = sc + p.da[l - 1];
                dg.dimg[g] = nullptr;
            } else {
                dg.dain[g] = t->dimg[g] ? sc + p.da[0] : nullptr;
                dg.dimg[g] = t->dimg[g];
            }
        }
        const int N = B * Ho * Ho;
        int nthr = N < 1024 ? ((N + 63) & ~63) : 1024;
        if (N <= 1024 * kPerThread)
            hipLaunchKernelGGL(k_tbn_bwd<true>, dim3(Cout, Gpar), dim3(nthr), 0, s, n, Gseq, B, Cout, Ho * Ho, t->slope);
        else
            hipLaunchKernelGGL(k_tbn_bwd<false>, dim3(Cout, Gpar), dim3(nthr), 0, s, n, Gseq, B, Cout, Ho * Ho, t->slope);
        {
            const int steps = N / 16;
            // waves of a workgroup split the pixel steps 16 ways; where that leaves a wave more than 4 steps (the wide early
            // layers) the range is also cut into PS chunks -> partial tiles + one ordered reduction
            const int PS = wgrad_chunks(steps);
            const int per = steps / PS;
            const int KS = largest_divisor_le(per, 16), spw = per / KS;
            const int nsplit = PS > 1 ? Gseq * PS : 1;
            float* part = t->scratch + p.wpart;
            const dim3 grid(Cin * nsplit, Cout / 16, Gpar), block(64 * KS);       // (per launch: Gpar * nsplit * Cout * Cin * 16 partials)
            if (spw % 4 == 0)
                hipLaunchKernelGGL(k_twgrad<4>, grid, block, 0, s, wg, Gseq, PS, part, t->n_state, B, Cin, Hin, Cout, KS, spw);
            else if (spw % 2 == 0)
                hipLaunchKernelGGL(k_twgrad<2>, grid, block, 0, s, wg, Gseq, PS, part, t->n_state, B, Cin, Hin, Cout, KS, spw);
            else
                hipLaunchKernelGGL(k_twgrad<1>, grid, block, 0, s, wg, Gseq, PS, part, t->n_state, B, Cin, Hin, Cout, KS, spw);
            if (nsplit > 1) {
                const int n = Cout * Cin * 16;
                WgIO rd{};
                for (int g = 0; g < Gpar; ++g) rd.dw[g] = wg.dw[g * Gseq];
                hipLaunchKernelGGL(k_twgrad_reduce, dim3((n + 255) / 256, 1, Gpar), dim3(256), 0, s, rd, part, nsplit, n);
            }
        }
        if (l > 1 || want_input) {
            int KS = Cout / kU;
            if (KS > 16) KS = 16;
            while (Cout % (KS * kU)) --KS;
            const int cpw = Cout / KS;
            hipLaunchKernelGGL(k_tdgrad, dim3(B * Ho * Ho / 16, (Cin + 15) / 16, 4 * G), dim3(64 * KS), 0, s, dg, B, Cin, Hin,
                               Cout, KS, cpw);
        }
    }
    if (want_input && t->n_state) {
        PlaneIO pl{};
        for (int g = 0; g < G; ++g) {
            pl.planes[g] = t->scratch + p.scr_per_g * g + p.da[0];
            pl.dsvec[g] = t->dsvec[g];
        }
        hipLaunchKernelGGL(k_tplane_sum, dim3(t->n_state, B, G), dim3(256), 0, s, pl, t->n_state, 64 * 64);
    }
    return hipGetLastError() == hipSuccess ? ADAISP_OK : ADAISP_ELAUNCH;
}

}  // extern "C"
