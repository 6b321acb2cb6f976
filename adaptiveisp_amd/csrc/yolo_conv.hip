// Conv + folded-BN bias + SiLU (+ residual) as an implicit GEMM on the gfx950 bf16 matrix cores.
//
//   D[co][pixel] = sum_k W[co][k] * A[pixel][k],   k = (kh*KW + kw)*Cin + ci,   pixel = (b,ho,wo)
//
// Both operands are K-contiguous in memory (NHWC activations, [Cout][KH][KW][Cin] weights), so every
// global access is a 16-byte vector of 8 bf16 channels and every LDS fragment read is a ds_read_b128.
// Weights are the MFMA "A" operand (rows = output channels), activations the "B" operand
// (columns = pixels): v_mfma_f32_16x16x32_bf16 then leaves 4 consecutive CHANNELS of one pixel in a
// lane, which the epilogue packs to bf16, transposes through LDS and stores as 16-byte NHWC vectors.
//
// Workgroup = 256 threads = 4 waves; tile BM pixels x BN channels x BK=64. K loop: one tap (kh,kw) x 64
// input channels per step; the next step's global loads are issued before the current step's MFMAs and
// written to LDS after them (register-staged software pipeline, single LDS buffer, 2 barriers/step).
// LDS rows are padded to 144 B so the 16 lanes of a ds_read_b128 group hit 16 distinct 16-B slots.
// Workgroup ids are remapped so that the N-tiles of one M-tile run on the same XCD (shared L2).
//
// Reference semantics: Conv.forward_fuse = SiLU(conv2d(x) + b), yolov3/models/common.py:58-59;
// Bottleneck shortcut x + cv2(cv1(x)), common.py:119-120.
#include "yolo_internal.h"

namespace adayolo {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

constexpr int kThreads = 256;
constexpr int BK = 64;
constexpr int PITCH = BK + 8;   // bf16 elements per LDS row (144 B)

__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {   // round to nearest even
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
typedef __attribute__((ext_vector_type(2))) __bf16 hw_bf16x2;
typedef __attribute__((ext_vector_type(2))) float hw_f32x2;
// round-to-nearest-even pair conversion on the hardware unit (v_cvt_pk_bf16_f32) instead of ~8 integer VALU ops
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(hw_f32x2{lo, hi}, hw_bf16x2));
}
__device__ __forceinline__ float silu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}

// Bijective XCD remap: consecutive logical ids share an XCD (observed placement: block b -> XCD b % 8).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(kThreads) void k_conv_igemm(const ConvArgs a) {
    static_assert(WM * WN == 4, "4 waves");
    constexpr int TM = BM / WM, TN = BN / WN;     // per-wave tile
    constexpr int MI = TM / 16, NI = TN / 16;
    constexpr int AR = BM / 32, WR = BN / 32;     // staged rows per thread
    constexpr int CP = BN + 8;                    // epilogue tile pitch (bf16)
    constexpr int kStageElems = (BM + BN) * PITCH;
    constexpr int kEpiElems = BM * CP;
    constexpr int kLdsElems = kStageElems > kEpiElems ? kStageElems : kEpiElems;
    __shared__ __attribute__((aligned(16))) unsigned short lds[kLdsElems];
    unsigned short* As = lds;                 // [BM][PITCH] activations
    unsigned short* Ws = lds + BM * PITCH;    // [BN][PITCH] weights

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int lid = xcd_remap(blockIdx.x, a.mtiles * a.ntiles);
    const int m0 = (lid / a.ntiles) * BM, n0 = (lid % a.ntiles) * BN;

    // ---- staging roles: chunk q (8 channels) of rows srow + 32*i ---------------------------------
    const int q = tid & 7, srow = tid >> 3;
    int hi0[AR], wi0[AR];
    long abase[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + srow + 32 * i;
        if (m < a.M) {
            const int b = a.sh_hw < 0 ? m : (int)(__umulhi((unsigned)m, a.magic_hw) >> a.sh_hw);   // multiply-high division
            const int rem = m - b * (a.Ho * a.Wo);
            const int ho = a.sh_w < 0 ? rem : (int)(__umulhi((unsigned)rem, a.magic_w) >> a.sh_w);
            const int wo = rem - ho * a.Wo;
            hi0[i] = ho * a.stride - a.pad;
            wi0[i] = wo * a.stride - a.pad;
            abase[i] = (long)b * a.H * a.W * a.in_cs;
        } else {
            hi0[i] = -100000; wi0[i] = 0; abase[i] = 0;
        }
    }
    const int Ktot = a.ks * a.ks * a.Cin;
    const int cpt = (a.Cin + BK - 1) / BK;          // k-steps per tap
    const int nsteps = a.ks * a.ks * cpt;

    u32x4 ra[AR], rw[WR];
    auto issue_loads = [&](int step) {
        const int tap = step / cpt, c = (step - tap * cpt) * BK + 8 * q;
        const int kh = tap / a.ks, kw = tap - kh * a.ks;
        const bool cok = c < a.Cin;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int hi = hi0[i] + kh, wi = wi0[i] + kw;
            const bool ok = cok && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
            ra[i] = ok ? *reinterpret_cast<const u32x4*>(a.in + abase[i] + ((long)hi * a.W + wi) * a.in_cs + c)
                       : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            const int n = n0 + srow + 32 * i;
            const bool ok = cok && n < a.Cout;
            rw[i] = ok ? *reinterpret_cast<const u32x4*>(a.w + (long)n * Ktot + (long)tap * a.Cin + c)
                       : u32x4{0u, 0u, 0u, 0u};
        }
    };
    auto write_lds = [&]() {
#pragma unroll
        for (int i = 0; i < AR; ++i) *reinterpret_cast<u32x4*>(As + (srow + 32 * i) * PITCH + 8 * q) = ra[i];
#pragma unroll
        for (int i = 0; i < WR; ++i) *reinterpret_cast<u32x4*>(Ws + (srow + 32 * i) * PITCH + 8 * q) = rw[i];
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fk = (lane >> 4) * 8;
    issue_loads(0);
    for (int step = 0; step < nsteps; ++step) {
        write_lds();
        __syncthreads();
        if (step + 1 < nsteps) issue_loads(step + 1);     // in flight behind the MFMAs below
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            bf16x8 wf[NI], af[MI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                wf[ni] = *reinterpret_cast<const bf16x8*>(Ws + (wn * TN + ni * 16 + frow) * PITCH + kk * 32 + fk);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                af[mi] = *reinterpret_cast<const bf16x8*>(As + (wm * TM + mi * 16 + frow) * PITCH + kk * 32 + fk);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: bias + activation in fp32, bf16 pack, transpose through LDS, 16-B NHWC stores ------
    unsigned short* Cs = lds;   // [BM][CP]
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int nl = wn * TN + ni * 16 + (lane >> 4) * 4;          // 4 consecutive channels of this lane
        float bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[i] = (n0 + nl + i < a.Cout) ? a.bias[n0 + nl + i] : 0.0f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = acc[ni][mi][i] + bv[i];
                if (a.act == ADAYOLO_ACT_SILU) v[i] = silu(v[i]);
            }
            const int ml = wm * TM + mi * 16 + (lane & 15);
            *reinterpret_cast<u32x2*>(Cs + ml * CP + nl) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        }
    }
    __syncthreads();
    constexpr int CPR = BN / 8;   // 16-B pieces per pixel row
    for (int idx = tid; idx < BM * CPR; idx += kThreads) {
        const int ml = idx / CPR, ch = (idx - ml * CPR) * 8;
        const int m = m0 + ml, n = n0 + ch;
        if (m >= a.M || n >= a.Cout) continue;
        u32x4 v = *reinterpret_cast<const u32x4*>(Cs + ml * CP + ch);
        if (a.res) {
            const u32x4 r = *reinterpret_cast<const u32x4*>(a.res + (long)m * a.res_cs + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float lo = bf16_to_f32((unsigned short)(v[j] & 0xFFFFu)) + bf16_to_f32((unsigned short)(r[j] & 0xFFFFu));
                const float hi = bf16_to_f32((unsigned short)(v[j] >> 16)) + bf16_to_f32((unsigned short)(r[j] >> 16));
                v[j] = pack_bf16x2(lo, hi);
            }
        }
        *reinterpret_cast<u32x4*>(a.out + (long)m * a.out_cs + n) = v;
    }
}

template <int BM, int BN, int WM, int WN>
static hipError_t launch(ConvArgs a, hipStream_t s) {
    a.mtiles = (a.M + BM - 1) / BM;
    a.ntiles = (a.Cout + BN - 1) / BN;
    hipLaunchKernelGGL((k_conv_igemm<BM, BN, WM, WN>), dim3(a.mtiles * a.ntiles), dim3(kThreads), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_conv(ConvArgs a, hipStream_t s) {
    if (a.Cout <= 32) return launch<128, 32, 4, 1>(a, s);
    if (a.Cout <= 64) return launch<128, 64, 4, 1>(a, s);
    // small feature maps: halve the pixel tile so the grid still covers the 256 CUs a few times
    const long blocks128 = (long)((a.M + 127) / 128) * ((a.Cout + 127) / 128);
    if (blocks128 < 512) return launch<64, 128, 2, 2>(a, s);
    return launch<128, 128, 2, 2>(a, s);
}

}  // namespace adayolo
