// 1x1 Conv + bias + SiLU for the deep, narrow-map layers (Bottleneck.cv1 of the C = 512 stage: 512 -> 256 on 46 x 80 maps,
// yolov3/models/common.py:45-59,110-120) — "whole K at once": no k-loop, no ring, no barrier between memory and matrix work.
//
// These layers are 7.7 GFLOP on 29 440 pixels: 230 tiles of the 256 x 128 ring kernel, ONE round on 256 CUs, 18-21 us of which
// the matrix pipes work 2 (eight k-tiles of prologue / barrier / epilogue structure around 4k cycles of MFMA). Here a workgroup
// takes 128 pixels x 256 output channels and
//   * requests its WHOLE activation tile (128 px x K x 2 B = 128 KB at K = 512) with LDS-DMA in one burst — every byte the
//     workgroup needs is in flight after ~150 issue cycles per wave, nothing waits for a ring slot;
//   * keeps the weights in REGISTERS: wave w owns output channels 32 w .. 32 w + 31, its K / 16 MFMA operand fragments
//     (128 registers at K = 512) are loaded once, 1 KB contiguous per instruction, from a FRAGMENT-MAJOR copy of the weight matrix
//     (include/adayolo.h: adayolo_conv1x1_stream_fwd; row-major, the same loads touch 32 cache lines each instead of 8);
//   * runs K / 16 x 4 = 128 MFMAs per wave straight through (one ds_read_b128 per MFMA; rows are 2 K bytes, so 16-byte chunks
//     are XOR-keyed by pixel & 15: every ds_read_b128 lane group — pixels {0-3, 12-15, 20-27} or {4-11, 16-19, 28-31} — meets
//     16 distinct keys, conflict-free);
//   * transposes the bf16 result through the (dead) activation tile and stores whole 512-byte pixel rows, non-temporal.
// Restrictions (hipErrorInvalidValue otherwise, the caller keeps its ring kernel): ksize 1, stride 1, no residual,
// Cin in {256, 512}, Cout % 256 == 0.
#include "yolo_internal.h"

namespace adayolo {
namespace k1 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ void dma16(unsigned long long gaddr, void* l) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)gaddr, (lds_ptr_t)l, 16, 0, 0);
}
__device__ __forceinline__ void barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

constexpr int BM = 128, BN = 256;
constexpr int kOutPitch = BN * 2 + 16;               // bytes per pixel row of the output tile in LDS (2-way write conflicts at most)

template <int K>
struct Geo {
    static constexpr int RB = 2 * K;                 // bytes per activation row (one pixel)
    static constexpr int kTile = BM * RB;            // 128 KB at K = 512
    static constexpr int NDMA = kTile / 1024 / 8;    // DMA instructions per wave
    static constexpr int KK = K / 16;                // MFMA k-steps = weight fragments per wave
    static constexpr int kSmem = (kTile > BM * kOutPitch ? kTile : BM * kOutPitch);
};

template <int K, bool SILU>
__global__ __launch_bounds__(512) void k_conv_k1(const ConvArgs a) {
    using G = Geo<K>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = blockIdx.x / a.ntiles, nt = blockIdx.x - mt * a.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const unsigned long long zaddr = (unsigned long long)(const void*)g_zero16;

    // ---- activation tile: instruction g of the tile fills LDS bytes [1024 g, 1024 g + 1024); lane -> (pixel, chunk') there,
    //      source chunk = chunk' ^ (pixel & 15). Pixels past M read the zero page.
#pragma unroll
    for (int i = 0; i < G::NDMA; ++i) {
        const int g = wave + 8 * i;
        const int o = g * 1024 + lane * 16;
        const int p = o / G::RB, c = ((o % G::RB) >> 4) ^ (p & 15);
        const bool ok = m0 + p < a.M;
        const unsigned long long src = (unsigned long long)(a.in + (long)(m0 + p) * a.in_cs) + 16ull * c;
        dma16(ok ? src : zaddr, smem + g * 1024);
    }
    // ---- weights: fragment kk of channels n0 + 32 wave .. + 31 (fragment-major: 1 KB per instruction)
    bf16x8 wf[G::KK];
    {
        const unsigned short* wp = a.w + ((long)((n0 >> 5) + wave) * G::KK * 64 + lane) * 8;
#pragma unroll
        for (int kk = 0; kk < G::KK; ++kk) wf[kk] = *reinterpret_cast<const bf16x8*>(wp + kk * 64 * 8);
    }
    const int r = lane & 31, fq = lane >> 5;
    float4 bq[4];
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) bq[qd] = *reinterpret_cast<const float4*>(a.bias + n0 + 32 * wave + 8 * qd + 4 * fq);

    f32x16 acc[4];
#pragma unroll
    for (int pf = 0; pf < 4; ++pf)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[pf][e] = 0.0f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    barrier();

    // ---- K / 16 steps x 4 pixel fragments; the fragments of step kk + 1 are requested before the MFMAs of step kk
    const unsigned char* abase = smem + r * G::RB;
    const int key = r & 15;
    bf16x8 af[2][4];
#pragma unroll
    for (int pf = 0; pf < 4; ++pf) af[0][pf] = *reinterpret_cast<const bf16x8*>(abase + pf * 32 * G::RB + ((fq ^ key) << 4));
    // (a fence per step: left alone, hipcc sinks every read to just in front of its MFMA — lgkmcnt(0) four times per step)
#pragma unroll
    for (int kk = 0; kk < G::KK; ++kk) {
        if (kk + 1 < G::KK) {
#pragma unroll
            for (int pf = 0; pf < 4; ++pf)
                af[(kk + 1) & 1][pf] = *reinterpret_cast<const bf16x8*>(abase + pf * 32 * G::RB + (((2 * (kk + 1) + fq) ^ key) << 4));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pf = 0; pf < 4; ++pf) acc[pf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk], af[kk & 1][pf], acc[pf], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    barrier();                                            // every wave has read the activation tile: the output tile overlays it

    // ---- epilogue: D[row = channel][col = pixel]; lane holds pixel r and channels 8 qd + 4 fq + (0..3) of its wave's 32
#pragma unroll
    for (int pf = 0; pf < 4; ++pf) {
        unsigned char* const wr = smem + (pf * 32 + r) * kOutPitch + (32 * wave + 4 * fq) * 2;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            unsigned lo, hi;
            bias_act_pack4<SILU>(acc[pf][4 * qd], acc[pf][4 * qd + 1], acc[pf][4 * qd + 2], acc[pf][4 * qd + 3], bq[qd], lo, hi);
            *reinterpret_cast<u32x2*>(wr + 8 * qd * 2) = u32x2{lo, hi};
        }
    }
    barrier();
    // 128 px x 512 B: a lane takes 16 B (8 channels); one wave instruction = two whole pixel rows
#pragma unroll
    for (int it = 0; it < BM * (BN / 8) / 512; ++it) {
        const int idx = it * 512 + tid, px = idx >> 5, chunk = idx & 31;
        const u32x4 v = *reinterpret_cast<const u32x4*>(smem + px * kOutPitch + chunk * 16);
        if (m0 + px < a.M)
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(a.out + (long)(m0 + px) * a.out_cs + n0 + chunk * 8));
    }
}

template <int K, bool SILU>
static hipError_t launch(ConvArgs a, hipStream_t s) {
    using G = Geo<K>;
    static_assert(G::kSmem <= 160 * 1024, "LDS budget");
    auto kern = k_conv_k1<K, SILU>;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::kSmem);
        if (e != hipSuccess) return e;
        configured = true;
    }
    a.mtiles = (a.M + BM - 1) / BM;
    a.ntiles = a.Cout / BN;
    hipLaunchKernelGGL(kern, dim3(a.mtiles * a.ntiles), dim3(512), G::kSmem, s, a);
    return hipGetLastError();
}

}  // namespace k1

// a.w: the FRAGMENT-MAJOR weight copy (include/adayolo.h). hipErrorInvalidValue -> the shape is not served.
hipError_t launch_conv_k1(ConvArgs a, hipStream_t s) {
    if (a.ks != 1 || a.stride != 1 || a.res || a.Cout % 256 || (a.Cin != 256 && a.Cin != 512)) return hipErrorInvalidValue;
    const bool silu = a.act == ADAYOLO_ACT_SILU;
    if (a.Cin == 512) return silu ? k1::launch<512, true>(a, s) : k1::launch<512, false>(a, s);
    return silu ? k1::launch<256, true>(a, s) : k1::launch<256, false>(a, s);
}

}  // namespace adayolo
