// Conv + bias + SiLU (+ residual) — implicit GEMM, LDS-DMA ring, second generation.
//
// Same tile / ring / swizzle scheme as yolo_conv_dma.hip; two changes aimed at what the PMC counters showed
// for that kernel (SQ_ACTIVE_INST_ANY ~47 % of wave cycles with only ~15 % of them MFMA: the waves were busy
// issuing ~180 address/select VALU instructions per k-step, not waiting for memory — L2 hit rate 92 %):
//
//   * lean DMA addressing: everything that depends on the tile row (image base, top-left tap position,
//     swizzled k-chunk) is folded once into a per-row base pointer and a 9-bit tap-validity mask; a k-step
//     adds one wave-uniform offset and selects the zero block with two v_cndmask — ~6 VALU per DMA
//     instead of ~20, no branches;
//   * v_mfma_f32_32x32x16_bf16: half the MFMA instructions for the same flops and the same LDS traffic;
//     with the (row>>1)&7 XOR key the 32-row fragment reads are bank-conflict-free.
#include "yolo_internal.h"
#include <type_traits>
#ifdef ADAYOLO_PLAIN_STORES   // A/B switch (measurement): keep the output lines in the XCD L2 instead of streaming them
#define ADAYOLO_STORE(v, p) (*(p) = (v))
#else
#define ADAYOLO_STORE(v, p) __builtin_nontemporal_store((v), (p))
#endif

namespace adayolo {
namespace dma2 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
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
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
template <int N>
__device__ __forceinline__ void wait_vm_and_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
__device__ __forceinline__ void dma16(unsigned long long gaddr, void* l) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)gaddr, (lds_ptr_t)l, 16, 0, 0);
}
__device__ __forceinline__ unsigned long long sel(bool ok, unsigned long long p, unsigned long long z) {
    const unsigned long long m = ok ? ~0ull : 0ull;
    return (p & m) | (z & ~m);
}

// ABL: ablation switch for measurements only (0 = real kernel, 1 = no DMA inside the k-loop, 2 = no LDS reads/MFMA)
// MINW: minimum waves per SIMD the register allocation must allow (2 co-resident workgroups of 8 waves need 4)
template <int BM, int BN, int WM, int WN, int STAGES, int ABL = 0, int BK = 64, int MINW = 1>
__global__ __launch_bounds__(64 * WM * WN, MINW) void k_conv_igemm_dma32(const ConvArgs a) {
    constexpr int NW = WM * WN, kThreads = 64 * NW;
    constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
    static_assert(MI >= 1 && NI >= 1, "wave tile must be a multiple of 32x32");
    static_assert(BK == 64 || BK == 32, "k-step");
    constexpr int CH = BK / 8;                 // 16-byte chunks per tile row
    constexpr int RPD = 64 / CH;               // tile rows covered by one DMA instruction (64 lanes x 16 B)
    constexpr int RB = 256 / (BK * 2);         // tile rows per 256-byte LDS bank row
    constexpr int AI = BM / (RPD * NW), WI = BN / (RPD * NW);   // DMA instructions per wave per stage
    static_assert(AI >= 1 && WI >= 1, "every wave must own at least one DMA row group per operand");
    constexpr int kStageBytes = (BM + BN) * BK * 2;
    constexpr int CP = BN + 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kRing = STAGES * kStageBytes, kEpi = BM * CP * 2;
    float* bias_s = reinterpret_cast<float*>(smem + (kRing > kEpi ? kRing : kEpi));   // [BN], behind ring and epilogue tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int lid = xcd_remap(blockIdx.x, a.mtiles * a.ntiles);
    const int m0 = (lid / a.ntiles) * BM, n0 = (lid % a.ntiles) * BN;
    // the tile's bias vector is fetched once, coalesced, while the first DMA stage is in flight (the epilogue used to
    // issue 64 dependent 4-byte global loads per lane for it: ~1/3 of the kernel time)
    for (int i = tid; i < BN; i += kThreads) bias_s[i] = (n0 + i < a.Cout) ? a.bias[n0 + i] : 0.0f;
    const unsigned long long zaddr = (unsigned long long)(const void*)g_zero16;

    // ---- per-row DMA state (fixed for the whole kernel) ------------------------------------------------
    const int slot = lane & (CH - 1), rsub = lane / CH;
    unsigned long long arow[AI], wrow[WI];
    unsigned amask[AI];                                    // bit t: tap t is inside the image for this row
    int acap[AI], wcap[WI];                                // number of valid channels from this lane's chunk start
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int r = AI * RPD * wave + RPD * i + rsub;
        const int q = slot ^ ((r / RB) & (CH - 1));
        const int m = m0 + r;
        unsigned mask = 0;
        long off = 0;
        if (m < a.M) {
            // m -> (image, row, column) by multiply-high (ConvArgs::magic_*): two runtime divisions per row cost more
            // than the whole k-loop of a 1x1 layer's tile otherwise
            const int b = a.sh_hw < 0 ? m : (int)(__umulhi((unsigned)m, a.magic_hw) >> a.sh_hw);
            const int rem = m - b * (a.Ho * a.Wo);
            const int ho = a.sh_w < 0 ? rem : (int)(__umulhi((unsigned)rem, a.magic_w) >> a.sh_w);
            const int wo = rem - ho * a.Wo;
            const int hi0 = ho * a.stride - a.pad, wi0 = wo * a.stride - a.pad;
            unsigned vw = 0;                                   // tap validity is separable: rows x columns
            // ks is 1 or 3 (checked at the ABI): three straight-line taps, no loop
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) vw |= (unsigned)(kw < a.ks && wi0 + kw >= 0 && wi0 + kw < a.W) << kw;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
                mask |= (kh < a.ks && hi0 + kh >= 0 && hi0 + kh < a.H) ? vw << (kh * a.ks) : 0u;
            off = ((long)b * a.H * a.W + (long)hi0 * a.W + wi0) * a.in_cs + 8 * q;
        }
        amask[i] = mask;
        arow[i] = (unsigned long long)(a.in + off);        // only dereferenced where the mask allows
        acap[i] = a.Cin - 8 * q;
    }
    const int Ktot = a.ks * a.ks * a.Cin;
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int r = WI * RPD * wave + RPD * i + rsub;
        const int q = slot ^ ((r / RB) & (CH - 1));
        const bool ok = (n0 + r) < a.Cout;
        wrow[i] = (unsigned long long)(a.w + (ok ? (long)(n0 + r) * Ktot + 8 * q : 0));
        wcap[i] = ok ? a.Cin - 8 * q : 0;
    }
    const int cpt = (a.Cin + BK - 1) / BK;
    const int nsteps = a.ks * a.ks * cpt;

    auto issue = [&](int step) {
        unsigned char* st = smem + (step % STAGES) * kStageBytes;
        const int tap = step / cpt, c0 = (step - tap * cpt) * BK;
        const int kh = tap / a.ks, kw = tap - kh * a.ks;
        const long aoff = 2 * (((long)kh * a.W + kw) * a.in_cs + c0);      // wave-uniform byte offsets
        const long woff = 2 * ((long)tap * a.Cin + c0);
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const bool ok = ((amask[i] >> tap) & 1u) && c0 < acap[i];
            dma16(sel(ok, arow[i] + aoff, zaddr), st + (AI * RPD * wave + RPD * i) * (BK * 2));
        }
        unsigned char* sw = st + BM * BK * 2;
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const bool ok = c0 < wcap[i];
            dma16(sel(ok, wrow[i] + woff, zaddr), sw + (WI * RPD * wave + RPD * i) * (BK * 2));
        }
    };

    f32x16 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ni][mi][e] = 0.0f;

    // fragment addressing (32x32x16): lane -> row (lane & 31), k-chunk 2*kk + (lane >> 5)
    const int frow = lane & 31, fq = lane >> 5;
    int arow_off[MI], akey[MI], wrow_off[NI], wkey[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int R = wm * TM + mi * 32 + frow;
        arow_off[mi] = R * (BK * 2);
        akey[mi] = (R / RB) & (CH - 1);
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int R = wn * TN + ni * 32 + frow;
        wrow_off[ni] = BM * BK * 2 + R * (BK * 2);
        wkey[ni] = (R / RB) & (CH - 1);
    }

    constexpr int PER = AI + WI;
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nsteps) issue(s);

    for (int step = 0; step < (ABL == 6 ? 0 : nsteps); ++step) {
        if (step + (STAGES - 2) < nsteps) wait_vm_and_barrier<PER * (STAGES - 2)>();
        else wait_vm_and_barrier<0>();
        if ((ABL == 0 || ABL == 2) && step + STAGES - 1 < nsteps) issue(step + STAGES - 1);
        const unsigned char* st = smem + (step % STAGES) * kStageBytes;
        if (ABL == 2) continue;
        // fragment reads run one 16-deep k-slice ahead of the MFMAs that consume them (register double buffer),
        // so the LDS latency hides behind the previous slice's matrix work instead of stalling every slice
        bf16x8 wf[2][NI], af[2][MI];
        auto load_frags = [&](int buf, int kk) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                wf[buf][ni] = *reinterpret_cast<const bf16x8*>(st + wrow_off[ni] + (((kk * 2 + fq) ^ wkey[ni]) << 4));
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                af[buf][mi] = *reinterpret_cast<const bf16x8*>(st + arow_off[mi] + (((kk * 2 + fq) ^ akey[mi]) << 4));
        };
        if (ABL == 3) {                      // ablation: MFMAs on register-resident fragments, no LDS reads
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) { wf[0][ni] = wf[1][ni] = __builtin_bit_cast(bf16x8, u32x4{0x3f803f80u + lane, 1u, 2u, 3u}); }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) { af[0][mi] = af[1][mi] = __builtin_bit_cast(bf16x8, u32x4{0x3f803f80u, 5u + lane, 6u, 7u}); }
        } else {
            load_frags(0, 0);
        }
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            if (ABL != 3 && kk + 1 < BK / 16) load_frags((kk + 1) & 1, kk + 1);
            if (ABL == 4) {                  // ablation: LDS reads only
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) asm volatile("" ::"v"(wf[kk & 1][ni]));
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) asm volatile("" ::"v"(af[kk & 1][mi]));
                continue;
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk & 1][ni], af[kk & 1][mi], acc[ni][mi], 0, 0, 0);
        }
    }
    wait_vm_and_barrier<0>();
    if constexpr (ABL == 5) {            // ablation: no epilogue (the never-true store keeps the MFMAs alive)
        float sum = 0.0f;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int e = 0; e < 16; ++e) sum += acc[ni][mi][e];
        if (sum == 12345.678f) a.out[0] = 1;
        return;
    }

    // ---- epilogue: D[row = channel][col = pixel]; lane holds channels (e&3) + 8*(e>>2) + 4*(lane>>5) -------
    unsigned short* Cs = reinterpret_cast<unsigned short*>(smem);
    auto convert = [&](auto silu_tag) {
        constexpr bool kSilu = decltype(silu_tag)::value;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int nl = wn * TN + ni * 32 + 8 * qd + 4 * (lane >> 5);     // 4 consecutive channels
                const float4 b4 = *reinterpret_cast<const float4*>(bias_s + nl);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    unsigned lo, hi;
                    bias_act_pack4<kSilu>(acc[ni][mi][4 * qd], acc[ni][mi][4 * qd + 1], acc[ni][mi][4 * qd + 2], acc[ni][mi][4 * qd + 3], b4, lo, hi);
                    const int ml = wm * TM + mi * 32 + (lane & 31);
                    *reinterpret_cast<u32x2*>(Cs + ml * CP + nl) = u32x2{lo, hi};
                }
            }
        }
    };
    if (a.act == ADAYOLO_ACT_SILU && !a.pre) convert(std::true_type{});   // two copies: the activation is not a per-element select
    else convert(std::false_type{});
    __syncthreads();
    constexpr int CPR = BN / 8;
    // fixed trip count -> fully unrolled, so all residual loads / LDS reads are in flight before the first store
#pragma unroll
    for (int idx = tid; idx < BM * CPR; idx += kThreads) {
        const int ml = idx / CPR, ch = (idx - ml * CPR) * 8;
        const int m = m0 + ml, n = n0 + ch;
        if (m >= a.M || n >= a.Cout) continue;
        u32x4 v = *reinterpret_cast<const u32x4*>(Cs + ml * CP + ch);
        long pix;
        int nn;
        epilogue_pos(a, m, n, pix, nn);                       // (m, n) itself, or its depth-to-space image (stride-2 data gradient)
        if (a.pre && !a.gpre) {                               // training forward: keep the pre-activation, activate its bf16 value
            *reinterpret_cast<u32x4*>(a.pre + pix * a.pre_cs + nn) = v;
            if (a.act == ADAYOLO_ACT_SILU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = silu_bf16x2(v[j]);
            }
        }
        if (a.res) {
            const u32x4 r = *reinterpret_cast<const u32x4*>(a.res + pix * a.res_cs + nn);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float lo = bf16_to_f32((unsigned short)(v[j] & 0xFFFFu)) + bf16_to_f32((unsigned short)(r[j] & 0xFFFFu));
                const float hi = bf16_to_f32((unsigned short)(v[j] >> 16)) + bf16_to_f32((unsigned short)(r[j] >> 16));
                v[j] = pack_bf16x2(lo, hi);
            }
        }
        if (ABL == 7 && v[0] != 0x12345678u) continue;        // ablation: everything but the global stores
        if (a.gpre) {                                         // backward: v is dL/d(layer output); gpre = v * silu'(pre)
            if (a.out) ADAYOLO_STORE(v, reinterpret_cast<u32x4*>(a.out + pix * a.out_cs + nn));
            const u32x4 p = *reinterpret_cast<const u32x4*>(a.pre + pix * a.pre_cs + nn);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = dsilu_bf16x2(v[j], p[j]);
            ADAYOLO_STORE(v, reinterpret_cast<u32x4*>(a.gpre + pix * a.gpre_cs + nn));
            continue;
        }
        ADAYOLO_STORE(v, reinterpret_cast<u32x4*>(a.out + pix * a.out_cs + nn));
    }
}

template <int BM, int BN, int WM, int WN, int STAGES, int ABL = 0, int BK = 64, int MINW = 1>
static hipError_t launch(ConvArgs a, hipStream_t s) {
    constexpr int ring = STAGES * (BM + BN) * BK * 2, epi = BM * (BN + 8) * 2;
    constexpr int smem = (ring > epi ? ring : epi) + BN * 4;   // the epilogue tile reuses (and may exceed) the ring; + bias
    static_assert(smem <= 160 * 1024, "LDS budget");
    auto kern = k_conv_igemm_dma32<BM, BN, WM, WN, STAGES, ABL, BK, MINW>;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        configured = true;
    }
    a.mtiles = (a.M + BM - 1) / BM;
    a.ntiles = (a.Cout + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3(a.mtiles * a.ntiles), dim3(64 * WM * WN), smem, s, a);
    return hipGetLastError();
}

}  // namespace dma2

// Variants the autotuner chooses from (each one wins layers in yolo/tuning/mi355x.json and is covered by
// tests/test_gpu_yolo_variants.py):
//    5  128x128 / 64x128 / 128x64 / 128x32 tiles by shape, BK 64, 2-stage ring
//   22  128 px x 64 ch, 4 waves, BK 32, 4-stage ring (small Cin / Cout)
//   26  128 px x 256 ch, 8 waves, BK 32, 3 stages, <= 128 VGPRs: 2 workgroups/CU
//   27  256 px x 128 ch, 8 waves, BK 32, 3 stages, 72 KB: 2 workgroups/CU
// With -DADAYOLO_MEASURE the ablation builds the DESIGN.md timeline was measured with are compiled in as well.
hipError_t launch_conv_dma2(ConvArgs a, hipStream_t s, int variant) {
    using namespace dma2;
    if (variant == 22) return launch<128, 64, 4, 1, 4, 0, 32>(a, s);
    if (variant == 26) return launch<128, 256, 2, 4, 3, 0, 32, 4>(a, s);
    if (variant == 27) return launch<256, 128, 4, 2, 3, 0, 32, 4>(a, s);
#ifdef ADAYOLO_MEASURE
    if (variant == 7) return launch<128, 128, 2, 2, 2, 1>(a, s);     // compute only
    if (variant == 8) return launch<128, 128, 2, 2, 2, 2>(a, s);     // DMA only
    if (variant == 13) return launch<256, 256, 4, 2, 2>(a, s);       // lock-step 256 x 256 (the ping-pong kernel's ancestor)
    if (variant == 20) return launch<256, 256, 4, 2, 2, 3>(a, s);    // MFMA only
    if (variant == 21) return launch<256, 256, 4, 2, 2, 4>(a, s);    // LDS reads only
    if (variant == 35) return launch<256, 256, 4, 2, 2, 5>(a, s);    // no epilogue
    if (variant == 36) return launch<256, 256, 4, 2, 2, 6>(a, s);    // no k-loop
    if (variant == 37) return launch<256, 256, 4, 2, 2, 7>(a, s);    // no global stores
#endif
    const long blocks128 = (long)((a.M + 127) / 128) * ((a.Cout + 127) / 128);
    if (a.Cout <= 32) return launch<128, 32, 4, 1, 2>(a, s);
    if (a.Cout <= 64) return launch<128, 64, 4, 1, 2>(a, s);
    if (blocks128 < 512) return launch<64, 128, 2, 2, 2>(a, s);
    return launch<128, 128, 2, 2, 2>(a, s);
}

}  // namespace adayolo
