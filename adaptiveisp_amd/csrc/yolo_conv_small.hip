// 3x3 Conv + bias + SiLU (+ residual) for the SHALLOW layers (Cin = 32 or 64): the whole reduction is resident.
//
// With K = 9*Cin <= 576 the implicit-GEMM ring has only 9-18 tiny k-steps per tile; each one pays a barrier and an
// LDS-DMA round trip for 4-8 MFMAs, so those layers ran latency-bound at 2-3x their HBM floor. Here a workgroup
// loads, in ONE DMA round, (a) the input patch of its output tile (stride 1 or 2, 1-px halo) and (b) ALL nine
// weight taps of its output channels, then runs the 9 x Cin/16 MFMA steps back to back from LDS and stores.
// Several workgroups per CU hide the single load latency. HBM traffic = input once (+halo) + output once.
#include "yolo_internal.h"
#include <type_traits>

namespace adayolo {
namespace smallk {

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
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
__device__ __forceinline__ void dma16(unsigned long long gaddr, void* l) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)gaddr, (lds_ptr_t)l, 16, 0, 0);
}

// swizzle key of a tile row whose rows are CIN*2 bytes (64 B -> 4 rows per 256-B bank row, 128 B -> 2)
template <int CIN>
__device__ __forceinline__ int row_key(int r) {
    return CIN == 32 ? ((r >> 2) & 3) : ((r >> 1) & 7);
}

template <int CIN, int BN, int S, int TPH, int TPW, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void k_conv3x3_small(const ConvArgs a, int tiles_x, int tiles_y) {
    constexpr int NW = WM * WN, kThreads = 64 * NW, BM = TPH * TPW;
    constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
    static_assert(MI >= 1 && NI >= 1 && TPW % 32 == 0, "wave tiles are 32 px (one output row segment) x 32 ch");
    constexpr int PH = (TPH - 1) * S + 3, PW = (TPW - 1) * S + 3, PROWS = PH * PW;
    constexpr int RB = CIN * 2;                   // bytes per LDS row (one pixel / one weight row of one tap)
    constexpr int CH = CIN / 8;                   // 16-byte chunks per row
    constexpr int RPD = 64 / CH;                  // rows per DMA instruction
    constexpr int PINS = (PROWS + RPD - 1) / RPD; // patch DMA instructions
    constexpr int WINS = 9 * BN / RPD;            // weight DMA instructions (tap-major tiles of BN rows)
    constexpr int PBYTES = PINS * RPD * RB;
    constexpr int CP = BN + 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* pbuf = smem;
    unsigned char* wbuf = smem + PBYTES;          // [9][BN][CIN]
    constexpr int kRing = PBYTES + 9 * BN * RB, kEpi = BM * CP * 2;
    float* bias_s = reinterpret_cast<float*>(smem + (kRing > kEpi ? kRing : kEpi));   // [BN]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int ntn = (a.Cout + BN - 1) / BN;
    int t = blockIdx.x;
    const int nt = t % ntn; t /= ntn;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int ox0 = tx * TPW, oy0 = ty * TPH, n0 = nt * BN;
    const int ix0 = ox0 * S - 1, iy0 = oy0 * S - 1;
    const unsigned long long zaddr = (unsigned long long)(const void*)g_zero16;
    for (int i = tid; i < BN; i += kThreads) bias_s[i] = (n0 + i < a.Cout) ? a.bias[n0 + i] : 0.0f;   // once, coalesced

    // ---- one DMA round: patch rows, then the nine weight tiles; instructions are dealt round-robin to the waves ---
    const int slot = lane % CH, rsub = lane / CH;
    for (int d = wave; d < PINS + WINS; d += NW) {
        unsigned long long src = zaddr;
        unsigned char* dst;
        if (d < PINS) {
            const int rr = d * RPD + rsub;
            const int q = slot ^ row_key<CIN>(rr);
            const int pr = rr / PW, pc = rr - pr * PW;
            const int iy = iy0 + pr, ix = ix0 + pc;
            if (rr < PROWS && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && 8 * q < a.Cin)
                src = (unsigned long long)(a.in + (((long)b * a.H + iy) * a.W + ix) * a.in_cs + 8 * q);
            dst = pbuf + d * RPD * RB;
        } else {
            const int wr = (d - PINS) * RPD + rsub;       // row index over [tap][BN]
            const int tap = wr / BN, n = wr - tap * BN;
            const int q = slot ^ row_key<CIN>(n);
            if (n0 + n < a.Cout && 8 * q < a.Cin)
                src = (unsigned long long)(a.w + ((long)(n0 + n) * 9 + tap) * a.Cin + 8 * q);
            dst = wbuf + (d - PINS) * RPD * RB;
        }
        dma16(src, dst);
    }

    f32x16 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ni][mi][e] = 0.0f;

    const int frow = lane & 31, fq = lane >> 5;
    int rr0[MI], woff[NI], wkey[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int p = wm * TM + mi * 32 + frow;           // output pixel inside the tile
        rr0[mi] = (p / TPW) * S * PW + (p % TPW) * S;     // patch row of its (kh=0,kw=0) tap
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int n = wn * TN + ni * 32 + frow;
        woff[ni] = n * RB;
        wkey[ni] = row_key<CIN>(n);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int kh = tap / 3, kw = tap - kh * 3;
        int aoff[MI], akey[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int rr = rr0[mi] + kh * PW + kw;
            aoff[mi] = rr * RB;
            akey[mi] = row_key<CIN>(rr);
        }
        const unsigned char* wt = wbuf + tap * BN * RB;
#pragma unroll
        for (int kk = 0; kk < CIN / 16; ++kk) {
            bf16x8 wf[NI], af[MI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                wf[ni] = *reinterpret_cast<const bf16x8*>(wt + woff[ni] + (((kk * 2 + fq) ^ wkey[ni]) << 4));
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                af[mi] = *reinterpret_cast<const bf16x8*>(pbuf + aoff[mi] + (((kk * 2 + fq) ^ akey[mi]) << 4));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
        }
    }
    __syncthreads();                                       // all fragment reads done: LDS becomes the output tile

    unsigned short* Cs = reinterpret_cast<unsigned short*>(smem);
    auto convert = [&](auto silu_tag) {                     // compile-time activation, packed fp32 math (yolo_internal.h)
        constexpr bool kSilu = decltype(silu_tag)::value;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int nl = wn * TN + ni * 32 + 8 * qd + 4 * (lane >> 5);
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
    if (a.act == ADAYOLO_ACT_SILU) convert(std::true_type{});
    else convert(std::false_type{});
    __syncthreads();
    constexpr int CPR = BN / 8;
    // fixed trip count -> fully unrolled, so all residual loads / LDS reads are in flight before the first store
#pragma unroll
    for (int idx = tid; idx < BM * CPR; idx += kThreads) {
        const int ml = idx / CPR, ch = (idx - ml * CPR) * 8;
        const int oy = oy0 + ml / TPW, ox = ox0 + ml % TPW, n = n0 + ch;
        if (oy >= a.Ho || ox >= a.Wo || n >= a.Cout) continue;
        const long m = ((long)b * a.Ho + oy) * a.Wo + ox;
        u32x4 v = *reinterpret_cast<const u32x4*>(Cs + ml * CP + ch);
        if (a.res) {
            const u32x4 r = *reinterpret_cast<const u32x4*>(a.res + m * a.res_cs + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float lo = bf16_to_f32((unsigned short)(v[j] & 0xFFFFu)) + bf16_to_f32((unsigned short)(r[j] & 0xFFFFu));
                const float hi = bf16_to_f32((unsigned short)(v[j] >> 16)) + bf16_to_f32((unsigned short)(r[j] >> 16));
                v[j] = pack_bf16x2(lo, hi);
            }
        }
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(a.out + m * a.out_cs + n));
    }
}

template <int CIN, int BN, int S, int TPH, int TPW, int WM, int WN>
static hipError_t launch(ConvArgs a, hipStream_t s) {
    constexpr int PH = (TPH - 1) * S + 3, PW = (TPW - 1) * S + 3, RPD = 64 / (CIN / 8);
    constexpr int pbytes = ((PH * PW + RPD - 1) / RPD) * RPD * CIN * 2;
    constexpr int ring = pbytes + 9 * BN * CIN * 2, epi = TPH * TPW * (BN + 8) * 2;
    constexpr int smem = (ring > epi ? ring : epi) + BN * 4;
    static_assert(smem <= 160 * 1024, "LDS budget");
    auto kern = k_conv3x3_small<CIN, BN, S, TPH, TPW, WM, WN>;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        configured = true;
    }
    const int tiles_x = (a.Wo + TPW - 1) / TPW, tiles_y = (a.Ho + TPH - 1) / TPH;
    const int ntn = (a.Cout + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3(a.B * tiles_y * tiles_x * ntn), dim3(64 * WM * WN), smem, s, a, tiles_x, tiles_y);
    return hipGetLastError();
}

}  // namespace smallk

// variant 40: whole-K-resident kernel for Cin in {32, 64}, 3x3, stride 1 or 2 (hipErrorInvalidValue otherwise).
hipError_t launch_conv_small(ConvArgs a, hipStream_t s, int variant) {
    using namespace smallk;
    (void)variant;
    if (a.ks != 3 || (a.Cin != 32 && a.Cin != 64)) return hipErrorInvalidValue;
    if (a.Cin == 32) {
        if (a.stride == 1) return launch<32, 64, 1, 8, 32, 4, 2>(a, s);   // 256 px x 64 ch, 8 waves, 58 KB
        return launch<32, 64, 2, 4, 32, 4, 1>(a, s);                      // 128 px, 4 waves, 75 KB
    }
    if (a.stride == 1) return launch<64, 64, 1, 8, 32, 4, 2>(a, s);       // 256 px x 64 ch: 44 + 74 KB
    return launch<64, 64, 2, 4, 32, 4, 1>(a, s);                          // 75 + 74 KB
}

}  // namespace adayolo
