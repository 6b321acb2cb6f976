// Conv + bias + SiLU (+ residual) — implicit GEMM, LDS-DMA pipelined variant for gfx950.
//
// Same math and operand roles as yolo_conv.hip (weights = MFMA A operand, pixels = B operand,
// v_mfma_f32_16x16x32_bf16, fp32 accumulate, LDS-transposed bf16 epilogue). What changes is how the
// operand tiles reach LDS:
//
//   * global -> LDS goes through the LDS-DMA path (global_load_lds_dwordx4): no staging VGPRs, no
//     ds_write pass. One wave instruction lands 64 x 16 B = 8 tile rows of 128 B (BK = 64 bf16).
//     The implicit-GEMM gather (tap offset, zero padding, M/N tails) lives entirely in the per-lane
//     SOURCE address; out-of-range lanes read a 16-byte zero block.
//   * LDS is a ring of STAGES tiles; loads run STAGES-1 k-steps ahead of the MFMAs and are retired with
//     a counted s_waitcnt vmcnt(N) (never 0 in steady state) + ONE raw s_barrier per k-step.
//   * The LDS image is linear per wave instruction (DMA requirement), so the bank-conflict fix is an XOR
//     swizzle applied on both sides: lane (row r, slot j) fetches k-chunk j ^ ((r >> 1) & 7) and fragment
//     reads look up chunk c at slot c ^ ((r >> 1) & 7): the 16 rows of a ds_read_b128 group then cover
//     all sixteen 16-B slots of the 256-B bank row.
#include "yolo_internal.h"

namespace adayolo {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

namespace dma {

constexpr int kThreads = 256;
constexpr int BK = 64;                 // bf16 per tile row = 128 B = 8 chunks of 16 B

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

__device__ __forceinline__ void dma16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 16, 0, 0);
}

template <int BM, int BN, int WM, int WN, int STAGES>
__global__ __launch_bounds__(kThreads) void k_conv_igemm_dma(const ConvArgs a) {
    static_assert(WM * WN == 4, "4 waves");
    constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 16, NI = TN / 16;
    constexpr int AI = BM / 32, WI = BN / 32;            // DMA instructions per wave per stage (8 rows each)
    constexpr int kStageBytes = (BM + BN) * BK * 2;
    constexpr int CP = BN + 8;
    static_assert(BM * CP * 2 <= STAGES * kStageBytes, "epilogue tile must fit in the ring");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int lid = xcd_remap(blockIdx.x, a.mtiles * a.ntiles);
    const int m0 = (lid / a.ntiles) * BM, n0 = (lid % a.ntiles) * BN;

    // ---- DMA roles: this lane owns slot (lane & 7) of row 32*wave + 8*i + (lane >> 3) ----------------
    const int slot = lane & 7, rsub = lane >> 3;
    int hi0[AI], wi0[AI], qa[AI];
    long abase[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int r = AI * 8 * wave + 8 * i + rsub;          // tile row
        qa[i] = slot ^ ((r >> 1) & 7);                        // k-chunk this lane fetches for that row
        const int m = m0 + r;
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
    long wbase[WI];
    int qw[WI];
    bool wok[WI];
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int r = WI * 8 * wave + 8 * i + rsub;
        qw[i] = slot ^ ((r >> 1) & 7);
        wok[i] = (n0 + r) < a.Cout;
        wbase[i] = (long)(n0 + r) * Ktot;
    }
    const int cpt = (a.Cin + BK - 1) / BK;
    const int nsteps = a.ks * a.ks * cpt;

    auto issue = [&](int step) {
        unsigned char* st = smem + (step % STAGES) * kStageBytes;
        const int tap = step / cpt, c0 = (step - tap * cpt) * BK;
        const int kh = tap / a.ks, kw = tap - kh * a.ks;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int c = c0 + 8 * qa[i];
            const int hi = hi0[i] + kh, wi = wi0[i] + kw;
            const bool ok = c < a.Cin && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
            const void* src = ok ? (const void*)(a.in + abase[i] + ((long)hi * a.W + wi) * a.in_cs + c)
                                 : (const void*)g_zero16;
            dma16(src, st + (AI * 8 * wave + 8 * i) * (BK * 2));         // wave-uniform LDS base; lane -> +16*lane
        }
        unsigned char* sw = st + BM * BK * 2;
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int c = c0 + 8 * qw[i];
            const bool ok = wok[i] && c < a.Cin;
            const void* src = ok ? (const void*)(a.w + wbase[i] + (long)tap * a.Cin + c) : (const void*)g_zero16;
            dma16(src, sw + (WI * 8 * wave + 8 * i) * (BK * 2));
        }
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addressing: row R, chunk c -> byte R*128 + ((c ^ ((R>>1)&7)) << 4)
    const int frow = lane & 15, fq = lane >> 4;
    int arow_off[MI], akey[MI], wrow_off[NI], wkey[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int R = wm * TM + mi * 16 + frow;
        arow_off[mi] = R * (BK * 2);
        akey[mi] = (R >> 1) & 7;
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int R = wn * TN + ni * 16 + frow;
        wrow_off[ni] = BM * BK * 2 + R * (BK * 2);
        wkey[ni] = (R >> 1) & 7;
    }

    constexpr int PER = AI + WI;                // DMA instructions per wave per stage
    // prologue: fill STAGES-1 stages
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nsteps) issue(s);

    for (int step = 0; step < nsteps; ++step) {
        // my DMA for `step` has landed once at most (STAGES-2) later stages are still in flight
        if (step + (STAGES - 2) < nsteps) wait_vm_and_barrier<PER * (STAGES - 2)>();
        else wait_vm_and_barrier<0>();
        // everyone is past step-1: its stage is free -> refill it with step + STAGES - 1
        if (step + STAGES - 1 < nsteps) issue(step + STAGES - 1);
        const unsigned char* st = smem + (step % STAGES) * kStageBytes;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            bf16x8 wf[NI], af[MI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                wf[ni] = *reinterpret_cast<const bf16x8*>(st + wrow_off[ni] + (((kk * 4 + fq) ^ wkey[ni]) << 4));
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                af[mi] = *reinterpret_cast<const bf16x8*>(st + arow_off[mi] + (((kk * 4 + fq) ^ akey[mi]) << 4));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
        }
    }
    wait_vm_and_barrier<0>();      // all waves done reading the ring before the epilogue reuses it

    unsigned short* Cs = reinterpret_cast<unsigned short*>(smem);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int nl = wn * TN + ni * 16 + (lane >> 4) * 4;
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
    constexpr int CPR = BN / 8;
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

template <int BM, int BN, int WM, int WN, int STAGES>
static hipError_t launch(ConvArgs a, hipStream_t s) {
    constexpr int smem = STAGES * (BM + BN) * BK * 2;
    auto kern = k_conv_igemm_dma<BM, BN, WM, WN, STAGES>;
    static bool configured = false;          // idempotent attribute set (benign if raced)
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        configured = true;
    }
    a.mtiles = (a.M + BM - 1) / BM;
    a.ntiles = (a.Cout + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3(a.mtiles * a.ntiles), dim3(kThreads), smem, s, a);
    return hipGetLastError();
}

}  // namespace dma

// The library default (variant 2): 2-stage ring, 64 KB, 2 workgroups/CU; tile shape by Cout / grid size.
hipError_t launch_conv_dma(ConvArgs a, hipStream_t s, int variant) {
    using namespace dma;
    (void)variant;
    const long blocks128 = (long)((a.M + 127) / 128) * ((a.Cout + 127) / 128);
    if (a.Cout <= 32) return launch<128, 32, 4, 1, 2>(a, s);
    if (a.Cout <= 64) return launch<128, 64, 4, 1, 2>(a, s);
    if (blocks128 < 512) return launch<64, 128, 2, 2, 2>(a, s);
    return launch<128, 128, 2, 2, 2>(a, s);
}

}  // namespace adayolo
