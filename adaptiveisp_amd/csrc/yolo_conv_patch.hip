// 3x3 stride-1 Conv + bias + SiLU (+ residual) with the INPUT PATCH resident in LDS.
//
// The implicit-GEMM kernels (yolo_conv_dma*.hip) fetch every activation 9 times from L2/MALL — once per
// filter tap — and the ablations showed the LDS-DMA side (bytes in flight / latency), not the MFMAs, sets their
// ceiling. Here a workgroup owns an 8x32-pixel output tile: its (8+2)x(32+2) input patch (up to 128 channels
// per pass) is DMA'd into LDS ONCE and all nine taps read their operand fragments from it at shifted rows; only
// the weights keep streaming through a small ring (they are shared by every workgroup, hence L2-hot).
// Operand bytes per flop drop ~1.7-2x against the 256x256 implicit GEMM and ~4x against 128x128.
//
//   workgroup = 8 waves (4 along pixels x 2 along channels), tile 256 px x BN channels, v_mfma_f32_32x32x16_bf16
//   LDS: patch rows (one pixel x CKP channels each, XOR-swizzled 16-B chunks) + weight ring (BN x 64 ch stages)
//   k order: channel pass (CKP) -> tap (9) -> 64-channel sub-step
#include "yolo_internal.h"

namespace adayolo {
namespace patch {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

constexpr int TPH = 8, TPW = 32, BM = TPH * TPW;       // output tile
constexpr int PW2 = TPW + 2, PH2 = TPH + 2;            // patch with the 1-px halo
constexpr int PROWS = PH2 * PW2;                       // 340 patch pixels
constexpr int NW = 8, WM = 4, WN = 2, kThreads = 64 * NW;

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

// CKP: channels per pass held in the patch (64 or 128); BN: output channels per workgroup; NSW: weight ring stages
template <int CKP, int BN, int NSW>
__global__ __launch_bounds__(kThreads) void k_conv3x3_patch(const ConvArgs a, int tiles_x, int tiles_y) {
    constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;       // wave tile 64 px x TN ch
    constexpr int PCH = CKP / 8;                    // 16-byte chunks per patch row
    constexpr int PRB = CKP * 2;                    // bytes per patch row
    constexpr int RPI = 64 / PCH;                   // patch rows per DMA instruction
    constexpr int NPI = (PROWS + RPI * NW - 1) / (RPI * NW);   // patch DMA instructions per wave
    constexpr int PALLOC = NPI * NW * RPI * PRB;    // patch bytes in LDS
    constexpr int WSTAGE = BN * 128;                // weight stage: BN rows x 64 channels
    constexpr int WI = BN / (8 * NW);               // weight DMA instructions per wave per stage
    constexpr int SUB = CKP / 64;                   // 64-channel sub-steps per tap
    constexpr int CP = BN + 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* pbuf = smem;
    unsigned char* wring = smem + PALLOC;
    constexpr int kRing = PALLOC + NSW * WSTAGE, kEpi = BM * CP * 2;
    float* bias_s = reinterpret_cast<float*>(smem + (kRing > kEpi ? kRing : kEpi));   // [BN]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int ntn = (a.Cout + BN - 1) / BN;
    const int lid = xcd_remap(blockIdx.x, a.B * tiles_y * tiles_x * ntn);
    const int nt = lid % ntn;
    int t = lid / ntn;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int x0 = tx * TPW, y0 = ty * TPH, n0 = nt * BN;
    const unsigned long long zaddr = (unsigned long long)(const void*)g_zero16;
    for (int i = tid; i < BN; i += kThreads) bias_s[i] = (n0 + i < a.Cout) ? a.bias[n0 + i] : 0.0f;   // once, coalesced

    // ---- patch DMA state: instruction j of this wave covers patch rows RPI*(wave*NPI + j) .. +RPI-1 -----------
    const int pslot = lane % PCH, prsub = lane / PCH;
    unsigned long long prow[NPI];
    int pcap[NPI];                                  // valid channels from this lane's chunk (0 = outside the image)
#pragma unroll
    for (int j = 0; j < NPI; ++j) {
        const int rr = RPI * (wave * NPI + j) + prsub;
        const int q = pslot ^ (CKP == 128 ? (rr & 15) : ((rr >> 1) & 7));
        const int pr = rr / PW2, pc = rr - pr * PW2;
        const int iy = y0 - 1 + pr, ix = x0 - 1 + pc;
        const bool ok = rr < PROWS && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        prow[j] = (unsigned long long)(a.in + (ok ? (((long)b * a.H + iy) * a.W + ix) * a.in_cs + 8 * q : 0));
        pcap[j] = ok ? a.Cin - 8 * q : 0;
    }
    // ---- weight DMA state (rows of 64 channels = 8 chunks, key (r>>1)&7) --------------------------------------
    const int wslot = lane & 7, wrsub = lane >> 3;
    const int Ktot = 9 * a.Cin;
    unsigned long long wrow[WI];
    int wcap[WI];
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int r = WI * 8 * wave + 8 * i + wrsub;
        const int q = wslot ^ ((r >> 1) & 7);
        const bool ok = (n0 + r) < a.Cout;
        wrow[i] = (unsigned long long)(a.w + (ok ? (long)(n0 + r) * Ktot + 8 * q : 0));
        wcap[i] = ok ? a.Cin - 8 * q : 0;
    }
    const int npass = (a.Cin + CKP - 1) / CKP;
    const int steps_per_pass = 9 * SUB;
    const int nsteps = npass * steps_per_pass;

    auto issue_patch = [&](int pass) {
        const int c0 = pass * CKP;
#pragma unroll
        for (int j = 0; j < NPI; ++j)
            dma16(sel(c0 < pcap[j], prow[j] + 2 * c0, zaddr), pbuf + (wave * NPI + j) * RPI * PRB);
    };
    auto issue_w = [&](int step) {                 // global step index over (pass, tap, sub)
        const int pass = step / steps_per_pass, rem = step - pass * steps_per_pass;
        const int tap = rem / SUB, sub = rem - tap * SUB;
        const int c0 = pass * CKP + sub * 64;
        const long woff = 2 * ((long)tap * a.Cin + c0);
        unsigned char* st = wring + (step % NSW) * WSTAGE;
#pragma unroll
        for (int i = 0; i < WI; ++i)
            dma16(sel(c0 < wcap[i], wrow[i] + woff, zaddr), st + (WI * 8 * wave + 8 * i) * 128);
    };

    f32x16 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ni][mi][e] = 0.0f;

    // fragment rows: pixel p -> patch row of its (kh=0,kw=0) tap; weight rows as in the implicit-GEMM kernels
    const int frow = lane & 31, fq = lane >> 5;
    int rr0[MI], wrow_off[NI], wkey[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int p = wm * TM + mi * 32 + frow;
        rr0[mi] = (p / TPW) * PW2 + (p % TPW);
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int R = wn * TN + ni * 32 + frow;
        wrow_off[ni] = R * 128;
        wkey[ni] = (R >> 1) & 7;
    }

    int step = 0;
    for (int pass = 0; pass < npass; ++pass) {
        // every wave is done with the previous pass' patch and ring -> refill the patch, restart the ring.
        // DMA queue order per wave: [patch] [W step 0] ... [W step NSW-2], so the counted wait for W step 0 below
        // also covers the (older) patch loads.
        wait_vm_and_barrier<0>();
        issue_patch(pass);
#pragma unroll
        for (int s = 0; s < NSW - 1; ++s)
            if (s < steps_per_pass) issue_w(step + s);
        for (int ps = 0; ps < steps_per_pass; ++ps, ++step) {
            if (ps + (NSW - 2) < steps_per_pass) wait_vm_and_barrier<WI * (NSW - 2)>();
            else wait_vm_and_barrier<0>();
            if (ps + NSW - 1 < steps_per_pass) issue_w(step + NSW - 1);      // its slot was read at step-1: free now
            const int tap = ps / SUB, sub = ps - tap * SUB;
            const int kh = tap / 3, kw = tap - kh * 3;
            const unsigned char* wst = wring + (step % NSW) * WSTAGE;
            int aoff[MI], akey[MI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int rr = rr0[mi] + kh * PW2 + kw;
                aoff[mi] = rr * PRB;
                akey[mi] = CKP == 128 ? (rr & 15) : ((rr >> 1) & 7);
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                bf16x8 wf[NI], af[MI];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    wf[ni] = *reinterpret_cast<const bf16x8*>(wst + wrow_off[ni] + (((kk * 2 + fq) ^ wkey[ni]) << 4));
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    af[mi] = *reinterpret_cast<const bf16x8*>(pbuf + aoff[mi] + (((sub * 8 + kk * 2 + fq) ^ akey[mi]) << 4));
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
            }
        }
    }
    wait_vm_and_barrier<0>();

    // ---- epilogue (same transposition scheme as the implicit-GEMM kernels; pixels map back to (y,x)) ------------
    unsigned short* Cs = reinterpret_cast<unsigned short*>(smem);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int nl = wn * TN + ni * 32 + 8 * qd + 4 * (lane >> 5);
            const float4 b4 = *reinterpret_cast<const float4*>(bias_s + nl);
            const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = acc[ni][mi][4 * qd + i] + bv[i];
                    if (a.act == ADAYOLO_ACT_SILU) v[i] = silu(v[i]);
                }
                const int ml = wm * TM + mi * 32 + (lane & 31);
                *reinterpret_cast<u32x2*>(Cs + ml * CP + nl) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            }
        }
    }
    __syncthreads();
    constexpr int CPR = BN / 8;
    // fixed trip count -> fully unrolled, so all residual loads / LDS reads are in flight before the first store
#pragma unroll
    for (int idx = tid; idx < BM * CPR; idx += kThreads) {
        const int ml = idx / CPR, ch = (idx - ml * CPR) * 8;
        const int oy = y0 + ml / TPW, ox = x0 + ml % TPW, n = n0 + ch;
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
        *reinterpret_cast<u32x4*>(a.out + m * a.out_cs + n) = v;
    }
}

template <int CKP, int BN, int NSW>
static hipError_t launch(ConvArgs a, hipStream_t s) {
    constexpr int PCH = CKP / 8, RPI = 64 / PCH, NPI = (PROWS + RPI * NW - 1) / (RPI * NW);
    constexpr int ring = NPI * NW * RPI * CKP * 2 + NSW * BN * 128, epi = BM * (BN + 8) * 2;
    constexpr int smem = (ring > epi ? ring : epi) + BN * 4;
    static_assert(smem <= 160 * 1024, "LDS budget");
    auto kern = k_conv3x3_patch<CKP, BN, NSW>;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        configured = true;
    }
    const int tiles_x = (a.Wo + TPW - 1) / TPW, tiles_y = (a.Ho + TPH - 1) / TPH;
    const int ntn = (a.Cout + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3(a.B * tiles_y * tiles_x * ntn), dim3(kThreads), smem, s, a, tiles_x, tiles_y);
    return hipGetLastError();
}

}  // namespace patch

// variants 30..: patch-resident 3x3 stride-1 kernels. Returns hipErrorInvalidValue for shapes they do not serve.
hipError_t launch_conv_patch(ConvArgs a, hipStream_t s, int variant) {
    using namespace patch;
    if (a.ks != 3 || a.stride != 1) return hipErrorInvalidValue;
    switch (variant) {
        case 30: return launch<128, 256, 2>(a, s);     // 87 KB patch + 64 KB ring
        case 31: return launch<64, 256, 2>(a, s);      // 48 KB patch + 64 KB ring
        case 32: return launch<128, 128, 3>(a, s);     // 87 KB patch + 48 KB ring
        case 33: return launch<64, 128, 3>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace adayolo
