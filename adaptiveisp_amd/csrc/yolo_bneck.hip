// Whole Bottleneck in ONE launch:  out = x + SiLU(b2 + W2 (3x3) * SiLU(b1 + W1 (1x1) * x))      C = 256, hidden 128
// (yolov3/models/common.py:110-120, Bottleneck.forward = x + cv2(cv1(x)); the eight blocks of the C = 256 stage).
//
// What the pair [3x3 + residual | next block's 1x1] (yolo_conv_pp.hip, FUSE) leaves on the table, by its own stamps
// (profiles/round4_conv_pp_dephase_stamps.txt, 77.9k cycles per 256-px tile): the residual tile arrives cold from HBM in the
// epilogue (13.5k cycles against 6.4k without one), the hidden tensor h goes to memory and comes back (2 x 30 MB of the
// launch's 181 MB), and half of the k-loop's LDS-DMA stream re-stages h nine times, once per tap (the k-loop runs at 85 % of
// matrix issue on the 27 B/clk a CU's DMA path sustains beside its MFMAs). Fused the other way round — the block's OWN 1x1 in
// front of its 3x3 — all three go away:
//   * tile = 16 x 16 output pixels of one image; stage A computes h on the 18 x 18 patch the 3x3 needs (324 px, +27 % of a
//     layer that is 1/9 of the flops) straight into LDS: h never exists in HBM (launch traffic 181 -> 136 MB);
//   * stage B is the ping-pong k-loop of yolo_conv_pp.hip with the activation operand read from that patch — the tap is an
//     address offset — so only the weights stream (2 LDS-DMA instructions per wave and k-tile pair instead of 8);
//   * the residual is x, which this workgroup read a few microseconds earlier for stage A: an L2 / Infinity-Cache hit.
//
// LDS (150 KB):  stage A ring 2 x {x slice 384 rows x 128 B, W1 slice 128 rows x 128 B} = 128 KB, overlaid afterwards by
//                h patch [2 chunks of 64 ch][324 px][128 B] = 81 KB  +  W2 ring 2 x 32 KB; epilogue staging overlays all.
// Rows of 128 B hold 64 channels in eight 16-byte slots, slot = chunk ^ key: key = (row >> 1) & 7 for the DMA-staged tiles
// (consecutive rows per fragment), key = (patch column >> 1) & 7 for h — a fragment's 32 pixels are two patch rows of 16
// consecutive columns, and ds_read_b128's lane groups take eight pixels of each: sixteen consecutive columns, sixteen
// distinct 16-byte bank groups, for every tap.
// Numerics: h is rounded to bf16 exactly as the stand-alone 1x1 layer stores it; both GEMMs accumulate in fp32 over k in the
// order the stand-alone kernels use.
#include "yolo_internal.h"
#include <type_traits>
#include <cstdlib>

namespace adayolo {
namespace bnk {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};
#ifdef ADAYOLO_MEASURE
__device__ unsigned long long g_stamp[4096 * 8];
#define BN_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_stamp[blockIdx.x * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define BN_STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
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
__device__ __forceinline__ void barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int C = 256, CH = 128;             // block channels, hidden channels
constexpr int TS = 16, PS = TS + 2, NP = PS * PS;      // tile side, patch side, patch pixels (324)
constexpr int kRow = 128;                    // bytes per LDS row: 64 bf16 channels
constexpr int MA = 384;                      // stage A rows (324 padded to 12 fragments of 32)
constexpr int kXs = MA * kRow;               // x slice of one k-tile: 48 KB
constexpr int kBufA = kXs + CH * kRow;       // + W1 slice (128 rows): 64 KB
constexpr int kHc = NP * kRow;               // one 64-channel chunk of the h patch: 41,472 B
constexpr int kH = 2 * kHc;                  // 82,944
constexpr int kWt = C * kRow;                // one W2 k-tile: 256 rows x 128 B = 32 KB
constexpr int kWring = kH;                   // W2 ring offset
constexpr int kEpiPitch = 144;
constexpr int kEpi = 8 * 128 * kEpiPitch;    // 147,456: epilogue staging, overlays h + ring
constexpr int kSmem = kWring + 2 * kWt;      // 148,480 (>= 2 * kBufA = 131,072 and >= kEpi)
constexpr int NXI = (NP + 7) / 8;            // 41 LDS-DMA instructions stage the 324 patch rows of a k-tile

struct BneckArgs {
    const unsigned short* x; int x_cs;
    const unsigned short* w1; const float* b1;       // [128][256], [128]
    const unsigned short* w2; const float* b2;       // [256][3][3][128], [256]
    unsigned short* out; int out_cs;
    int B, H, W, tiles_x, tiles_y;
};

template <int ABL>
__global__ __launch_bounds__(512) void k_bneck(const BneckArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    BN_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntile = a.B * a.tiles_y * a.tiles_x;
    const int lid = xcd_remap(blockIdx.x, ntile);
    const int b = lid / (a.tiles_y * a.tiles_x), trem = lid - b * (a.tiles_y * a.tiles_x);
    const int y0 = (trem / a.tiles_x) * TS, x0 = (trem % a.tiles_x) * TS;
    const unsigned long long zaddr = (unsigned long long)(const void*)g_zero16;
    const int slot = lane & 7, rsub = lane >> 3;

    // =================================== stage A: h = SiLU(W1 x + b1) on the 18 x 18 patch ===================================
    // DMA state. x rows: instruction g = wave + 8 i covers patch pixels 8 g .. 8 g + 7 (g < 41); W1 rows: g = 2 wave + j.
    unsigned long long xrow[6];
    bool xok[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int r = 8 * (wave + 8 * i) + rsub;
        const int pr = r / PS, pc = r - pr * PS;
        const int gy = y0 - 1 + pr, gx = x0 - 1 + pc;
        xok[i] = r < NP && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        const int q = slot ^ ((r >> 1) & 7);
        xrow[i] = (unsigned long long)(a.x + ((long)(b * a.H + (xok[i] ? gy : 0)) * a.W + (xok[i] ? gx : 0)) * a.x_cs + 8 * q);
    }
    unsigned long long w1row[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 8 * (2 * wave + j) + rsub;
        const int q = slot ^ ((r >> 1) & 7);
        w1row[j] = (unsigned long long)(a.w1 + (long)r * C + 8 * q);
    }
    auto stage_a = [&](int kt, unsigned char* buf) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int g = wave + 8 * i;
            if (g < NXI) dma16(sel(xok[i], xrow[i] + 128 * kt, zaddr), buf + g * 1024);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) dma16(w1row[j] + 128 * kt, buf + kXs + (2 * wave + j) * 1024);
    };
    stage_a(0, smem);
    stage_a(1, smem + kBufA);

    // wave grid 4 (pixel quarters of 96 rows) x 2 (channel halves of 64): six 32 x 32 accumulators
    const int mq = wave >> 1, nh = wave & 1;
    const int frow = lane & 31, fq = lane >> 5, key = (frow >> 1) & 7;
    f32x16 ha[2][3];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 3; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) ha[ni][mi][e] = 0.0f;
    BN_STAMP(1);
#pragma unroll 1
    for (int kt = 0; kt < 4; ++kt) {
        // this wave's share of k-tile kt has landed when only the instructions of k-tile kt + 1 are outstanding
        if (kt < 3) { if (wave == 0) wait_vm<8>(); else wait_vm<7>(); }
        else wait_vm<0>();
        barrier();
        const unsigned char* buf = smem + (kt & 1) * kBufA;
        if (ABL != 2) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int ko = ((2 * kk + fq) ^ key) << 4;
                bf16x8 wf[2], xf[3];
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) wf[ni] = *reinterpret_cast<const bf16x8*>(buf + kXs + (nh * 64 + ni * 32 + frow) * kRow + ko);
#pragma unroll
                for (int mi = 0; mi < 3; ++mi) xf[mi] = *reinterpret_cast<const bf16x8*>(buf + (mq * 96 + mi * 32 + frow) * kRow + ko);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int mi = 0; mi < 3; ++mi)
                        ha[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni], xf[mi], ha[ni][mi], 0, 0, 0);
            }
        }
        barrier();                                        // every wave has read the buffer: re-stage it
        if (kt + 2 < 4) stage_a(kt + 2, smem + (kt & 1) * kBufA);
    }
    BN_STAMP(2);

    // =================================== stage B set-up: weights of the 3x3 start streaming ===================================
    constexpr int nK = 18;                               // 9 taps x 2 chunks of 64 hidden channels
    unsigned long long wrow[4];
    int wlds[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int h = i >> 1, g = 2 * wave + (i & 1);
        const int rb = (g >> 2) * 64 + h * 32 + (g & 3) * 8, r = rb + rsub;
        const int q = slot ^ ((r >> 1) & 7);
        wrow[i] = (unsigned long long)(a.w2 + (long)r * (9 * CH) + 8 * q);
        wlds[i] = rb * kRow;
    }
    unsigned char* const wring = smem + kWring;
    auto stage_w = [&](int h, unsigned char* buf, int t, bool live) {
#pragma unroll
        for (int j = 0; j < 2; ++j) dma16(sel(live, wrow[2 * h + j] + 128 * t, zaddr), buf + wlds[2 * h + j]);
    };
    // (the W2 ring overlaps the second stage-A buffer, dead behind the loop's last barrier; it does not touch the h region)
    stage_w(0, wring, 0, true);
    stage_w(1, wring, 0, true);
    stage_w(0, wring + kWt, 1, true);

    // ---- stage A epilogue: bias, SiLU, zero outside the image (the 3x3 pads h, not x), bf16 -> h patch
    {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            float4 bq[4];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) bq[qd] = *reinterpret_cast<const float4*>(a.b1 + nh * 64 + ni * 32 + 8 * qd + 4 * fq);
#pragma unroll
            for (int mi = 0; mi < 3; ++mi) {
                const int m = mq * 96 + mi * 32 + frow;
                const int pr = m / PS, pc = m - pr * PS;
                const int gy = y0 - 1 + pr, gx = x0 - 1 + pc;
                const bool inside = m < NP && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                const int hkey = (pc >> 1) & 7;
                // channels nh * 64 + ni * 32 + 8 qd + 4 fq + (0..3) -> chunk nh, 16-byte slot 4 ni + qd, 8-byte half fq
                unsigned char* const dst = smem + nh * kHc + m * kRow + 8 * fq;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    unsigned lo, hi;
                    bias_act_pack4<true>(ha[ni][mi][4 * qd], ha[ni][mi][4 * qd + 1], ha[ni][mi][4 * qd + 2], ha[ni][mi][4 * qd + 3],
                                         bq[qd], lo, hi);
                    if (!inside) { lo = 0u; hi = 0u; }
                    if (m < NP) *reinterpret_cast<u32x2*>(dst + (((4 * ni + qd) ^ hkey) << 4)) = u32x2{lo, hi};
                }
            }
        }
    }
    BN_STAMP(3);

    // =================================== stage B: 3x3 over the h patch, ping-pong wave groups ===================================
    const int wm = wave >> 2, wn = wave & 3;
    f32x16 acc[2][4];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ni][mi][e] = 0.0f;
    const int tx = frow & 15, tyl = frow >> 4;
    const int abase = ((wm * 8 + tyl) * PS + tx) * kRow;             // + f * 2 * PS * kRow per fragment, + tap, + chunk
    const int wbase = (wn * 64 + frow) * kRow;
    int wko[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) wko[kk] = ((2 * kk + fq) ^ key) << 4;

    wait_vm<2>();                                        // W2 k-tile 0 landed (this wave's share); W0 of k-tile 1 in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // h rows written
    barrier();

    bf16x8 af[2][4], wx[4], wy[4];
    auto read_a = [&](int t, int half) {
        if (ABL == 2) return;
        const int tap = t >> 1, kh = tap / 3, kw = tap - 3 * kh;
        const int hk = ((tx + kw) >> 1) & 7;
        const unsigned char* base = smem + (t & 1) * kHc + abase + (kh * PS + kw) * kRow;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                af[mi][kk] = *reinterpret_cast<const bf16x8*>(base + (2 * half + mi) * (2 * PS * kRow) + (((2 * kk + fq) ^ hk) << 4));
    };
    auto read_w = [&](const unsigned char* buf, int half, bf16x8 (&w)[4]) {
        if (ABL == 2) return;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) w[kk] = *reinterpret_cast<const bf16x8*>(buf + wbase + half * 32 * kRow + wko[kk]);
    };
    // MFMA section of one phase; STAGE: the phase's two weight pieces are issued in the shadow of its first MFMAs.
    // WAIT >= 0: counted wait at the end (the pieces issued two phases ago have landed; readable two phases on)
    auto mma = [&](int ni, int half, const bf16x8 (&w)[4], auto stage_tag, unsigned long long g0, unsigned char* d0,
                   unsigned long long g1, unsigned char* d1) {
        constexpr bool STAGE = decltype(stage_tag)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                if (ABL != 2)
                    acc[ni][2 * half + mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[kk], af[mi][kk], acc[ni][2 * half + mi], 0, 0, 0);
                const int n = 2 * kk + mi;
                if (STAGE && (n == 0 || n == 3)) {
                    __builtin_amdgcn_sched_barrier(0);
                    dma16(n == 0 ? g0 : g1, n == 0 ? d0 : d1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (!STAGE) wait_vm<2>();
    };
    read_w(wring, 0, wx);                                // W0 of k-tile 0
    if (wm == 1) barrier();                              // stagger group 1 by one barrier

    // Schedule (yolo_conv_pp.hip without the activation pieces):
    //   P1(t): reads A(t, px 0)          stages W1(t+1)         P3(t): reads A(t, px 1)       stages W0(t+2)
    //   P2(t): reads W1(t)               waits: W0(t+1) landed  P4(t): reads W0(t+1)          waits: W1(t+1) landed
    auto ktile = [&](unsigned char* cur, unsigned char* oth, int t, bf16x8 (&w0)[4], bf16x8 (&w1)[4]) {
        const bool live1 = t + 1 < nK, live2 = t + 2 < nK;
        unsigned long long g0, g1;
        read_a(t, 0);
        g0 = sel(live1, wrow[2] + 128 * (t + 1), zaddr); g1 = sel(live1, wrow[3] + 128 * (t + 1), zaddr);
        asm volatile("" : "+v"(g0), "+v"(g1));
        barrier();
        mma(0, 0, w0, std::true_type{}, g0, oth + wlds[2], g1, oth + wlds[3]);
        barrier();
        read_w(cur, 1, w1);
        barrier();
        mma(1, 0, w1, std::false_type{}, 0, nullptr, 0, nullptr);
        barrier();
        read_a(t, 1);
        g0 = sel(live2, wrow[0] + 128 * (t + 2), zaddr); g1 = sel(live2, wrow[1] + 128 * (t + 2), zaddr);
        asm volatile("" : "+v"(g0), "+v"(g1));
        barrier();
        mma(1, 1, w1, std::true_type{}, g0, cur + wlds[0], g1, cur + wlds[1]);
        barrier();
        read_w(oth, 0, w1);
        barrier();
        mma(0, 1, w0, std::false_type{}, 0, nullptr, 0, nullptr);
        barrier();
    };
#pragma unroll 1
    for (int t = 0; t < nK; t += 2) {
        ktile(wring, wring + kWt, t, wx, wy);
        ktile(wring + kWt, wring, t + 1, wy, wx);
    }
    BN_STAMP(4);
    if (wm == 0) barrier();                              // pairs with group 1's last barrier
    wait_vm<0>();                                        // the tail's zero-page pieces target the ring the epilogue overlays
    barrier();

    // =================================== epilogue: bias, SiLU, + x, store ===================================
    // D[row = channel][col = pixel]; each wave transposes its 128 px x 64 ch through a private LDS region (pitch 144 B) and
    // writes whole 128-byte row segments. Wave pixel r = r0 + 8 j (r0 = lane >> 3): tile row wm * 8 + (j >> 1), column
    // r0 + 8 (j & 1).
    unsigned char* my = smem + wave * (128 * kEpiPitch);
    {
        float4 bq[2][4];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) bq[ni][qd] = *reinterpret_cast<const float4*>(a.b2 + wn * 64 + ni * 32 + 8 * qd + 4 * fq);
        const int chunk = lane & 7, r0 = lane >> 3;
        const int n = wn * 64 + chunk * 8;
        unsigned char* const wr = my + frow * kEpiPitch + 8 * fq;
        const unsigned char* const rd = my + r0 * kEpiPitch + chunk * 16;
        const long pix0 = (long)(b * a.H + y0 + wm * 8) * a.W + x0 + r0;
        const bool colok0 = x0 + r0 < a.W, colok1 = x0 + r0 + 8 < a.W;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            u32x4 v[4], r[4];
            bool ok[4];
            long pix[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int j = 4 * mi + it;                       // pixel r0 + 8 j of the wave
                ok[it] = (y0 + wm * 8 + (j >> 1) < a.H) && ((j & 1) ? colok1 : colok0);
                pix[it] = pix0 + (long)(j >> 1) * a.W + 8 * (j & 1);
                r[it] = u32x4{0u, 0u, 0u, 0u};
                if (ok[it]) r[it] = *reinterpret_cast<const u32x4*>(a.x + pix[it] * a.x_cs + n);
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    unsigned lo, hi;
                    bias_act_pack4<true>(acc[ni][mi][4 * qd], acc[ni][mi][4 * qd + 1], acc[ni][mi][4 * qd + 2], acc[ni][mi][4 * qd + 3],
                                         bq[ni][qd], lo, hi);
                    *reinterpret_cast<u32x2*>(wr + mi * 32 * kEpiPitch + (ni * 32 + 8 * qd) * 2) = u32x2{lo, hi};
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same wave wrote and reads: in-order LDS, no barrier
#pragma unroll
            for (int it = 0; it < 4; ++it) v[it] = *reinterpret_cast<const u32x4*>(rd + (mi * 32 + it * 8) * kEpiPitch);
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x2_pk s = f32x2_pk{__uint_as_float(v[it][j] << 16), __uint_as_float(v[it][j] & 0xFFFF0000u)} +
                                       f32x2_pk{__uint_as_float(r[it][j] << 16), __uint_as_float(r[it][j] & 0xFFFF0000u)};
                    v[it][j] = pack_bf16x2(s.x, s.y);
                }
#pragma unroll
            for (int it = 0; it < 4; ++it)
                if (ok[it]) __builtin_nontemporal_store(v[it], reinterpret_cast<u32x4*>(a.out + pix[it] * a.out_cs + n));
        }
    }
    BN_STAMP(5);
}

template <int ABL>
static hipError_t launch(BneckArgs a, hipStream_t s) {
    static_assert(kSmem <= 160 * 1024 && kSmem >= 2 * kBufA && kSmem >= kEpi, "LDS budget");
    auto kern = k_bneck<ABL>;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kSmem);
        if (e != hipSuccess) return e;
        configured = true;
    }
    a.tiles_x = (a.W + TS - 1) / TS;
    a.tiles_y = (a.H + TS - 1) / TS;
    hipLaunchKernelGGL(kern, dim3(a.B * a.tiles_x * a.tiles_y), dim3(512), kSmem, s, a);
    return hipGetLastError();
}

}  // namespace bnk

hipError_t launch_bottleneck256(const void* x, int x_cs, const void* w1, const float* b1, const void* w2, const float* b2,
                                void* out, int out_cs, int B, int H, int W, hipStream_t s) {
    bnk::BneckArgs a;
    a.x = static_cast<const unsigned short*>(x); a.x_cs = x_cs;
    a.w1 = static_cast<const unsigned short*>(w1); a.b1 = b1;
    a.w2 = static_cast<const unsigned short*>(w2); a.b2 = b2;
    a.out = static_cast<unsigned short*>(out); a.out_cs = out_cs;
    a.B = B; a.H = H; a.W = W; a.tiles_x = a.tiles_y = 0;
#ifdef ADAYOLO_MEASURE
    if (getenv("ADAYOLO_BNECK_NOMFMA")) return bnk::launch<2>(a, s);
#endif
    return bnk::launch<0>(a, s);
}

#ifdef ADAYOLO_MEASURE
extern "C" int adayolo_debug_bneck_stamps(unsigned long long* dst, int n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(bnk::g_stamp), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif

}  // namespace adayolo
