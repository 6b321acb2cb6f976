// Conv + bias + SiLU (+ residual) — implicit GEMM, 256 px x 256 ch tile, two wave groups in ping-pong.
//
// yolo_conv_dma2.hip runs all eight waves of a workgroup in lock-step (barrier -> issue DMA -> LDS reads -> MFMA):
// while the waves read their fragments the matrix pipes idle, and with one look-ahead stage every k-step ends in
// vmcnt(0). This kernel keeps its tile / swizzle / lean DMA addressing and changes the schedule:
//
//   * wave grid 2 (px) x 4 (ch), wave tile 128 px x 64 ch = four 64x32 quadrants. A k-tile (BK = 64) is four phases,
//     one quadrant x full BK each (8 x v_mfma_f32_32x32x16_bf16 = 256 matrix-pipe cycles per phase per wave);
//   * the two wave groups (waves 0-3 / 4-7: one wave of each per SIMD) are staggered by one barrier, so on every
//     SIMD one wave's MFMA section runs beside the other wave's load section (ds_read_b128 fragment reads + the
//     LDS-DMA issue for one half-tile) — the matrix pipe always has a client;
//   * LDS = 2 k-tile buffers x {A-h0, A-h1, W-h0, W-h1} half-tiles of 16 KB (128 rows x 128 B). A half-tile holds
//     the rows ONE quadrant index touches in every wave, so a slot is dead as soon as that phase's reads retired
//     and is re-staged two phases later (the WAR distance the stagger needs), 4-5 phases before its data is read:
//         P1(t): reads A0        stages W1(t+1)      P3(t): reads A1        stages W0(t+2)
//         P2(t): reads W1        stages A1(t+1)      P4(t): reads W0(t+1)   stages A0(t+2)
//     The two DMA instructions of a phase are issued INSIDE the MFMA section (after the first two MFMAs): issuing
//     them costs the wave 60-180 cycles each, which in the load section made that section longer than the partner's
//     256-cycle MFMA section. Every MFMA section ends with one counted `s_waitcnt vmcnt(6)` (three half-tiles stay
//     in flight) — never vmcnt(0). A wait in phase X's MFMA section retires data that is first read in phase X+2's
//     load section (with the stagger, only then has every wave passed a barrier behind every other wave's wait).
//
// Restrictions (the launcher falls back otherwise): Cin % 64 == 0, Cout % 256 == 0.
// Measured and dropped (round 2): channel chunk outer / tap inner k order, so that consecutive k-tiles ask for almost the same
// activation lines (L1 hits instead of L2 fetches): 3-4 % SLOWER on every layer; a non-temporal / system-scope cache policy on
// the weight DMA (so that weights do not push those lines out): no change.
// And: the four waves of a group issuing their two pieces behind DIFFERENT MFMAs of a section (wave w behind the w-th and
// (w+4)-th, a scalar compare + branch per slot) instead of all behind the 1st and the 4th: 78 vs 70 us — the extra states between
// MFMAs cost more than the texture-address queue they were meant to spare.
#include "yolo_internal.h"
#include "yolo_chain.h"
#ifndef PP_PRIO_MODE
#define PP_PRIO_MODE 0      // 0: s_setprio 1 around every MFMA section (default); 1: no priority; 2: static priority for the second wave group (measurement builds)
#endif
#include <type_traits>
#include <cstdlib>
#ifdef ADAYOLO_PLAIN_STORES   // A/B switch (measurement): keep the output lines in the XCD L2 instead of streaming them
#define ADAYOLO_STORE(v, p) (*(p) = (v))
#else
#define ADAYOLO_STORE(v, p) __builtin_nontemporal_store((v), (p))
#endif

#ifndef PP_FUSE_HOIST
#define PP_FUSE_HOIST 0      // weight fragments of the fused second layer requested ahead of the first layer's epilogue
#endif

namespace adayolo {
namespace pp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};
#ifdef ADAYOLO_MEASURE
__device__ unsigned long long g_stamp[4096 * 8];     // ABL 7: per-workgroup s_memtime stamps (measurement build)
#else
__device__ unsigned long long g_stamp[8];
#endif
#ifdef ADAYOLO_CHAIN_STAMPS
// measurement build of the persistent chain: thread 0 adds the cycles since its previous stamp to slot k of 16 LDS accumulators
// (behind the scheduler words); the chain kernel adds them to its global accumulators when the workgroup leaves
#define PP_STAMP(k) do { if (ABL == 7 && threadIdx.x == 0 && blockIdx.x < 4096) g_stamp[blockIdx.x * 8 + (k)] = __builtin_readcyclecounter(); \
                         if (CHAIN && threadIdx.x == 0) { unsigned long long* acc_ = reinterpret_cast<unsigned long long*>(smem + kChainSchedOff + 32); \
                             const unsigned long long now_ = __builtin_readcyclecounter(); acc_[(k)] += now_ - acc_[15]; acc_[15] = now_; } } while (0)
#else
#define PP_STAMP(k) do { if (ABL == 7 && threadIdx.x == 0 && blockIdx.x < 4096) g_stamp[blockIdx.x * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#endif

__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {      // round-to-nearest-even: v_cvt_pk_bf16_f32
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
__device__ __forceinline__ float silu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
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

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int kRow = BK * 2;                 // bytes per tile row
constexpr int kTile = 256 * kRow;            // one operand k-tile: 32 KB
constexpr int kBuf = 2 * kTile;              // A + W of one k-tile
constexpr int kEpiPitch = 144;                // bytes per pixel row of a wave's private epilogue region (64 ch + pad)
constexpr int kEpi = 8 * 128 * kEpiPitch;     // 8 waves x 128 px: 144 KB, overlays the (finished) ring
constexpr int kSmem = (kEpi > 2 * kBuf ? kEpi : 2 * kBuf) + BN * 4;   // + bias

// wave-uniform state of the k-tile a stage call addresses
struct KPos {
    int c0, kh, kw, tap;
    long aoff, woff;
};

// ABL: 0 real kernel, 1 no DMA in the loop, 2 no LDS reads / MFMA, 3 no epilogue stores, 4 no k-loop, 5 no DMA
// instructions in the loop, 6 no epilogue, 8 / 9 activation lines for one kernel column in three / one tap in nine
// (measurement builds)
// FUSE: the tile holds ALL 256 output channels of its 256 pixels, so the 1x1 conv that consumes this layer's output
// (Bottleneck.cv1 of the next block, 256 -> 128, HBM-bound on its own: it re-reads 60 MB that were just written) is
// applied to the output tile while it sits in LDS: the epilogue puts the post-residual bf16 rows back into the waves'
// regions, one barrier, then every wave computes 128 px x 32 ch of the second layer — its 32 weight rows (K = 256) are
// loaded once into registers as the MFMA's channel operand (fragment-major copy of the weights, see below), the pixel
// operand is read from the tile (pitch 144 B:
// conflict-free ds_read_b128) — and writes it through its own region again. The second layer sees exactly the bf16 values
// the unfused kernel would read back from memory.
// One 256 x 256 tile. CHAIN: `lid` is handed in, output stores are written through (sc1) and this workgroup's previous tile is
// published once they are known complete; wave 0 fetches the next item and checks its inputs in the shadow of the epilogue.
template <int ABL, bool FUSE, bool CHAIN>
__device__ __forceinline__ void conv_tile(const ConvArgs& a, const int lid, unsigned char* smem, ChainCtx& cx) {
    float* bias_s = reinterpret_cast<float*>(smem + (kEpi > 2 * kBuf ? kEpi : 2 * kBuf));
    PP_STAMP(0);
#ifdef PP_DEPHASE            // measurement build: every other workgroup of an XCD starts PP_DEPHASE cycles late
    if (!CHAIN && ((blockIdx.x >> 3) & 1)) {
        const unsigned long long t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < (unsigned long long)(PP_DEPHASE)) __builtin_amdgcn_s_sleep(16);
    }
#endif

    int tid_ = threadIdx.x;
    // (CHAIN: opaque per tile, so that nothing derived from the thread index is hoisted out of the persistent loop and then
    // spilled across the tile — the tile body alone uses 253 of 256 registers; recomputing a dozen integers per tile is free)
    if (CHAIN) asm volatile("" : "+v"(tid_));
    const int tid = tid_, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;           // wm is also the ping-pong group
    const int m0 = (lid / a.ntiles) * BM, n0 = (lid % a.ntiles) * BN;
    // CHAIN: wave 0 looks ahead while the tile runs (yolo_chain.h: ChainLook)
    ChainLook look;
    auto sched_stage = [&](int stage) {
        if (CHAIN && wave == 0) look.stage(stage, *cx.c, smem + kChainSchedOff, lane);
    };
    const unsigned long long zaddr = (unsigned long long)(const void*)g_zero16;

    // ---- per-row DMA state: one DMA instruction moves 8 tile rows (64 lanes x 16 B); a half-tile is 16 of them,
    //      two per wave. Index i = 2*half + j.
    const int slot = lane & 7, rsub = lane >> 3;
    unsigned long long arow[4], wrow[4];
    unsigned amask[4];
    int alds[4], wlds[4];                                // wave-uniform LDS byte offsets inside an operand tile
    // weights first: their addresses need no division, so their DMA can be in flight while the pixel rows are decoded
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int h = i >> 1, g = 2 * wave + (i & 1);
        // weight rows of half h: the first / second 32 channels of each wave column's 64
        const int rb = (g >> 2) * 64 + h * 32 + (g & 3) * 8, r = rb + rsub;
        const int q = slot ^ ((r >> 1) & 7);
        wrow[i] = (unsigned long long)(a.w + (long)(n0 + r) * (a.ks * a.ks * a.Cin) + 8 * q);
        wlds[i] = kTile + rb * kRow;
    }
    auto decode_rows = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int h = i >> 1, g = 2 * wave + (i & 1);
            // activation rows of half h: the first (h=0) / second (h=1) 64 px of each wave row's 128
            const int rb = (g >> 3) * 128 + h * 64 + (g & 7) * 8, r = rb + rsub;
            const int q = slot ^ ((r >> 1) & 7);
            const int m = m0 + r;
            unsigned mask = 0;
            long off = 0;
            if (m < a.M) {
                // m -> (image, row, column) by multiply-high (the two runtime divisions cost ~80 VALU each otherwise)
                const int b = a.sh_hw < 0 ? m : (int)(__umulhi((unsigned)m, a.magic_hw) >> a.sh_hw);
                const int rem = m - b * (a.Ho * a.Wo);
                const int ho = a.sh_w < 0 ? rem : (int)(__umulhi((unsigned)rem, a.magic_w) >> a.sh_w);
                const int wo = rem - ho * a.Wo;
                const int hi0 = ho * a.stride - a.pad, wi0 = wo * a.stride - a.pad;
                unsigned vw = 0;                               // tap validity is separable: rows x columns
                // ks is 1 or 3 (checked at the ABI): three straight-line taps, no loop
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) vw |= (unsigned)(kw < a.ks && wi0 + kw >= 0 && wi0 + kw < a.W) << kw;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
                    mask |= (kh < a.ks && hi0 + kh >= 0 && hi0 + kh < a.H) ? vw << (kh * a.ks) : 0u;
                off = ((long)b * a.H * a.W + (long)hi0 * a.W + wi0) * a.in_cs + 8 * q;
            }
            amask[i] = mask;
            arow[i] = (unsigned long long)(a.in + off);
            alds[i] = rb * kRow;
        }
    };
    const int cpt = a.Cin / BK;
    const int nK = a.ks * a.ks * cpt;

    auto advance = [&](KPos& p) {
        p.c0 += BK;
        if (p.c0 >= a.Cin) {
            p.c0 = 0; ++p.tap; ++p.kw;
            if (p.kw == a.ks) { p.kw = 0; ++p.kh; }
        }
        p.aoff = 2 * (((long)p.kh * a.W + p.kw) * a.in_cs + p.c0);
        p.woff = 2 * ((long)p.tap * a.Cin + p.c0);
    };
    // one DMA instruction (j = 0 / 1) of a half-tile
    auto stage_a1 = [&](int h, int j, unsigned char* buf, const KPos& p, bool live) {
        if (ABL == 5 && !live) return;               // measurement build: no DMA instruction at all in the k-loop
        const int i = 2 * h + j;
        const bool ok = live && ((amask[i] >> p.tap) & 1u);
        dma16(sel(ok, arow[i] + p.aoff, zaddr), buf + alds[i]);
    };
    auto stage_w1 = [&](int h, int j, unsigned char* buf, const KPos& p, bool live) {
        if (ABL == 5 && !live) return;
        const int i = 2 * h + j;
        dma16(sel(live, wrow[i] + p.woff, zaddr), buf + wlds[i]);
    };
    // the same two pieces with the source address computed ahead of time (in the load section: ~10 SALU/VALU
    // instructions per piece that otherwise sit between two MFMAs of the issuing wave and overrun the 32-cycle slot)
    auto addr_a1 = [&](int h, int j, const KPos& p, bool live) {
        const int i = 2 * h + j;
        // ABL 8 / 9 (measurement): the activation pieces of one kernel column in three / one tap in nine fetch real lines, the
        // others the zero page — the distinct-line traffic a per-kernel-row activation strip / a whole patch in LDS would leave
        if (ABL == 8 && p.kw != 0) live = false;
        if (ABL == 9 && p.tap != 0) live = false;
        return sel(live && ((amask[i] >> p.tap) & 1u), arow[i] + p.aoff, zaddr);
    };
    auto addr_w1 = [&](int h, int j, const KPos& p, bool live) { return sel(live, wrow[2 * h + j] + p.woff, zaddr); };
    // (Measured and dropped: letting the twelve "dead" pieces a wave issues in a tile's last two k-tiles fetch residual
    // rows instead of the zero page, as an L2 prefetch for the epilogue — with a residual the epilogue is 14.9k cycles
    // instead of 6.4k, the 128 KB tile arriving cold at the per-CU streaming rate. The two extra 64-bit adds per piece
    // in the load sections cost the k-loop 44.0k -> 57.8k cycles and bought the epilogue 1.3k.)
    auto stage_a = [&](int h, unsigned char* buf, const KPos& p, bool live) { stage_a1(h, 0, buf, p, live); stage_a1(h, 1, buf, p, live); };
    auto stage_w = [&](int h, unsigned char* buf, const KPos& p, bool live) { stage_w1(h, 0, buf, p, live); stage_w1(h, 1, buf, p, live); };

    f32x16 acc[2][4];                                   // [channel frag][pixel frag]
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ni][mi][e] = 0.0f;

    // fragment addressing (32x32x16): lane -> tile row (lane & 31), 16-byte k-chunk 2*kk + (lane >> 5), XOR key
    // (row >> 1) & 7 — the same for every fragment of this lane because fragment origins are multiples of 32 rows
    const int frow = lane & 31, fq = lane >> 5, key = (frow >> 1) & 7;
    const int abase = (wm * 128 + frow) * kRow, wbase = kTile + (wn * 64 + frow) * kRow;
    int koff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koff[kk] = ((2 * kk + fq) ^ key) << 4;

    PP_STAMP(1);
    // ---- prologue: bias (one 1 KB DMA by wave 0), k-tile 0 complete, W0 / A0 of k-tile 1 (what P3 / P4 of the
    //      preceding k-tiles would have staged). Issue order = retire order: the 8 instructions of k-tile 0 come first.
    KPos p0{0, 0, 0, 0, 0, 0};
    KPos p1 = p0;
    advance(p1);
    if (wave == 0) dma16((unsigned long long)(a.bias + n0) + 16 * lane, bias_s);
    stage_w(0, smem, p0, true);
    stage_w(1, smem, p0, true);
    decode_rows();
    stage_a(0, smem, p0, true);
    stage_a(1, smem, p0, true);
    stage_w(0, smem + kBuf, p1, 1 < nK);
    stage_a(0, smem + kBuf, p1, 1 < nK);
    // (Measured and dropped: touching the tile's whole input region into L2 here, one 4-byte LDS-DMA per 128-byte line — L2-
    // resident data streams into LDS at 51.6 B/clk/CU for any row stride (tools/dma_pattern.hip) while the k-loop moves 27,
    // and about a quarter of its activation rows are first touches. The k-loop did not move: 44.1k -> 43.9k cycles,
    // prologue + 1.0k. First-touch latency is not what holds it.)
    if (CHAIN) {
        // ... and every store of this workgroup's PREVIOUS tile is written through: its arrival can be published (one lane,
        // behind the barrier: every wave has drained)
        wait_vm<0>();
        barrier();
        chain_publish(cx, tid);
        sched_stage(0);
    } else {
        wait_vm<4>();                                    // k-tile 0 landed (this wave's share)
        barrier();
    }

    bf16x8 af[2][4], wx[4], wy[4];
    auto read_a = [&](const unsigned char* buf, int half) {
        if (ABL == 2) return;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                af[mi][kk] = *reinterpret_cast<const bf16x8*>(buf + abase + (2 * half + mi) * 32 * kRow + koff[kk]);
    };
    auto read_w = [&](const unsigned char* buf, int half, bf16x8 (&w)[4]) {
        if (ABL == 2) return;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            w[kk] = *reinterpret_cast<const bf16x8*>(buf + wbase + half * 32 * kRow + koff[kk]);
    };
    // MFMA section of one phase: 8 MFMAs with the phase's two LDS-DMA pieces issued in their shadow, then the counted
    // wait that retires the half-tile issued three phases ago (readable from the load section two phases on)
    auto mma = [&](int ni, int half, const bf16x8 (&w)[4], unsigned long long g0, unsigned char* d0, unsigned long long g1,
                   unsigned char* d1) {
#if PP_PRIO_MODE == 0
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                if (ABL != 2)
                    acc[ni][2 * half + mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[kk], af[mi][kk], acc[ni][2 * half + mi], 0, 0, 0);
                const int n = 2 * kk + mi;
                if (n == 0 || n == 3) {              // one DMA instruction behind the 1st and the 4th MFMA
                    __builtin_amdgcn_sched_barrier(0);
                    if (ABL != 5) dma16(n == 0 ? g0 : g1, n == 0 ? d0 : d1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
#if PP_PRIO_MODE == 0
        __builtin_amdgcn_s_setprio(0);
#endif
        wait_vm<6>();
    };
    PP_STAMP(2);
    read_w(smem, 0, wx);                                 // W0 of k-tile 0
#if PP_PRIO_MODE == 2      // measurement: static priority for the second wave group, no flips around the MFMA sections
    if (wm == 1) __builtin_amdgcn_s_setprio(1);
#endif
    if (wm == 1) barrier();                              // stagger group 1 by one barrier

    KPos p2 = p1;
    // w0 / w1: the two weight-fragment register sets; they swap roles every k-tile (P4 loads the next W0 into the
    // set whose W1 died in P3)
    auto ktile = [&](unsigned char* cur, unsigned char* oth, int t, bf16x8 (&w0)[4], bf16x8 (&w1)[4]) {
        const bool live1 = (ABL != 1 && ABL != 5) && t + 1 < nK, live2 = (ABL != 1 && ABL != 5) && t + 2 < nK;
        // P1: quadrant (px half 0, ch half 0); stages W1(t+1)
        unsigned long long g0, g1;
        read_a(cur, 0);
        g0 = addr_w1(1, 0, p1, live1); g1 = addr_w1(1, 1, p1, live1);
        asm volatile("" : "+v"(g0), "+v"(g1));          // (pins the address math to this side of the barrier)
        barrier();
        mma(0, 0, w0, g0, oth + wlds[2], g1, oth + wlds[3]);
        barrier();
        // P2: (px 0, ch 1); stages A1(t+1)
        read_w(cur, 1, w1);
        g0 = addr_a1(1, 0, p1, live1); g1 = addr_a1(1, 1, p1, live1);
        asm volatile("" : "+v"(g0), "+v"(g1));
        barrier();
        mma(1, 0, w1, g0, oth + alds[2], g1, oth + alds[3]);
        barrier();
        // P3: (px 1, ch 1); stages W0(t+2)
        advance(p2);
        read_a(cur, 1);
        g0 = addr_w1(0, 0, p2, live2); g1 = addr_w1(0, 1, p2, live2);
        asm volatile("" : "+v"(g0), "+v"(g1));
        barrier();
        mma(1, 1, w1, g0, cur + wlds[0], g1, cur + wlds[1]);
        barrier();
        // P4: (px 1, ch 0); the load section fetches W0 of the NEXT k-tile; stages A0(t+2)
        read_w(oth, 0, w1);
        g0 = addr_a1(0, 0, p2, live2); g1 = addr_a1(0, 1, p2, live2);
        asm volatile("" : "+v"(g0), "+v"(g1));
        barrier();
        mma(0, 1, w0, g0, cur + alds[0], g1, cur + alds[1]);
        barrier();
        p1 = p2;
    };
    for (int t = 0; t < (ABL == 4 ? 0 : nK); t += 2) {
        ktile(smem, smem + kBuf, t, wx, wy);
        if (t + 1 < nK) ktile(smem + kBuf, smem, t + 1, wy, wx);
        else asm volatile("" ::"v"(wy[0]));
    }
    PP_STAMP(3);
    if (wm == 0) barrier();                              // pairs with group 1's last barrier
    wait_vm<0>();                                        // the tail's zero-fill DMAs target the ring the epilogue overlays
    barrier();

    if (ABL == 6) {                                      // measurement build: prologue + k-loop only
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) asm volatile("" ::"v"(acc[ni][mi]));
        return;
    }
    // ---- epilogue. D[row = channel][col = pixel]: lane holds pixel (lane & 31) and channels 8*qd + 4*(lane >> 5) + (0..3)
    //      of every 32x32 fragment. Each wave transposes its own 128 px x 64 ch through a PRIVATE LDS region (pitch
    //      144 B: 16-byte aligned rows, 2-way write conflicts at most) — no workgroup barrier, a wave's stores leave
    //      as soon as its own fragment is converted — and writes whole 128-byte row segments (8 lanes x 16 B).
    //      (Storing 8/16-byte pieces straight from the fragment layout was measured 2x slower: 32 rows x 32 B per
    //      instruction instead of 8 rows x 128 B.)
    unsigned char* my = smem + wave * (128 * kEpiPitch);
    // CHAIN: the tile's outputs leave as 16-byte WRITE-THROUGH (sc1) buffer stores — complete, for every other CU and XCD, once
    // the storing wave's vmcnt reaches 0 (no release fence, i.e. no whole-L2 write-back); byte offsets are 32-bit (the caller
    // checks the tensors are < 2 GB)
    constexpr int kSc1 = 16;
    __amdgpu_buffer_rsrc_t rs_out, rs_out2;
    if (CHAIN) {
        rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, 0x7FFFFFFF, 0x00020000);
        if (FUSE) rs_out2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.out2, 0, 0x7FFFFFFF, 0x00020000);
    }
    // The activation and the residual are compile-time copies (no per-element select, no per-chunk branch); bias and
    // row pointers are set up once; per 32-pixel group the four LDS reads, the four residual loads and the four stores
    // are issued back to back (one wait each), not read -> wait -> load -> wait -> store four times over.
    auto epilogue = [&](auto silu_tag, auto res_tag) {
        constexpr bool kSilu = decltype(silu_tag)::value, kRes = decltype(res_tag)::value;
        float4 bq[2][4];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
                bq[ni][qd] = *reinterpret_cast<const float4*>(bias_s + wn * 64 + ni * 32 + 8 * qd + 4 * (lane >> 5));
        const int chunk = lane & 7, r0 = lane >> 3;
        const int mrow = m0 + wm * 128 + r0, n = n0 + wn * 64 + chunk * 8;
        unsigned short* const op = a.out + (long)mrow * a.out_cs + n;
        const unsigned short* const rp = kRes ? a.res + (long)mrow * a.res_cs + n : nullptr;
        const long ostep = 8L * a.out_cs, rstep = kRes ? 8L * a.res_cs : 0;
        unsigned char* const wr = my + (lane & 31) * kEpiPitch + 8 * (lane >> 5);
        const unsigned char* const rd = my + r0 * kEpiPitch + chunk * 16;
        // (With a residual this epilogue is 14.9k cycles instead of 6.4k: the 128 KB residual tile arrives cold from HBM at the
        // per-CU streaming rate. Measured and dropped: requesting the rows 2 or 4 groups ahead instead of one — no change, the
        // stream is rate-bound, not latency-bound — and starting the accumulators at the bias to free the registers for
        // that — the k-loop lost 2-3k cycles to the changed register allocation.)
        const int obyte = CHAIN ? (int)(((long)mrow * a.out_cs + n) * 2) : 0;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            if (mi == 1 || mi == 2) sched_stage(mi);             // (stage 0: behind the prologue; 3: after the tile's last stores)
            u32x4 v[4], r[4];
            bool ok[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) ok[it] = mrow + 8 * (4 * mi + it) < a.M;
            if (kRes) {                                          // in flight while this group's SiLUs are computed
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    r[it] = u32x4{0u, 0u, 0u, 0u};
                    if (ok[it]) r[it] = *reinterpret_cast<const u32x4*>(rp + (4 * mi + it) * rstep);
                }
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    unsigned lo, hi;
                    bias_act_pack4<kSilu>(acc[ni][mi][4 * qd], acc[ni][mi][4 * qd + 1], acc[ni][mi][4 * qd + 2], acc[ni][mi][4 * qd + 3],
                                          bq[ni][qd], lo, hi);
                    *reinterpret_cast<u32x2*>(wr + mi * 32 * kEpiPitch + (ni * 32 + 8 * qd) * 2) = u32x2{lo, hi};
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same wave wrote and reads: in-order LDS, no barrier
#pragma unroll
            for (int it = 0; it < 4; ++it) v[it] = *reinterpret_cast<const u32x4*>(rd + (mi * 32 + it * 8) * kEpiPitch);
            if (kRes) {
#pragma unroll
                for (int it = 0; it < 4; ++it)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x2_pk x = f32x2_pk{__uint_as_float(v[it][j] << 16), __uint_as_float(v[it][j] & 0xFFFF0000u)} +
                                           f32x2_pk{__uint_as_float(r[it][j] << 16), __uint_as_float(r[it][j] & 0xFFFF0000u)};
                        v[it][j] = pack_bf16x2(x.x, x.y);
                    }
            }
#pragma unroll
            for (int it = 0; it < 4; ++it)
                if (ok[it] && !(ABL == 3 && v[it][0] != 0x12345678u)) {
                    if (CHAIN) __builtin_amdgcn_raw_buffer_store_b128(v[it], rs_out, obyte + (4 * mi + it) * (int)(2 * ostep), 0, kSc1);
                    else ADAYOLO_STORE(v[it], reinterpret_cast<u32x4*>(op + (4 * mi + it) * ostep));
                }
            if (FUSE) {                                          // the rows the second layer reads: post-residual
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    *reinterpret_cast<u32x4*>(const_cast<unsigned char*>(rd) + (mi * 32 + it * 8) * kEpiPitch) = v[it];
            }
        }
    };
    // fused second layer: half of its weight fragments (32 registers: the k-loop's fragment registers are dead by now) are
    // requested BEFORE the first layer's epilogue, which covers their latency
    const int pm = wave >> 2, cq = wave & 3;
    bf16x8 w2f[16];
    float4 b2q[4];
    if (FUSE) {
        // w2 is stored fragment-major by the caller ([4 cq][16 kk][64 lanes][8]: lane (r, fq) of step kk holds
        // w2[32 cq + r][16 kk + 8 fq .. + 8]): a wave's load is 1 KB contiguous. Read straight from the [128][256] matrix the
        // same 16 loads touch 32 cache lines each — 4096 line lookups per workgroup on the CU's one texture-address path,
        // measured +21 us per launch instead of +8.
        // (the first half only: all sixteen next to the 128 accumulators of the epilogue spill; the second half is
        // requested behind the epilogue and lands under the first eight steps of the second layer)
        const unsigned short* wr2 = a.w2 + ((long)cq * 16 * 64 + lane) * 8;
#pragma unroll
        for (int kk = 0; kk < PP_FUSE_HOIST; ++kk) w2f[kk] = *reinterpret_cast<const bf16x8*>(wr2 + kk * 64 * 8);
    }
    if (a.act == ADAYOLO_ACT_SILU) {
        if (a.res) epilogue(std::true_type{}, std::true_type{});
        else epilogue(std::true_type{}, std::false_type{});
    } else {
        if (a.res) epilogue(std::false_type{}, std::true_type{});
        else epilogue(std::false_type{}, std::false_type{});
    }
    PP_STAMP(6);
    if (ABL == 7 && !FUSE) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PP_STAMP(7); }
    if (FUSE) {
        // ---- second layer: out2[px][n] = SiLU(bias2[n] + sum_k y[px][k] * w2[n][k]), k < 256, n < 128.
        //      Wave -> pixel half pm (128 px = the regions of waves 4 pm .. 4 pm + 3, 64 channels of k each) x channel
        //      quarter cq (32 ch). D[row = channel][col = pixel] as in the main loop.
        {
            const unsigned short* wr2 = a.w2 + ((long)cq * 16 * 64 + lane) * 8;
#pragma unroll
            for (int kk = PP_FUSE_HOIST; kk < 16; ++kk) w2f[kk] = *reinterpret_cast<const bf16x8*>(wr2 + kk * 64 * 8);
        }
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) b2q[qd] = *reinterpret_cast<const float4*>(a.bias2 + cq * 32 + 8 * qd + 4 * (lane >> 5));
        f32x16 acc2[4];
#pragma unroll
        for (int pf = 0; pf < 4; ++pf)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[pf][e] = 0.0f;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's rows are in LDS
        barrier();                                           // ... and every other wave's
        PP_STAMP(4);
        const unsigned char* ybase = smem + pm * 4 * (128 * kEpiPitch) + (lane & 31) * kEpiPitch + 16 * (lane >> 5);
        bf16x8 yf[2][4];                                     // pixel fragments of step kk / kk + 1
#pragma unroll
        for (int pf = 0; pf < 4; ++pf) yf[0][pf] = *reinterpret_cast<const bf16x8*>(ybase + pf * 32 * kEpiPitch);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            if (kk + 1 < 16) {
                const unsigned char* yk = ybase + ((kk + 1) >> 2) * (128 * kEpiPitch) + ((kk + 1) & 3) * 32;
#pragma unroll
                for (int pf = 0; pf < 4; ++pf) yf[(kk + 1) & 1][pf] = *reinterpret_cast<const bf16x8*>(yk + pf * 32 * kEpiPitch);
            }
#pragma unroll
            for (int pf = 0; pf < 4; ++pf)
                acc2[pf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2f[kk], yf[kk & 1][pf], acc2[pf], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        barrier();                                           // every wave has read the tile: the regions are free again
        PP_STAMP(5);
        {
            unsigned char* const wr = my + (lane & 31) * kEpiPitch + 8 * (lane >> 5);
#pragma unroll
            for (int pf = 0; pf < 4; ++pf)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    unsigned lo, hi;
                    bias_act_pack4<true>(acc2[pf][4 * qd], acc2[pf][4 * qd + 1], acc2[pf][4 * qd + 2], acc2[pf][4 * qd + 3], b2q[qd], lo, hi);
                    *reinterpret_cast<u32x2*>(wr + pf * 32 * kEpiPitch + 8 * qd * 2) = u32x2{lo, hi};
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same wave wrote and reads
            // 128 px x 64 B: a lane takes 16 B (8 channels) of pixel (lane >> 2) + 16 it
            const int c4 = lane & 3, r0 = lane >> 2;
            const int mrow = m0 + pm * 128 + r0;
            unsigned short* const op2 = a.out2 + (long)mrow * a.out2_cs + cq * 32 + c4 * 8;
            u32x4 v2[8];
#pragma unroll
            for (int it = 0; it < 8; ++it) v2[it] = *reinterpret_cast<const u32x4*>(my + (r0 + 16 * it) * kEpiPitch + c4 * 16);
            const int obyte2 = CHAIN ? (int)(((long)mrow * a.out2_cs + cq * 32 + c4 * 8) * 2) : 0;
#pragma unroll
            for (int it = 0; it < 8; ++it)
                if (mrow + 16 * it < a.M) {
                    if (CHAIN) __builtin_amdgcn_raw_buffer_store_b128(v2[it], rs_out2, obyte2 + 16 * it * a.out2_cs * 2, 0, kSc1);
                    else ADAYOLO_STORE(v2[it], reinterpret_cast<u32x4*>(op2 + (long)(16 * it) * a.out2_cs));
                }
        }
        if (ABL == 7) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PP_STAMP(7); }
    }
    if (CHAIN) {
        PP_STAMP(8);                                         // (chain stamps: 6/5 -> here = the tile's last stores issued)
        sched_stage(3);
        PP_STAMP(11);
        barrier();                                           // the tile's LDS is free; {next item, ready} is in place
        PP_STAMP(9);
    }
}

template <int ABL, bool FUSE>
__global__ __launch_bounds__(512) void k_conv_pp(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ChainCtx none{nullptr, -1};
    conv_tile<ABL, FUSE, false>(a, xcd_remap(blockIdx.x, a.mtiles * a.ntiles), smem, none);
}

#ifndef ADAYOLO_TILE_ONLY
template <int ABL, bool FUSE = false>
static hipError_t launch(ConvArgs a, hipStream_t s) {
    static_assert(kSmem <= 160 * 1024, "LDS budget");
    auto kern = k_conv_pp<ABL, FUSE>;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kSmem);
        if (e != hipSuccess) return e;
        configured = true;
    }
    a.mtiles = (a.M + BM - 1) / BM;
    a.ntiles = a.Cout / BN;
    hipLaunchKernelGGL(kern, dim3(a.mtiles * a.ntiles), dim3(512), kSmem, s, a);
    return hipGetLastError();
}

#endif  // ADAYOLO_TILE_ONLY

}  // namespace pp

#ifndef ADAYOLO_TILE_ONLY

// variant 50 = the kernel; with -DADAYOLO_MEASURE 51..57 = the measurement builds (ABL above). hipErrorInvalidValue ->
// the shape is not served (the caller falls back to the default kernel).
hipError_t launch_conv_pp(ConvArgs a, hipStream_t s, int variant) {
    if (a.Cin % 64 || a.Cout % 256) return hipErrorInvalidValue;
#ifdef ADAYOLO_MEASURE
    if (variant == 51) return pp::launch<1>(a, s);
    if (variant == 52) return pp::launch<2>(a, s);
    if (variant == 53) return pp::launch<3>(a, s);
    if (variant == 54) return pp::launch<4>(a, s);
    if (variant == 55) return pp::launch<5>(a, s);
    if (variant == 56) return pp::launch<6>(a, s);
    if (variant == 57) return pp::launch<7>(a, s);
    if (variant == 58) return pp::launch<8>(a, s);
    if (variant == 59) return pp::launch<9>(a, s);
#endif
    (void)variant;
    if (a.w2) {                                          // fused 1x1 second layer: the tile must hold all channels
        if (a.Cout != 256 || !a.bias2 || !a.out2) return hipErrorInvalidValue;
#ifdef ADAYOLO_MEASURE
        if (getenv("ADAYOLO_PP_STAMPS")) return pp::launch<7, true>(a, s);
#endif
        return pp::launch<0, true>(a, s);
    }
    return pp::launch<0>(a, s);
}

#ifdef ADAYOLO_MEASURE
// measurement helper (not part of the ABI): copies the stamps of the last variant-57 launch
extern "C" int adayolo_debug_stamps(unsigned long long* dst, int n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(pp::g_stamp), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif

#endif  // ADAYOLO_TILE_ONLY

}  // namespace adayolo
