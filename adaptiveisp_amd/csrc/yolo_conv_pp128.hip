// Conv + bias + SiLU (+ residual) — implicit GEMM, 256 px x 128 ch tile, two wave groups in ping-pong, 3-deep ring.
//
// Sibling of yolo_conv_pp.hip for the layers it cannot serve well: Cout = 128 (its tile is 256 channels wide) and
// small pixel counts where 256x256 tiles leave half the CUs idle (8x23x40 px: 116 tiles; here 232). Same ideas —
// waves 0-3 / 4-7 staggered by one barrier so that one wave of each SIMD issues MFMAs while the other reads
// fragments, LDS-DMA issued in the MFMA sections behind counted waits — with a different decomposition:
//
//   * wave grid 4 (px) x 2 (ch): wave tile 64 px x 64 ch (four 32x32 accumulators), the channel half is the
//     ping-pong group. Per k-tile (BK = 64) a wave reads 8 activation + 8 weight fragments for 16 MFMAs (a
//     2 x 4 grid with 128 x 32 wave tiles would need 20): LDS read + DMA write time stays below the MFMA time;
//   * a k-tile is TWO phases of 8 MFMAs (channel fragment 0, then 1). The activation fragments are read in P1's load
//     section, both weight fragments of the phase pair — W1 of this k-tile and W0 of the NEXT — in P2's: 8 reads per
//     load section; three weight register sets rotate;
//   * with two phases per k-tile a 2-buffer ring leaves one phase of DMA latency, so the ring is 3 k-tiles deep
//     (3 x 48 KB). Slots are units of what one load section reads: A (4 DMA instructions per wave), W0, W1 (1 each):
//         MFMA section of P1(t): issues W1(t+2), W0(t+3)  then s_waitcnt vmcnt(8)
//         MFMA section of P2(t): issues A(t+3)            then s_waitcnt vmcnt(10)
//     i.e. everything issued three phases ago is retired, and is first read two phases later (the distance the
//     stagger needs): issue -> read = 5 phases. A slot is re-staged at least one phase after its last read.
//
// Restrictions (the launcher falls back otherwise): Cin % 64 == 0, Cout % 128 == 0.
// Measured and dropped (round 2, same outputs bit for bit): ONE phase per k-tile — all 16 fragments in one load section, 16
// MFMAs per section, two barriers per k-tile instead of four, 190 registers — 71 vs 70 us on 512 -> 1024 @ 8x23x40; the same
// with the six DMA pieces issued in the LOAD section so that the MFMA section is MFMAs only: 88 us (a piece costs the issuing
// wave ~150 cycles there, four waves at once). The barriers are not what holds this kernel; the operand stream is
// (profiles/round2_conv_pp_ablation.txt).
#include "yolo_internal.h"
#include "yolo_chain.h"
#ifndef PP_PRIO_MODE
#define PP_PRIO_MODE 0      // 0: s_setprio 1 around every MFMA section (default); 1: no priority; 2: static priority for the second wave group (measurement builds)
#endif
#include <type_traits>

namespace adayolo {
namespace pp128 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
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

constexpr int BM = 256, BN = 128, BK = 64;
constexpr int kRow = BK * 2;                  // bytes per tile row
constexpr int kATile = BM * kRow;             // 32 KB
constexpr int kBuf = (BM + BN) * kRow;        // one k-tile: 48 KB
constexpr int kRing = 3 * kBuf;               // 144 KB
constexpr int kEpiPitch = 144;                // bytes per pixel row of a wave's private epilogue region (64 ch + pad)
constexpr int kSmem = kRing + BN * 4 + 16;    // + bias + the split-K ticket; the epilogue (8 x 64 x 144 B = 72 KB) overlays the finished ring

struct KPos {                                 // wave-uniform position of a k-tile
    int c0, kh, kw, tap;
    long aoff, woff;
};

// ABL: 0 real kernel, 5 no DMA instructions in the k-loop, 6 no epilogue, 7 activation DMA for one tap in nine
// (measurement builds; profiles/round2_conv_pp_ablation.txt)
//
// SPLIT (variants 100 + S, round 3): the k-tiles of one output tile are cut into S = a.ksplit equal ranges, one workgroup
// each — for the layers whose pixel count leaves most CUs without a tile (8 x 16 x 16 px at 1024 channels: 32 tiles of 144
// k-tiles; the training shapes of config 4). Every workgroup stores its fp32 accumulators to a.partial in its own lane
// order (16 B per lane, 1 KB per wave and instruction), takes a ticket of the tile, and the workgroup that draws the last
// one adds the S partial tiles IN SPLIT ORDER (its own included, read back: the sum does not depend on who arrives last)
// and runs the ordinary epilogue. The partials may cross XCDs, i.e. L2s: they are stored and loaded at device scope (sc1)
// and ordered by s_waitcnt vmcnt(0) + the ticket atomic — NOT by __threadfence(), whose release is a write-back of the
// whole L2 per workgroup (measured on 1024 -> 512 k3 @ 8x16x16, S = 8: k-loop 18.8 us, + partial stores 23.4, + fence and
// ticket 97; with sc1 accesses instead 27.8, + the last workgroup's S x 128 KB read-back and epilogue 43).
// Workgroup -> (tile, range) with the range varying fastest: the S workgroups of a tile are neighbours on one XCD
// (2-4 us better than tile-fastest).
// CHAIN (yolo_chain.h): the tile is a work item of the persistent chain kernel — `gid` is handed in, the outputs leave as
// written-through stores, the previous tile of the workgroup is published behind the prologue, wave 0 looks ahead.
template <int ABL, bool SPLIT, bool CHAIN>
__device__ __forceinline__ void conv_tile(const ConvArgs& a, const int gid, unsigned char* smem, ChainCtx& cx) {
    static_assert(!(SPLIT && CHAIN), "the chain runs whole tiles");
    float* bias_s = reinterpret_cast<float*>(smem + kRing);
    int* ticket_s = reinterpret_cast<int*>(smem + kRing + BN * 4);

    int tid_ = threadIdx.x;
    if (CHAIN) asm volatile("" : "+v"(tid_));            // (nothing derived from the thread index is hoisted out of the chain's loop)
    const int tid = tid_, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;             // wn is also the ping-pong group
    const int lid = SPLIT ? gid / a.ksplit : gid, kpart = SPLIT ? gid - lid * a.ksplit : 0;   // the ranges of a tile are neighbours
    ChainLook look;
    auto sched_stage = [&](int stage) {
        if (CHAIN && wave == 0) look.stage(stage, *cx.c, smem + kChainSchedOff, lane);
    };
    const int m0 = (lid / a.ntiles) * BM, n0 = (lid % a.ntiles) * BN;
    const unsigned long long zaddr = (unsigned long long)(const void*)g_zero16;

    // ---- per-row DMA state: one DMA instruction moves 8 tile rows. Activations: 32 instructions per k-tile, this
    //      wave issues the four of rows [32*wave, 32*wave + 32). Weights: unit W0 = rows [0,32) + [64,96) (channel
    //      fragment 0 of both groups), W1 = the other 64 rows; one instruction per wave and unit.
    const int slot = lane & 7, rsub = lane >> 3;
    unsigned long long arow[4], wrow[2];
    unsigned amask[4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int r = (wave >> 2) * 64 + u * 32 + (wave & 3) * 8 + rsub;
        const int q = slot ^ ((r >> 1) & 7);
        wrow[u] = (unsigned long long)(a.w + (long)(n0 + r) * (a.ks * a.ks * a.Cin) + 8 * q);
    }
    const int wlds0 = kATile + ((wave >> 2) * 64 + (wave & 3) * 8) * kRow;       // unit W0; W1 = + 32 rows
    const int alds0 = wave * 32 * kRow;                                          // + 8 rows per instruction
    auto decode_rows = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = wave * 32 + i * 8 + rsub;
            const int q = slot ^ ((r >> 1) & 7);
            const int m = m0 + r;
            unsigned mask = 0;
            long off = 0;
            if (m < a.M) {
                const int b = a.sh_hw < 0 ? m : (int)(__umulhi((unsigned)m, a.magic_hw) >> a.sh_hw);
                const int rem = m - b * (a.Ho * a.Wo);
                const int ho = a.sh_w < 0 ? rem : (int)(__umulhi((unsigned)rem, a.magic_w) >> a.sh_w);
                const int wo = rem - ho * a.Wo;
                const int hi0 = ho * a.stride - a.pad, wi0 = wo * a.stride - a.pad;
                unsigned vw = 0;                               // tap validity is separable: rows x columns
                // ks <= 3 (1 or 3 at the ABI; 2 for the stride-2 data gradient): three straight-line taps, no loop
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) vw |= (unsigned)(kw < a.ks && wi0 + kw >= 0 && wi0 + kw < a.W) << kw;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
                    mask |= (kh < a.ks && hi0 + kh >= 0 && hi0 + kh < a.H) ? vw << (kh * a.ks) : 0u;
                off = ((long)b * a.H * a.W + (long)hi0 * a.W + wi0) * a.in_cs + 8 * q;
            }
            amask[i] = mask;
            arow[i] = (unsigned long long)(a.in + off);
        }
    };
    const int cpt = a.Cin / BK;
    const int nK = SPLIT ? a.ks * a.ks * cpt / a.ksplit : a.ks * a.ks * cpt;     // k-tiles of THIS workgroup

    auto advance = [&](KPos& p) {
        p.c0 += BK;
        if (p.c0 >= a.Cin) {
            p.c0 = 0; ++p.tap; ++p.kw;
            if (p.kw == a.ks) { p.kw = 0; ++p.kh; }
        }
        p.aoff = 2 * (((long)p.kh * a.W + p.kw) * a.in_cs + p.c0);
        p.woff = 2 * ((long)p.tap * a.Cin + p.c0);
    };
    auto stage_a1 = [&](int i, unsigned char* buf, const KPos& p, bool live) {
        if (ABL == 5 && !live) return;
        const bool ok = live && ((amask[i] >> p.tap) & 1u);
        dma16(sel(ok, arow[i] + p.aoff, zaddr), buf + alds0 + i * 8 * kRow);
    };
    auto stage_w1 = [&](int u, unsigned char* buf, const KPos& p, bool live) {
        if (ABL == 5 && !live) return;
        dma16(sel(live, wrow[u] + p.woff, zaddr), buf + wlds0 + u * 32 * kRow);
    };
    auto stage_a = [&](unsigned char* buf, const KPos& p, bool live) {
#pragma unroll
        for (int i = 0; i < 4; ++i) stage_a1(i, buf, p, live);
    };

    // ---- prologue, in the steady-state issue order: W0(0); A(0); W1(0), W0(1); A(1); W1(1), W0(2); A(2)
    KPos q0{0, 0, 0, 0, 0, 0};
    if (SPLIT) {                                         // first k-tile of this workgroup's range
        const int t0 = kpart * nK;
        q0.tap = t0 / cpt; q0.c0 = (t0 - q0.tap * cpt) * BK;
        q0.kh = q0.tap / a.ks; q0.kw = q0.tap - q0.kh * a.ks;
        q0.aoff = 2 * (((long)q0.kh * a.W + q0.kw) * a.in_cs + q0.c0);
        q0.woff = 2 * ((long)q0.tap * a.Cin + q0.c0);
    }
    KPos q1 = q0; advance(q1);
    KPos q2 = q1; advance(q2);
    if (wave == 0 && lane < 32) dma16((unsigned long long)(a.bias + n0) + 16 * lane, bias_s);   // 512 B: half a wave
    stage_w1(0, smem, q0, true);
    decode_rows();
    stage_a(smem, q0, true);
    stage_w1(1, smem, q0, true);
    stage_w1(0, smem + kBuf, q1, 1 < nK);
    stage_a(smem + kBuf, q1, 1 < nK);
    stage_w1(1, smem + kBuf, q1, 1 < nK);
    stage_w1(0, smem + 2 * kBuf, q2, 2 < nK);
    stage_a(smem + 2 * kBuf, q2, 2 < nK);
    if (CHAIN) {
        wait_vm<0>();                                    // ... and the previous tile's written-through stores are complete
        barrier();
        chain_publish(cx, tid);
        sched_stage(0);
    } else {
        wait_vm<10>();                                   // W0(0), A(0), W1(0), W0(1) landed (this wave's share)
        barrier();
    }

    f32x16 acc[2][2];                                    // [channel frag][pixel frag]
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ni][mi][e] = 0.0f;

    // fragment addressing (32x32x16): lane -> tile row (lane & 31), 16-byte k-chunk 2*kk + (lane >> 5), XOR key
    const int frow = lane & 31, fq = lane >> 5, key = (frow >> 1) & 7;
    const int abase = (wm * 64 + frow) * kRow, wbase = kATile + (wn * 64 + frow) * kRow;
    int koff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koff[kk] = ((2 * kk + fq) ^ key) << 4;

    bf16x8 af[2][4], wx[4], wy[4], wz[4];
    auto read_a = [&](const unsigned char* buf) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                af[mi][kk] = *reinterpret_cast<const bf16x8*>(buf + abase + mi * 32 * kRow + koff[kk]);
    };
    auto read_w = [&](const unsigned char* buf, int ni, bf16x8 (&w)[4]) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            w[kk] = *reinterpret_cast<const bf16x8*>(buf + wbase + ni * 32 * kRow + koff[kk]);
    };
    // MFMA section: 8 MFMAs, the phase's DMA instructions issued behind the 1st, 3rd, 5th and 7th, then the counted wait
    // source addresses are computed in the load section in front (see yolo_conv_pp.hip): between two MFMAs only
    // s_mov m0 + the DMA instruction remain
    auto mma = [&](int ni, const bf16x8 (&w)[4], const unsigned long long (&g)[4], unsigned char* const (&d)[4], int npieces) {
#if PP_PRIO_MODE == 0
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[kk], af[mi][kk], acc[ni][mi], 0, 0, 0);
                const int n = 2 * kk + mi;
                if ((n & 1) == 0 && (n >> 1) < npieces) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (ABL != 5) dma16(g[n >> 1], d[n >> 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#if PP_PRIO_MODE == 0
        __builtin_amdgcn_s_setprio(0);
#endif
    };

    read_w(smem, 0, wx);                                 // W0 of k-tile 0
#if PP_PRIO_MODE == 2
    if (wn == 1) __builtin_amdgcn_s_setprio(1);
#endif
    if (wn == 1) barrier();                              // stagger group 1 by one barrier

    KPos p2 = q2, p3 = q2;                               // p2: k-tile t+2, p3: k-tile t+3 (advanced inside the loop)
    // one k-tile: `b0` its buffer, `b1` / `b2` the buffers of k-tiles t+1 / t+2 (t+3 lands in b0 again)
    auto ktile = [&](unsigned char* b0, unsigned char* b1, unsigned char* b2, int t, bf16x8 (&w0)[4], bf16x8 (&w1)[4],
                     bf16x8 (&wnx)[4]) {
        const bool live2 = ABL != 5 && t + 2 < nK, live3 = ABL != 5 && t + 3 < nK;
        advance(p3);                                      // -> k-tile t+3
        // P1: channel fragment 0; stages W1(t+2), W0(t+3)
        read_a(b0);
        {
            unsigned long long g[4] = {sel(live2, wrow[1] + p2.woff, zaddr), sel(live3, wrow[0] + p3.woff, zaddr), 0, 0};
            unsigned char* const d[4] = {b2 + wlds0 + 32 * kRow, b0 + wlds0, nullptr, nullptr};
            asm volatile("" : "+v"(g[0]), "+v"(g[1]));
            barrier();
            mma(0, w0, g, d, 2);
        }
        wait_vm<8>();
        barrier();
        // P2: channel fragment 1; the load section also fetches W0 of the NEXT k-tile; stages A(t+3)
        read_w(b0, 1, w1);
        read_w(b1, 0, wnx);
        {
            unsigned long long g[4];
            unsigned char* d[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                g[j] = sel(live3 && ((amask[j] >> p3.tap) & 1u), arow[j] + p3.aoff, zaddr);
                d[j] = b0 + alds0 + j * 8 * kRow;
            }
            asm volatile("" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]));
            barrier();
            unsigned char* const dc[4] = {d[0], d[1], d[2], d[3]};
            mma(1, w1, g, dc, (ABL == 7 && p3.tap != 0) ? 0 : 4);   // ABL 7 (measurement): activation DMA for one tap in nine
        }
        wait_vm<10>();
        barrier();
        p2 = p3;
    };
    unsigned char* B0 = smem;
    unsigned char* B1 = smem + kBuf;
    unsigned char* B2 = smem + 2 * kBuf;
    for (int t = 0; t < nK; t += 3) {
        ktile(B0, B1, B2, t, wx, wy, wz);
        if (t + 1 < nK) ktile(B1, B2, B0, t + 1, wz, wx, wy);
        if (t + 2 < nK) ktile(B2, B0, B1, t + 2, wy, wz, wx);
    }
    asm volatile("" ::"v"(wx[0]), "v"(wy[0]), "v"(wz[0]));
    if (wn == 0) barrier();                              // pairs with group 1's last barrier
    wait_vm<0>();                                        // the tail's zero-fill DMAs target the ring the epilogue overlays
    barrier();
    if (ABL == 6) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) asm volatile("" ::"v"(acc[ni][mi]));
        return;
    }

    if constexpr (SPLIT) {
        // partial tiles cross XCDs, i.e. L2s: stores and loads at DEVICE scope (sc1: written through / read past the
        // non-coherent lines), ordered by s_waitcnt + the ticket — a __threadfence() here is a whole-L2 write-back per
        // workgroup (measured: +70 us on a 25 us launch)
        constexpr int kSc1 = 16;
        const int S = a.ksplit;
        const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.partial + (long)lid * S * (BM * BN)), 0, 0xFFFFFF00u, 0x00020000u);
        const int mine = (kpart * 16 * 512 + tid) * 16;              // byte offset of this lane's first 16 B
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const u32x4 v = {__float_as_uint(acc[ni][mi][4 * g]), __float_as_uint(acc[ni][mi][4 * g + 1]),
                                     __float_as_uint(acc[ni][mi][4 * g + 2]), __float_as_uint(acc[ni][mi][4 * g + 3])};
                    __builtin_amdgcn_raw_buffer_store_b128(v, rsP, mine + ((ni * 2 + mi) * 4 + g) * (512 * 16), 0, kSc1);
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's partials are written through
        __syncthreads();
        if (tid == 0) *ticket_s = __hip_atomic_fetch_add(a.tickets + lid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*ticket_s != S - 1) return;
        if (tid == 0) __hip_atomic_store(a.tickets + lid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[ni][mi][e] = 0.0f;
        // S rounds of 16 loads per lane, the next round in flight while this one is added (a reducing CU pulls S x 128 KB)
        u32x4 va[16], vb[16];
        auto fetch = [&](u32x4 (&v)[16], int sp) {
            const int src = (sp * 16 * 512 + tid) * 16;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                v[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsP, src + j * (512 * 16), 0, kSc1));
        };
        auto add = [&](const u32x4 (&v)[16]) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[ni][mi][4 * g + c] += __uint_as_float(v[(ni * 2 + mi) * 4 + g][c]);
        };
        fetch(va, 0);
        for (int sp = 0; sp < S; sp += 2) {
            if (sp + 1 < S) fetch(vb, sp + 1);
            add(va);
            if (sp + 2 < S) fetch(va, sp + 2);
            if (sp + 1 < S) add(vb);
        }
    }

    // ---- epilogue: wave-private LDS transpose (see yolo_conv_pp.hip), 32 px x 64 ch at a time, 128-byte row segments
    unsigned char* my = smem + wave * (64 * kEpiPitch);
    __amdgpu_buffer_rsrc_t rs_out;                       // CHAIN: written-through (sc1) stores, 32-bit byte offsets
    if (CHAIN) rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, 0x7FFFFFFF, 0x00020000);
    // see yolo_conv_pp.hip: compile-time activation / residual copies, bias and pointers hoisted, batched reads and stores
    auto epilogue = [&](auto silu_tag, auto res_tag, auto keep_tag, auto ds_tag, auto d2s_tag) {
        // kKeep: the tile goes through LDS as the bf16 PRE-activation (kSilu off), is stored to a.pre, then activated
        // kDs (backward): the result (+ residual) is dL/d(layer output), stored when a.out is set; a.gpre = it * silu'(a.pre)
        // kD2s (stride-2 data gradient): every tensor the epilogue touches is addressed depth-to-space (epilogue_pos)
        constexpr bool kKeep = decltype(keep_tag)::value, kAct = decltype(silu_tag)::value, kDs = decltype(ds_tag)::value;
        constexpr bool kSilu = kAct && !kKeep, kRes = decltype(res_tag)::value, kD2s = decltype(d2s_tag)::value;
        static_assert(!(kD2s && (kKeep || kAct)), "depth-to-space addressing serves the backward forms only");
        typedef __attribute__((ext_vector_type(2))) float f32x2v;
        float4 bq[2][4];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
                bq[ni][qd] = *reinterpret_cast<const float4*>(bias_s + wn * 64 + ni * 32 + 8 * qd + 4 * (lane >> 5));
        const int chunk = lane & 7, r0 = lane >> 3;
        const int mrow = m0 + wm * 64 + r0, n = n0 + wn * 64 + chunk * 8;
        unsigned short* const op = a.out + (long)mrow * a.out_cs + n;
        const unsigned short* const rp = kRes ? a.res + (long)mrow * a.res_cs + n : nullptr;
        const long ostep = 8L * a.out_cs, rstep = kRes ? 8L * a.res_cs : 0;
        unsigned char* const wr = my + (lane & 31) * kEpiPitch + 8 * (lane >> 5);
        const unsigned char* const rd = my + r0 * kEpiPitch + chunk * 16;
        const int obyte = CHAIN ? (int)(((long)mrow * a.out_cs + n) * 2) : 0;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            sched_stage(mi + 1);                                 // CHAIN look-ahead: records, then arrival counters
            u32x4 v[4], r[4];
            bool ok[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) ok[it] = mrow + 8 * (4 * mi + it) < a.M;
            long px[4];                                          // kD2s: the pixel each row lands on, nn its channel there
            int nn = n;
            if (kD2s) {
#pragma unroll
                for (int it = 0; it < 4; ++it) epilogue_pos(a, ok[it] ? mrow + 8 * (4 * mi + it) : 0, n, px[it], nn);
            }
            if (kRes) {                                          // in flight while this group's SiLUs are computed
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    r[it] = u32x4{0u, 0u, 0u, 0u};
                    if (ok[it]) r[it] = *reinterpret_cast<const u32x4*>(kD2s ? a.res + px[it] * a.res_cs + nn : rp + (4 * mi + it) * rstep);
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
            if (kKeep) {                                         // training forward: store the pre-activation, activate its bf16 value
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    if (ok[it]) *reinterpret_cast<u32x4*>(a.pre + (long)(mrow + 8 * (4 * mi + it)) * a.pre_cs + n) = v[it];
                    if (kAct) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[it][j] = silu_bf16x2(v[it][j]);
                    }
                }
            }
            if (kRes) {
#pragma unroll
                for (int it = 0; it < 4; ++it)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x2v x = f32x2v{__uint_as_float(v[it][j] << 16), __uint_as_float(v[it][j] & 0xFFFF0000u)} +
                                         f32x2v{__uint_as_float(r[it][j] << 16), __uint_as_float(r[it][j] & 0xFFFF0000u)};
                        v[it][j] = pack_bf16x2(x.x, x.y);
                    }
            }
            if (kDs) {
                u32x4 p[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    p[it] = u32x4{0u, 0u, 0u, 0u};
                    if (ok[it])
                        p[it] = *reinterpret_cast<const u32x4*>(a.pre + (kD2s ? px[it] * a.pre_cs + nn
                                                                               : (long)(mrow + 8 * (4 * mi + it)) * a.pre_cs + n));
                }
                if (a.out) {
#pragma unroll
                    for (int it = 0; it < 4; ++it)
                        if (ok[it])
                            __builtin_nontemporal_store(v[it], reinterpret_cast<u32x4*>(kD2s ? a.out + px[it] * a.out_cs + nn
                                                                                             : op + (4 * mi + it) * ostep));
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[it][j] = dsilu_bf16x2(v[it][j], p[it][j]);
                    if (ok[it])
                        __builtin_nontemporal_store(v[it], reinterpret_cast<u32x4*>(
                            a.gpre + (kD2s ? px[it] * a.gpre_cs + nn : (long)(mrow + 8 * (4 * mi + it)) * a.gpre_cs + n)));
                }
                continue;
            }
#pragma unroll
            for (int it = 0; it < 4; ++it)
                if (ok[it]) {
                    if (CHAIN) __builtin_amdgcn_raw_buffer_store_b128(v[it], rs_out, obyte + (4 * mi + it) * (int)(2 * ostep), 0, 16);
                    else __builtin_nontemporal_store(v[it], reinterpret_cast<u32x4*>(kD2s ? a.out + px[it] * a.out_cs + nn
                                                                                          : op + (4 * mi + it) * ostep));
                }
        }
    };
    const std::false_type no{};
    const std::true_type yes{};
    if constexpr (CHAIN) {                               // the chain runs the plain forward forms only (yolo_api.hip checks)
        if (a.act == ADAYOLO_ACT_SILU) { if (a.res) epilogue(yes, yes, no, no, no); else epilogue(yes, no, no, no, no); }
        else { if (a.res) epilogue(no, yes, no, no, no); else epilogue(no, no, no, no, no); }
    } else if (a.d2s_c) {                                // stride-2 data gradient (act none, no kept pre-activation)
        if (a.gpre) { if (a.res) epilogue(no, yes, no, yes, yes); else epilogue(no, no, no, yes, yes); }
        else { if (a.res) epilogue(no, yes, no, no, yes); else epilogue(no, no, no, no, yes); }
    } else if (a.gpre) {
        if (a.res) epilogue(no, yes, no, yes, no); else epilogue(no, no, no, yes, no);
    } else if (a.pre) {
        if (a.act == ADAYOLO_ACT_SILU) { if (a.res) epilogue(yes, yes, yes, no, no); else epilogue(yes, no, yes, no, no); }
        else { if (a.res) epilogue(no, yes, yes, no, no); else epilogue(no, no, yes, no, no); }
    } else if (a.act == ADAYOLO_ACT_SILU) {
        if (a.res) epilogue(yes, yes, no, no, no); else epilogue(yes, no, no, no, no);
    } else {
        if (a.res) epilogue(no, yes, no, no, no); else epilogue(no, no, no, no, no);
    }
    if (CHAIN) {
        sched_stage(3);
        barrier();                                       // the tile's LDS is free; {next item, ready, ...} is in place
    }
}

template <int ABL, bool SPLIT>
__global__ __launch_bounds__(512) void k_conv_pp128(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ChainCtx none{nullptr, -1};
    const int ntile = a.mtiles * a.ntiles;
    conv_tile<ABL, SPLIT, false>(a, xcd_remap(blockIdx.x, SPLIT ? ntile * a.ksplit : ntile), smem, none);
}

#ifndef ADAYOLO_TILE_ONLY

template <int ABL, bool SPLIT = false>
static hipError_t launch(ConvArgs a, hipStream_t s) {
    static_assert(kSmem <= 160 * 1024, "LDS budget");
    auto kern = k_conv_pp128<ABL, SPLIT>;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kSmem);
        if (e != hipSuccess) return e;
        configured = true;
    }
    a.mtiles = (a.M + BM - 1) / BM;
    a.ntiles = a.Cout / BN;
    hipLaunchKernelGGL(kern, dim3(a.mtiles * a.ntiles * (SPLIT ? a.ksplit : 1)), dim3(512), kSmem, s, a);
    return hipGetLastError();
}

#endif  // ADAYOLO_TILE_ONLY

}  // namespace pp128

#ifndef ADAYOLO_TILE_ONLY

// Split-K form: S ranges of k-tiles per output tile. Served when the k-tiles divide evenly into ranges of at least
// kMinRange, and the tile count leaves CUs free (otherwise the plain kernel is the better one anyway). Returns the bytes of
// workspace the launch needs (tickets, then S fp32 partial tiles per output tile), 0 = not served.
size_t conv_pp128_splitk_bytes(const ConvArgs& a, int S) {
    constexpr int kMinRange = 4;
    if (a.Cin % 64 || a.Cout % 128 || S < 2 || S > 16) return 0;
    const int nK = a.ks * a.ks * (a.Cin / pp128::BK);
    if (nK % S || nK / S < kMinRange) return 0;
    const long tiles = (long)((a.M + pp128::BM - 1) / pp128::BM) * (a.Cout / pp128::BN);
    if (tiles * S > 512) return 0;
    return (size_t)((tiles * 4 + 1023) / 1024 * 1024) + (size_t)tiles * S * (pp128::BM * pp128::BN * 4);
}

hipError_t launch_conv_pp128_splitk(ConvArgs a, hipStream_t s, int S, void* workspace, size_t workspace_bytes) {
    const size_t need = conv_pp128_splitk_bytes(a, S);
    if (need == 0 || !workspace || workspace_bytes < need) return hipErrorInvalidValue;
    const long tiles = (long)((a.M + pp128::BM - 1) / pp128::BM) * (a.Cout / pp128::BN);
    a.ksplit = S;
    a.tickets = static_cast<int*>(workspace);
    a.partial = reinterpret_cast<float*>(static_cast<unsigned char*>(workspace) + (tiles * 4 + 1023) / 1024 * 1024);
    return pp128::launch<0, true>(a, s);
}

// variant 60 = the kernel; with -DADAYOLO_MEASURE 65 / 66 = measurement builds. hipErrorInvalidValue -> not served.
hipError_t launch_conv_pp128(ConvArgs a, hipStream_t s, int variant) {
    if (a.Cin % 64 || a.Cout % 128) return hipErrorInvalidValue;
#ifdef ADAYOLO_MEASURE
    if (variant == 65) return pp128::launch<5>(a, s);
    if (variant == 66) return pp128::launch<6>(a, s);
    if (variant == 67) return pp128::launch<7>(a, s);
#endif
    (void)variant;
    return pp128::launch<0>(a, s);
}

#endif  // ADAYOLO_TILE_ONLY

}  // namespace adayolo
