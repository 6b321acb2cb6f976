// Conv + bias + SiLU (+ residual) — implicit GEMM, 256 px x 128 ch tile, FOUR waves, TWO workgroups per CU.
//
// What the ping-pong kernels (yolo_conv_pp.hip / yolo_conv_pp128.hip) cannot hide is everything outside the k-loop: a
// workgroup owns its CU (148 KB LDS, 8 waves x 246 registers), so its prologue (first k-tiles in flight) and its epilogue
// (SiLU + stores, VALU-issue-bound) leave the matrix pipes idle — 20-28 % of a workgroup's life at K = 1152. The
// persistent one-wave-per-SIMD form (512 registers, accumulators double-buffered so that a tile drains under the next
// tile's MFMAs; measured) does not hide them either: with a single wave per SIMD every stall of that
// wave — an LDS-DMA issue waits ~47 cycles for the texture-address path the four lock-stepped waves share, a SiLU chain
// waits for its transcendentals — is a stall of the SIMD's matrix pipe.
//
// Here the second client of a matrix pipe is a wave of ANOTHER workgroup: 256 threads, <= 256 registers, 72 KB LDS, so
// two workgroups are resident per CU and the hardware interleaves them. They are independent (different tiles, no common
// barrier) and drift apart by themselves: one's prologue / epilogue / DMA stalls run beside the other's MFMAs.
//
//   * wave grid 2 (px) x 2 (ch), wave tile 128 px x 64 ch (eight 32x32 accumulators), BK = 32: a k-tile is two steps
//     of 8 MFMAs; fragments double-buffered in registers (step kk's MFMAs run while step kk+1's six ds_read_b128 land);
//   * LDS ring 3 x 24 KB (rows of 64 B, 16-byte chunks XOR-swizzled by (row >> 2) & 3: conflict-free ds_read_b128),
//     staged by LDS-DMA, 6 instructions per wave and k-tile (16 rows each), issued between the MFMAs;
//   * ONE barrier per k-tile, at the top of its last step: behind it the k-tile's buffer is dead (re-staged with k-tile
//     t+3) and k-tile t+1 is visible (counted vmcnt(6) in front of the barrier, never 0);
//   * buffer addressing: 32-bit row offsets against a descriptor; rows beyond M, taps in the zero padding and k-tiles
//     beyond K use an out-of-range offset (the DMA writes zeros) — no zero page, no 64-bit address arithmetic;
//   * the bias enters through the matrix pipe (one extra MFMA per accumulator: fp32 bias split into three bf16 terms
//     against ones), the epilogue is the wave-private LDS transpose of yolo_conv_pp.hip overlaid on the finished ring.
//
// Restrictions (the launcher falls back otherwise): Cin % 32 == 0, Cout % 128 == 0, K >= 96, tensors addressable with
// 32-bit byte offsets, Ho*Wo > 1, Wo > 1.
#include "yolo_internal.h"
#include <type_traits>
#include <cstdlib>

namespace adayolo {
namespace pq {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
template <int N> using IC = std::integral_constant<int, N>;

constexpr int BN = 128, BK = 32;
constexpr int kRow = BK * 2;                  // 64 bytes per tile row
constexpr int kEpiPitch = 144;                // bytes per pixel row of a wave's private epilogue region (64 ch + pad)
// MI = 32-pixel fragments per wave: 4 -> 256 px tile (24 KB per k-tile, 72 KB ring), 2 -> 128 px tile (16 KB, 48 KB): the
// small maps (8x23x40: 29 tiles of 256 px) get twice the workgroups. (A 64 px tile — MI = 1, 82 registers, four workgroups
// per CU — was measured and dropped: slower on every layer, 16 vs 13 us on 1024 -> 512 @ 8x23x40, four MFMAs per barrier.)
template <int MI> struct Geo {
    static constexpr int BM = 64 * MI;
    static constexpr int kATile = BM * kRow;
    static constexpr int kBuf = (BM + BN) * kRow;
    static constexpr int kRing = 3 * kBuf;
    static constexpr int kEpi = 4 * 32 * MI * kEpiPitch;       // the epilogue overlays the finished ring
    static constexpr int kSmem = kRing > kEpi ? kRing : kEpi;
    static constexpr int NA = MI;                              // activation DMA pieces per wave and k-tile (16 rows each)
    static constexpr int NP = MI + 2;                          // ... + 2 weight pieces
};
constexpr unsigned kOOB = 0xFFFFFFFFu;
constexpr unsigned kRecords = 0xFFFFFF00u;
constexpr unsigned kDescFlags = 0x00020000u;

__device__ __forceinline__ void fence() { __builtin_amdgcn_sched_barrier(0); }
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_pk{lo, hi}, bf16x2));
}
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
#define PQ_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

// ABL (measurement builds): 0 the kernel, 1 no LDS-DMA in the k-loop, 2 no epilogue
template <int ABL, int MI>
__global__ __launch_bounds__(256, 2) void k_conv_pq(const ConvArgs a) {
    constexpr int BM = Geo<MI>::BM, kATile = Geo<MI>::kATile, kBuf = Geo<MI>::kBuf, NA = Geo<MI>::NA, NP = Geo<MI>::NP;
    constexpr int H1 = NP / 2;                           // pieces issued behind the barrier (step 1); the rest in step 0
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lid = xcd_remap(blockIdx.x, a.mtiles * a.ntiles);
    const int m0 = (lid / a.ntiles) * BM, n0 = (lid % a.ntiles) * BN;
    const int Kw = a.ks * a.ks * a.Cin;
    const int nK = Kw / BK;

    const unsigned guard = 2u * (unsigned)(a.W + 1) * (unsigned)a.in_cs;
    const __amdgpu_buffer_rsrc_t rsA =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)a.in - guard), 0, kRecords, kDescFlags);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, kRecords, kDescFlags);

    // ---- LDS-DMA: one instruction moves 16 tile rows (64 lanes x 16 B): lane -> row (lane >> 2), 16-byte position
    //      (lane & 3) whose source chunk is pos ^ ((row >> 2) & 3). Per k-tile a wave issues 4 activation pieces (rows
    //      [64 wave, 64 wave + 64)) and 2 weight pieces (rows [32 wave, 32 wave + 32)).
    const int slot = lane & 3, rsub = lane >> 2;
    // (arrays the lambdas capture keep a FIXED size — the 256-pixel form's — and the loops run to MI / NA: sized by the template
    // parameter they make hipcc drop the kernel's host-side launch stub without a diagnostic; the unused elements are dead)
    unsigned wvoff[2], vsel[4], msel[4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int r = wave * 32 + u * 16 + rsub;
        wvoff[u] = 2u * (unsigned)(n0 + r) * (unsigned)Kw + 16u * (unsigned)(slot ^ ((r >> 2) & 3));
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int r = wave * (16 * NA) + i * 16 + rsub;
        const int q = slot ^ ((r >> 2) & 3);
        const int m = m0 + r;
        const int mc = m < a.M ? m : a.M - 1;
        const int b = (int)(__umulhi((unsigned)mc, a.magic_hw) >> a.sh_hw);
        const int rem = mc - b * (a.Ho * a.Wo);
        const int ho = (int)(__umulhi((unsigned)rem, a.magic_w) >> a.sh_w);
        const int wo = rem - ho * a.Wo;
        const int hi0 = ho * a.stride - a.pad, wi0 = wo * a.stride - a.pad;
        unsigned vw = 0, mask = 0;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) vw |= (unsigned)(kw < a.ks && wi0 + kw >= 0 && wi0 + kw < a.W) << kw;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
            mask |= (kh < a.ks && hi0 + kh >= 0 && hi0 + kh < a.H) ? vw << (kh * a.ks) : 0u;
        mask = m < a.M ? mask : 0u;
        vsel[i] = guard + 2u * (unsigned)(((b * a.H + hi0) * a.W + wi0)) * (unsigned)a.in_cs + 16u * (unsigned)q;
        msel[i] = ~mask;
    }
    // DMA cursor: the k-tile staged next
    int c_c0 = 0, c_kh = 0, c_kw = 0, c_tap = 0, c_t = 0;
    unsigned c_soffA = 0, c_soffW = 0;
    unsigned c_dead = 0u;                                // all ones behind the last k-tile: every piece out of range
    auto cursor_advance = [&]() __attribute__((always_inline)) {
        c_c0 += BK;
        if (c_c0 >= a.Cin) {
            c_c0 = 0; ++c_tap;
            if (++c_kw == a.ks) { c_kw = 0; ++c_kh; }
        }
        if (++c_t >= nK) { c_dead = kOOB; c_tap = 0; }
        c_soffA = 2u * (unsigned)((c_kh * a.W + c_kw) * a.in_cs + c_c0);
        c_soffW = 2u * (unsigned)(c_tap * a.Cin + c_c0);
    };
    auto issue = [&](auto ptag, int dst) __attribute__((always_inline)) {
        constexpr int p = decltype(ptag)::value;
        if (ABL == 1) return;
        if (ABL == 3 && p < NA) return;                // measurement: weights only (what a patch-resident activation operand would leave)
        if constexpr (p < NA)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(smem + dst + (wave * (16 * NA) + p * 16) * kRow), 16,
                                                     vsel[p] | (unsigned)__builtin_amdgcn_sbfe((int)msel[p], (unsigned)c_tap, 1u) | c_dead,
                                                     c_soffA, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(smem + dst + kATile + (wave * 32 + (p - NA) * 16) * kRow), 16,
                                                     wvoff[p - NA] | c_dead, c_soffW, 0, 0);
    };
    // fragments (32x32x16): lane -> tile row (lane & 31), 16-byte k-chunk 2 kk + (lane >> 5), XOR key (row >> 2) & 3
    const int frow = lane & 31, fq = lane >> 5, key = (frow >> 2) & 3;
    int aoffk[2], woffk[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int ko = ((2 * kk + fq) ^ key) << 4;
        aoffk[kk] = (wm * (32 * MI) + frow) * kRow + ko;
        woffk[kk] = kATile + (wn * 64 + frow) * kRow + ko;
    }
    auto read_frags = [&](int buf, auto kktag, bf16x8 (&ra)[4], bf16x8 (&rw)[2]) __attribute__((always_inline)) {
        constexpr int kk = decltype(kktag)::value;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) rw[ni] = *reinterpret_cast<const bf16x8*>(smem + buf + woffk[kk] + ni * 32 * kRow);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) ra[mi] = *reinterpret_cast<const bf16x8*>(smem + buf + aoffk[kk] + mi * 32 * kRow);
    };

    // ---- prologue
    int b0 = 0, b1 = kBuf, b2 = 2 * kBuf;
    auto issue_first = [&](int dst) __attribute__((always_inline)) {          // pieces [0, H1)
        issue(IC<0>{}, dst); issue(IC<1>{}, dst);
        if constexpr (H1 > 2) issue(IC<2>{}, dst);
    };
    auto issue_rest = [&](int dst) __attribute__((always_inline)) {           // pieces [H1, NP)
        issue(IC<H1>{}, dst); issue(IC<H1 + 1>{}, dst);
        if constexpr (NP - H1 > 2) issue(IC<H1 + 2>{}, dst);
    };
    auto all6 = [&](int dst) __attribute__((always_inline)) { issue_first(dst); issue_rest(dst); };
    // bias of channel 32 ni + (lane & 31) of the wave's 64 — ordinary loads, issued AHEAD of the DMA stream so that the
    // compiler's wait for them leaves the stream in flight
    float bias_f[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) bias_f[ni] = a.bias[n0 + wn * 64 + ni * 32 + (lane & 31)];
    asm volatile("" : "+v"(bias_f[0]), "+v"(bias_f[1]));
    // k-tiles 0 and 1 complete, the first three pieces of k-tile 2 (what step 1 of the k-tile before would have staged)
    all6(b0); cursor_advance();
    all6(b1); cursor_advance();
    issue_first(b2);                                                     // cursor stays at k-tile 2

    f32x16 acc[2][4];                                    // [channel frag][pixel frag < MI]
    {
        // bias = hi + mid + lo, three bf16 terms (exact), in k-slots 0..2 of the channel operand of the lanes that hold
        // k-chunk 0; the pixel operand has ones there: acc = bias without 128 register writes
        const unsigned on = fq == 0 ? 0xFFFFFFFFu : 0u;
        const u32x4 ones = {0x3F803F80u & on, 0x00003F80u & on, 0u, 0u};
        f32x16 zero;
#pragma unroll
        for (int e = 0; e < 16; ++e) zero[e] = 0.0f;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const unsigned bu = __float_as_uint(bias_f[ni]);
            const unsigned hi = bu & 0xFFFF0000u;
            const float r1 = bias_f[ni] - __uint_as_float(hi);
            const unsigned mid = __float_as_uint(r1) & 0xFFFF0000u;
            const float r2 = r1 - __uint_as_float(mid);
            const u32x4 wb = {((hi >> 16) | mid) & on, (__float_as_uint(r2) >> 16) & on, 0u, 0u};
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wb), __builtin_bit_cast(bf16x8, ones), zero, 0, 0, 0);
        }
    }
    fence();
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(ABL == 1 ? 0 : ABL == 3 ? 2 : NP + H1) : "memory");   // k-tile 0 landed
    fence();
    bf16x8 fa[4], fw[2];
    read_frags(b0, IC<0>{}, fa, fw);

    auto mma = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[ni], fa[mi], acc[ni][mi], 0, 0, 0);
    };
    // ---- k-loop. Step 0: MFMAs of chunk 0 beside the reads of chunk 1 and pieces 3..5 of k-tile t+2 (its buffer was
    //      freed by the previous barrier). Step 1: barrier, then MFMAs of chunk 1 beside the reads of k-tile t+1's
    //      chunk 0 and pieces 0..2 of k-tile t+3 into the buffer just freed.
    for (int t = 0; t < nK; ++t) {
        bf16x8 na[4], nw[2];
        fence();
        read_frags(b0, IC<1>{}, na, nw);
        issue_rest(b2);
        mma();
#pragma unroll
        for (int n = 0; n < 2 * MI; ++n) {
            PQ_SGB(0x008, 1);
            if (n < (MI + 3) / 2) PQ_SGB(0x100, 2);
            else if (n < (MI + 3) / 2 + 3) { PQ_SGB(0x004, 1); PQ_SGB(0x020, 1); }
            PQ_SGB(0x002, 2);
        }
        fence();
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[i] = na[i];
        fw[0] = nw[0]; fw[1] = nw[1];
        cursor_advance();
        fence();
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(ABL == 1 ? 0 : ABL == 3 ? 2 : NP) : "memory");
        fence();
        read_frags(b1, IC<0>{}, na, nw);
        issue_first(b0);
        mma();
#pragma unroll
        for (int n = 0; n < 2 * MI; ++n) {
            PQ_SGB(0x008, 1);
            if (n < (MI + 3) / 2) PQ_SGB(0x100, 2);
            else if (n < (MI + 3) / 2 + 3) { PQ_SGB(0x004, 1); PQ_SGB(0x020, 1); }
            PQ_SGB(0x002, 2);
        }
        fence();
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[i] = na[i];
        fw[0] = nw[0]; fw[1] = nw[1];
        const int t0 = b0; b0 = b1; b1 = b2; b2 = t0;
    }
    asm volatile("" ::"v"(fa[0]), "v"(fw[0]));
    fence();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");    // the tail's zero-fill DMAs target the ring the epilogue overlays
    fence();
    if (ABL == 2) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) asm volatile("" ::"v"(acc[ni][mi]));
        return;
    }

    // ---- epilogue (see yolo_conv_pp.hip): D[row = channel][col = pixel]; each wave transposes its 128 px x 64 ch through a
    //      private LDS region (pitch 144 B) and writes 128-byte row segments; activation / residual are compile-time copies
    unsigned char* my = smem + wave * (32 * MI * kEpiPitch);
    // kKeep (training forward): the tile goes through LDS as the bf16 PRE-activation, is stored to a.pre, then activated from
    // that rounded value; kDs (backward): the result (+ residual) is dL/d(layer output), stored when a.out is set, and
    // a.gpre = it * silu'(a.pre) — the contracts of yolo_conv_pp128.hip's epilogue, bit for bit the separate SiLU launches
    auto epilogue = [&](auto silu_tag, auto res_tag, auto keep_tag, auto ds_tag) __attribute__((always_inline)) {
        constexpr bool kKeep = decltype(keep_tag)::value, kAct = decltype(silu_tag)::value, kDs = decltype(ds_tag)::value;
        constexpr bool kSilu = kAct && !kKeep, kRes = decltype(res_tag)::value;
        const int chunk = lane & 7, r0 = lane >> 3;
        const int mrow = m0 + wm * (32 * MI) + r0, n = n0 + wn * 64 + chunk * 8;
        unsigned short* const op = a.out + (long)mrow * a.out_cs + n;
        const unsigned short* const rp = kRes ? a.res + (long)mrow * a.res_cs + n : nullptr;
        const long ostep = 8L * a.out_cs, rstep = kRes ? 8L * a.res_cs : 0;
        unsigned char* const wr = my + (lane & 31) * kEpiPitch + 8 * (lane >> 5);
        const unsigned char* const rd = my + r0 * kEpiPitch + chunk * 16;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            u32x4 v[4], r[4];
            bool ok[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) ok[it] = mrow + 8 * (4 * mi + it) < a.M;
            if (kRes) {
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
                    f32x2_pk x0 = {acc[ni][mi][4 * qd], acc[ni][mi][4 * qd + 1]}, x1 = {acc[ni][mi][4 * qd + 2], acc[ni][mi][4 * qd + 3]};
                    if (kSilu) { x0 = silu_pk(x0); x1 = silu_pk(x1); }
                    *reinterpret_cast<u32x2*>(wr + mi * 32 * kEpiPitch + (ni * 32 + 8 * qd) * 2) =
                        u32x2{pack_bf16x2(x0.x, x0.y), pack_bf16x2(x1.x, x1.y)};
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < 4; ++it) v[it] = *reinterpret_cast<const u32x4*>(rd + (mi * 32 + it * 8) * kEpiPitch);
            if (kKeep) {
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
                        const f32x2_pk x = f32x2_pk{__uint_as_float(v[it][j] << 16), __uint_as_float(v[it][j] & 0xFFFF0000u)} +
                                           f32x2_pk{__uint_as_float(r[it][j] << 16), __uint_as_float(r[it][j] & 0xFFFF0000u)};
                        v[it][j] = pack_bf16x2(x.x, x.y);
                    }
            }
            if (kDs) {
                u32x4 p[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    p[it] = u32x4{0u, 0u, 0u, 0u};
                    if (ok[it]) p[it] = *reinterpret_cast<const u32x4*>(a.pre + (long)(mrow + 8 * (4 * mi + it)) * a.pre_cs + n);
                }
                if (a.out) {
#pragma unroll
                    for (int it = 0; it < 4; ++it)
                        if (ok[it]) __builtin_nontemporal_store(v[it], reinterpret_cast<u32x4*>(op + (4 * mi + it) * ostep));
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[it][j] = dsilu_bf16x2(v[it][j], p[it][j]);
                    if (ok[it])
                        __builtin_nontemporal_store(v[it], reinterpret_cast<u32x4*>(a.gpre + (long)(mrow + 8 * (4 * mi + it)) * a.gpre_cs + n));
                }
                continue;
            }
#pragma unroll
            for (int it = 0; it < 4; ++it)
                if (ok[it]) __builtin_nontemporal_store(v[it], reinterpret_cast<u32x4*>(op + (4 * mi + it) * ostep));
        }
    };
    const std::false_type no{};
    const std::true_type yes{};
    if (a.gpre) {
        if (a.res) epilogue(no, yes, no, yes); else epilogue(no, no, no, yes);
    } else if (a.pre) {
        if (a.act == ADAYOLO_ACT_SILU) { if (a.res) epilogue(yes, yes, yes, no); else epilogue(yes, no, yes, no); }
        else { if (a.res) epilogue(no, yes, yes, no); else epilogue(no, no, yes, no); }
    } else if (a.act == ADAYOLO_ACT_SILU) {
        if (a.res) epilogue(yes, yes, no, no); else epilogue(yes, no, no, no);
    } else {
        if (a.res) epilogue(no, yes, no, no); else epilogue(no, no, no, no);
    }
}

template <int ABL, int MI = 4>
static hipError_t launch(ConvArgs a, hipStream_t s) {
    constexpr int kSmem = Geo<MI>::kSmem;
    static_assert(2 * kSmem <= 160 * 1024, "two workgroups per CU");
    a.mtiles = (a.M + Geo<MI>::BM - 1) / Geo<MI>::BM;
    auto kern = k_conv_pq<ABL, MI>;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kSmem);
        if (e != hipSuccess) return e;
        configured = true;
    }
    int smem_bytes = kSmem;
#ifdef ADAYOLO_MEASURE
    static const int extra = getenv("ADAYOLO_PQ_EXTRA_SMEM") ? atoi(getenv("ADAYOLO_PQ_EXTRA_SMEM")) : 0;   // > 8 KB: one workgroup per CU
    smem_bytes += extra;
    if (extra) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
#endif
    hipLaunchKernelGGL(kern, dim3(a.mtiles * a.ntiles), dim3(256), smem_bytes, s, a);
    return hipGetLastError();
}

}  // namespace pq

// variant 80 = the kernel (256-pixel tile), 85 = its 128-pixel form; with -DADAYOLO_MEASURE 81 .. 83 = measurement builds.
// hipErrorInvalidValue -> not served.
hipError_t launch_conv_pq(ConvArgs a, hipStream_t s, int variant) {
    const long nK = (long)a.ks * a.ks * a.Cin / 32;
    if (a.Cin % 32 || a.Cout % 128 || nK < 3 || a.sh_hw < 0 || a.sh_w < 0 || a.d2s_c) return hipErrorInvalidValue;
    const unsigned long long lim = 0xFFFFFF00ull - 64;
    const unsigned long long in_b = 2ull * a.B * a.H * a.W * a.in_cs + 4ull * (a.W + 1) * a.in_cs + 2ull * a.Cin;
    if (in_b > lim || 2ull * a.Cout * a.ks * a.ks * a.Cin > lim) return hipErrorInvalidValue;
    a.ntiles = a.Cout / pq::BN;
#ifdef ADAYOLO_MEASURE
    if (variant == 81) return pq::launch<1>(a, s);
    if (variant == 82) return pq::launch<2>(a, s);
    if (variant == 83) return pq::launch<3>(a, s);
#endif
    if (variant == 85) return pq::launch<0, 2>(a, s);  // 128-pixel tile
    return pq::launch<0>(a, s);
}

}  // namespace adayolo
