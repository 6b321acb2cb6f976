// EXPERIMENT — not part of libadayolo.so (measured in round 2, lost to the shipped kernels; kept as the record of the
// measurement, see DESIGN.md section 4 "What bounds the conv kernels"). To rebuild it: copy next to csrc/yolo_internal.h,
// add it to build.py and a `case 70` to yolo_api.hip; variants 71-73 (-DADAYOLO_MEASURE) are the ablation builds.
//   128->256 3x3 @8x92x160 (18 k-tiles): 2094 cycles per k-tile (1024 = MFMA-bound) = 84 us vs 69-73 us of yolo_conv_pp;
//   without the LDS-DMA 1507, without the in-loop drain 1580, without both 1178.
//
// Conv + bias + SiLU (+ residual) — implicit GEMM, persistent workgroups, ONE wave per SIMD with 512 registers,
// accumulators double-buffered so that a tile's epilogue runs in the matrix-pipe shadow of the NEXT tile's k-loop.
//
// The ping-pong kernels (yolo_conv_pp.hip / yolo_conv_pp128.hip) keep the matrix pipe fed inside the k-loop but leave it
// idle in every workgroup's prologue (first k-tiles in flight) and epilogue (SiLU + stores): 20-28 % of a workgroup's
// life at K = 1152. This kernel removes both:
//
//   * 256 threads = 4 waves, one per SIMD, up to 512 VGPR+AGPR each. Workgroup tile 256 px x 128 ch, wave grid
//     2 (px) x 2 (ch), wave tile 128 px x 64 ch = eight 32x32 accumulators (128 registers) — held TWICE: while tile j
//     accumulates into one set, the other set (tile j-1) is drained: SiLU, bf16 convert, LDS transpose, residual add,
//     128-byte row stores, cut into pieces that sit between the MFMAs of tile j's first k-tiles;
//   * workgroups are persistent (one per CU, static tile list per XCD): the LDS-DMA stream never stops at a tile boundary —
//     the k-tiles of tile j+1 are staged while tile j's last three k-tiles are computed, so there is no prologue after
//     the first tile and the ring is always three k-tiles (3 x 48 KB) deep;
//   * one s_barrier per k-tile (BK = 64, 32 MFMAs per wave). Fragments are double-buffered in registers: step kk's
//     MFMAs run while step kk+1's six ds_read_b128 are in flight. The barrier sits at the top of step 3, when a wave
//     holds the last fragments of the k-tile: behind it the k-tile's buffer is dead (re-staged with k-tile g+3) and
//     k-tile g+1 is visible (each wave's counted vmcnt before the barrier);
//   * every VMEM operation is issued unconditionally so that the counted waits are exact: rows beyond the image border /
//     beyond M, the taps of the zero padding and the stages behind the last tile use a buffer-descriptor offset that is
//     out of range (the load writes zeros to LDS, the store is dropped). Residual and bias loads are inline-asm buffer
//     loads (hipcc would otherwise drain the DMA queue with vmcnt(0) at their first use);
//   * the bias enters through the matrix pipe: a tile starts with one extra MFMA per accumulator whose channel operand
//     holds the fp32 bias split into three bf16 terms (exact) against a pixel operand of ones — two registers of bias
//     state per lane instead of 32, no add in the epilogue, no zeroing of 128 registers.
//
// LDS: 3 x 48 KB ring + 4 x 4 KB wave-private transpose regions = exactly 160 KB.
// Restrictions (the launcher falls back otherwise): Cin % 64 == 0, Cout % 128 == 0, SiLU, K >= 4 k-tiles, tensors
// addressable with 32-bit byte offsets.
#include "../../adaptiveisp_amd/csrc/yolo_internal.h"
#include <type_traits>

namespace adayolo {
namespace ps {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
template <int N> using IC = std::integral_constant<int, N>;

constexpr int BM = 256, BN = 128, BK = 64;
constexpr int kRow = BK * 2;                  // bytes per tile row
constexpr int kATile = BM * kRow;             // 32 KB
constexpr int kBuf = (BM + BN) * kRow;        // one k-tile: 48 KB
constexpr int kRing = 3 * kBuf;               // 144 KB
constexpr int kEpiWave = 32 * 128;            // 32 px x 64 ch bf16 per wave
constexpr int kSmem = kRing + 4 * kEpiWave;   // 160 KB
constexpr unsigned kOOB = 0xFFFFFFFFu;        // voffset no descriptor range contains
constexpr unsigned kRecords = 0xFFFFFF00u;
constexpr unsigned kDescFlags = 0x00020000u;
#ifndef PS_SCHED
#define PS_SCHED 1
#endif

__device__ __forceinline__ void fence() { __builtin_amdgcn_sched_barrier(0); }
// Issue order of one step (between two fences): every MFMA is followed by what fits in its 32-cycle shadow — first the
// six fragment reads of the next step (two per gap), then the three LDS-DMA pieces (one per gap), and up to four VALU
// instructions of drain work in every gap. Left alone hipcc issues all reads and DMA pieces ahead of the first MFMA
// (a ~50-cycle bubble per 256-cycle step) and the drain work in clumps of eight.
#define PS_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
__device__ __forceinline__ void step_schedule() {
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        PS_SGB(0x008, 1);                     // one MFMA
        if (n < 3) PS_SGB(0x100, 2);          // two ds_read
        else if (n < 6) { PS_SGB(0x004, 1); PS_SGB(0x020, 1); }   // s_mov m0 + one LDS-DMA piece
        PS_SGB(0x002, 4);                     // VALU
        if (n >= 6) PS_SGB(0x200, 1);         // a ds_write of the drain
    }
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_pk{lo, hi}, bf16x2));
}
// inline-asm buffer load the compiler does not count (see the header): issue, and a wait that names the destinations
__device__ __forceinline__ void asm_load16(u32x4& dst, unsigned voff, const u32x4& desc) {
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(desc) : "memory");
}

#ifdef ADAYOLO_MEASURE
__device__ unsigned long long g_dbg[8];      // block 0: shader cycles and 100 MHz ticks of the whole workgroup, tiles, k-tiles
#endif
// ABL (measurement builds): 0 the kernel; 1 no drain work inside the k-loop (outputs of all but the last tile of a
// workgroup are not written); 2 no LDS-DMA in the k-loop
template <bool RES, int SPREAD, int ABL>
__global__ __launch_bounds__(256) void k_conv_ps(const ConvArgs a, const int total) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int slot = lane & 7, rsub = lane >> 3;
#ifdef ADAYOLO_MEASURE
    const unsigned long long dbg_c0 = __builtin_readcyclecounter(), dbg_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- static tile list: XCD x owns the contiguous tile range [tbase, tbase + cnt); its workgroups stride through it,
    //      so the <= 32 tiles an XCD works on at any time are neighbours (channel tiles of the same pixels, then the next
    //      pixels): the activation rows they share stay in that XCD's L2
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int tq = total >> 3, tr = total & 7;
    const int cnt = tq + (xcd < tr ? 1 : 0);
    const int tbase = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int ntl = (cnt - idx + 31) >> 5;                 // tiles of this workgroup (>= 1: the grid is min(total, 256))
    const int Kw = a.ks * a.ks * a.Cin;                    // weight row length
    const int nK = Kw / BK;

    // ---- buffer descriptors. Activations: base moved back by one image row + one pixel so that every row offset is
    //      >= 0 (the rows of the zero padding are masked per tap, never read)
    const unsigned guard = 2u * (unsigned)(a.W + 1) * (unsigned)a.in_cs;
    const __amdgpu_buffer_rsrc_t rsA =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)a.in - guard), 0, kRecords, kDescFlags);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, kRecords, kDescFlags);
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, kRecords, kDescFlags);
    auto desc_of = [](const void* p) {
        const unsigned long long u = (unsigned long long)p;
        return u32x4{(unsigned)u, (unsigned)(u >> 32) & 0xFFFFu, kRecords, kDescFlags};
    };
    const u32x4 dB = desc_of(a.bias), dR = desc_of(RES ? (const void*)a.res : (const void*)a.bias);

    // ---- per-lane constants
    // LDS-DMA: one instruction moves 8 tile rows (64 lanes x 16 B): lane -> row rsub, 16-byte position `slot` whose
    // source chunk is slot ^ ((row >> 1) & 7) (the swizzle the fragment reads undo). Per k-tile a wave issues 8
    // activation pieces (rows [64 wave, 64 wave + 64)) and 4 weight pieces (rows [32 wave, 32 wave + 32)).
    unsigned wvoff[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = wave * 32 + u * 8 + rsub;
        wvoff[u] = 2u * (unsigned)r * (unsigned)Kw + 16u * (unsigned)(slot ^ ((r >> 1) & 7));
    }
    // fragments (32x32x16): lane -> tile row (lane & 31), 16-byte k-chunk 2 kk + (lane >> 5), XOR key (row >> 1) & 7
    const int frow = lane & 31, fq = lane >> 5, key = (frow >> 1) & 7;
    int aoffk[4], woffk[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int ko = ((2 * kk + fq) ^ key) << 4;
        aoffk[kk] = (wm * 128 + frow) * kRow + ko;
        woffk[kk] = kATile + (wn * 64 + frow) * kRow + ko;
    }
    // epilogue: D[row = channel][col = pixel]; lane holds pixel (lane & 31) and channels 8 qd + 4 (lane >> 5) + (0..3) of
    // a 32x32 fragment. Transpose through the wave's 4 KB region (32 px x 128 B, 16-byte chunks XOR-swizzled by pixel)
    unsigned char* const my = smem + kRing + wave * kEpiWave;
    const int pk = lane & 7;
    unsigned char* const wr0 = my + (lane & 31) * 128 + (lane >> 5) * 8;
    const int chunk = lane & 7, r0 = lane >> 3;
    const unsigned char* const rd0 = my + r0 * 128 + ((chunk ^ (r0 & 7)) << 4);
    const int ml = wm * 128 + r0, nl = wn * 64 + chunk * 8;          // this lane's first output row / channel in a tile
    const unsigned bvoff = 4u * (unsigned)(wn * 64 + (lane & 31));   // bias: channel 32 ni + (lane & 31) of the wave's 64

    // ---- state
    f32x16 accA[2][4], accB[2][4];                      // [channel frag][pixel frag], two tiles
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) { accA[ni][mi][e] = 0.0f; accB[ni][mi][e] = 0.0f; }
    bf16x8 fa[4], fw[2];                                // fragments of the next step to execute
    unsigned biasv[2] = {0u, 0u};                       // bias of the next tile to start: channel 32 ni + (lane & 31)
    unsigned vsel[8], msel[8];                 // activation rows of the tile the DMA cursor is in: offset, ~tap mask
    unsigned rsNv[8], rsNm[8];                          // ... of the tile after it (decoded ahead)
    // DMA cursor: the k-tile staged next
    int c_c0 = 0, c_kh = 0, c_kw = 0, c_tap = 0, c_t = 0;
    unsigned c_soffA = 0, c_soffW = 0, c_wbase = 0;
    unsigned n_wbase = 0;
    bool n_dead = false;
    int b0 = 0, b1 = kBuf, b2 = 2 * kBuf;               // ring: buffer of the computed k-tile, of the next, of the one after
    int m0d = a.M, n0d = 0;                             // tile in the draining accumulator set (none yet: every row masked)

    auto tile_mn = [&](int j, int& m0, int& n0) __attribute__((always_inline)) {
        const int L = tbase + idx + 32 * j;
        const int mt = L / a.ntiles;
        m0 = mt * BM;
        n0 = (L - mt * a.ntiles) * BN;
    };
    // activation row `i` of a tile at m0 -> buffer offset (incl. guard and chunk swizzle) and inverted 9-bit tap mask
    auto decode_row = [&](int m0, int i, unsigned& voff, unsigned& inv) __attribute__((always_inline)) {
        const int r = wave * 64 + i * 8 + rsub;
        const int q = slot ^ ((r >> 1) & 7);
        const int m = m0 + r;
        const int mc = m < a.M ? m : a.M - 1;             // branch-free: a divergent branch would split the MFMA schedule
        const int b = (int)(__umulhi((unsigned)mc, a.magic_hw) >> a.sh_hw);      // Ho*Wo > 1 and Wo > 1 (launcher)
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
        // ((b H + hi0) W + wi0) can be -W-1 at least: the guard makes the sum non-negative (mod 2^32 arithmetic)
        const unsigned off = guard + 2u * (unsigned)(((b * a.H + hi0) * a.W + wi0)) * (unsigned)a.in_cs + 16u * (unsigned)q;
        voff = off;
        inv = ~mask;
    };
    auto cursor_offsets = [&]() __attribute__((always_inline)) {
        c_soffA = 2u * (unsigned)((c_kh * a.W + c_kw) * a.in_cs + c_c0);
        c_soffW = c_wbase + 2u * (unsigned)(c_tap * a.Cin + c_c0);
    };
    auto cursor_advance = [&]() __attribute__((always_inline)) {
        c_c0 += BK;
        if (c_c0 >= a.Cin) {
            c_c0 = 0; ++c_tap;
            if (++c_kw == a.ks) { c_kw = 0; ++c_kh; }
        }
        if (++c_t == nK) {                               // into the next tile of this workgroup (once per tile)
            c_t = 0; c_c0 = c_kh = c_kw = c_tap = 0;
            c_wbase = n_wbase;
#pragma unroll
            for (int i = 0; i < 8; ++i) { vsel[i] = rsNv[i]; msel[i] = rsNm[i]; }
            if (n_dead) {
#pragma unroll
                for (int u = 0; u < 4; ++u) wvoff[u] = kOOB;
            }
        }
        cursor_offsets();
    };
    // piece p (0..7 activations, 8..11 weights) of the cursor's k-tile into ring buffer `dst`
    auto issue = [&](auto ptag, int dst) __attribute__((always_inline)) {
        constexpr int p = decltype(ptag)::value;
        if (ABL == 2 || ABL == 3) return;
        if constexpr (p < 8)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(smem + dst + (wave * 64 + p * 8) * kRow), 16,
                                                     vsel[p] | (unsigned)__builtin_amdgcn_sbfe((int)msel[p], (unsigned)c_tap, 1u),
                                                     c_soffA, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(smem + dst + kATile + (wave * 32 + (p - 8) * 8) * kRow),
                                                     16, wvoff[p - 8], c_soffW, 0, 0);
    };
    auto read_frags = [&](int buf, auto kktag, bf16x8 (&ra)[4], bf16x8 (&rw)[2]) __attribute__((always_inline)) {
        constexpr int kk = decltype(kktag)::value;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) rw[ni] = *reinterpret_cast<const bf16x8*>(smem + buf + woffk[kk] + ni * 32 * kRow);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) ra[mi] = *reinterpret_cast<const bf16x8*>(smem + buf + aoffk[kk] + mi * 32 * kRow);
    };
    auto load_bias = [&](int n0) __attribute__((always_inline)) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
            asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, 0 offen" : "=v"(biasv[ni]) : "v"(bvoff + 4u * (unsigned)(n0 + ni * 32)), "s"(dB) : "memory");
    };

    // ---- drain pieces (accumulator set D = the tile before the one being computed)
    u32x4 dv[4], dr[4];                                  // transposed rows / residual rows of the chunk in flight
    unsigned dob = 0, drb = 0;                           // output / residual offset of this lane's first row of the drained tile
    auto drain_setup = [&]() __attribute__((always_inline)) {
        dob = 2u * ((unsigned)(m0d + ml) * (unsigned)a.out_cs + (unsigned)(n0d + nl));
        if (RES) drb = 2u * ((unsigned)(m0d + ml) * (unsigned)a.res_cs + (unsigned)(n0d + nl));
    };
    auto e_resload = [&](auto ctag) __attribute__((always_inline)) {                    // 4 residual rows of chunk c
        constexpr int c = decltype(ctag)::value;
        if constexpr (RES) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = 32 * c + 8 * it;
                const unsigned v = (m0d + ml + row < a.M) ? drb + 2u * (unsigned)(row * a.res_cs) : kOOB;
                asm_load16(dr[it], v, dR);
            }
        }
    };
    auto e_act = [&](f32x16 (&D)[2][4], auto ctag, auto nitag, auto qdtag, auto hztag) __attribute__((always_inline)) {   // SiLU + bf16 of 4 channels -> LDS
        constexpr int c = decltype(ctag)::value, ni = decltype(nitag)::value, qd = decltype(qdtag)::value;
        // The drained set stays in the accumulator file until the piece that needs it: read here, explicitly (left to
        // itself hipcc copies all 128 registers to VGPRs at the top of the tile: 128 live VGPRs and a 128-instruction
        // bubble). `first`: the set's last MFMA may be only a few instructions back -> wait states inside the string.
        float d0, d1, d2, d3;
        if constexpr (decltype(hztag)::value != 0)
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\tv_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"
                         : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3)
                         : "a"(D[ni][c][4 * qd]), "a"(D[ni][c][4 * qd + 1]), "a"(D[ni][c][4 * qd + 2]), "a"(D[ni][c][4 * qd + 3]));
        else
            asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"
                         : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3)
                         : "a"(D[ni][c][4 * qd]), "a"(D[ni][c][4 * qd + 1]), "a"(D[ni][c][4 * qd + 2]), "a"(D[ni][c][4 * qd + 3]));
        const f32x2_pk x0 = silu_pk(f32x2_pk{d0, d1});
        const f32x2_pk x1 = silu_pk(f32x2_pk{d2, d3});
        *reinterpret_cast<u32x2*>(wr0 + (((ni * 4 + qd) ^ pk) << 4)) = u32x2{pack_bf16x2(x0.x, x0.y), pack_bf16x2(x1.x, x1.y)};
    };
    auto e_read = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < 4; ++it) dv[it] = *reinterpret_cast<const u32x4*>(rd0 + it * 1024);
    };
    auto e_store = [&](auto ctag, auto waittag) __attribute__((always_inline)) {         // (+ residual) and the four 128-byte row stores of chunk c
        constexpr int c = decltype(ctag)::value, NW = decltype(waittag)::value;
        if constexpr (RES) {
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(dr[0]), "+v"(dr[1]), "+v"(dr[2]), "+v"(dr[3]) : "n"(NW) : "memory");
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x2_pk x = f32x2_pk{__uint_as_float(dv[it][j] << 16), __uint_as_float(dv[it][j] & 0xFFFF0000u)} +
                                       f32x2_pk{__uint_as_float(dr[it][j] << 16), __uint_as_float(dr[it][j] & 0xFFFF0000u)};
                    dv[it][j] = pack_bf16x2(x.x, x.y);
                }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = 32 * c + 8 * it;
            const unsigned v = (m0d + ml + row < a.M) ? dob + 2u * (unsigned)(row * a.out_cs) : kOOB;
            __builtin_amdgcn_raw_buffer_store_b128(dv[it], rsO, v, 0, 2);
        }
    };

    // ---- one compute step: fragments of the following step, three DMA pieces, filler work, eight MFMAs
    auto mma = [&](f32x16 (&C)[2][4], auto firsttag) __attribute__((always_inline)) {
        constexpr bool first = decltype(firsttag)::value != 0;
        if constexpr (first) {
            // bias = hi + mid + lo, three bf16 terms (exact: 8 + 8 + 8 mantissa bits), in k-slots 0..2 of the channel operand
            // of the lanes that hold k-chunk 0; the pixel operand has ones there
            const unsigned on = fq == 0 ? 0xFFFFFFFFu : 0u;
            const u32x4 ones = {0x3F803F80u & on, 0x00003F80u & on, 0u, 0u};
            f32x16 zero;
#pragma unroll
            for (int e = 0; e < 16; ++e) zero[e] = 0.0f;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const float bf = __uint_as_float(biasv[ni]);
                const unsigned hi = biasv[ni] & 0xFFFF0000u;
                const float r1 = bf - __uint_as_float(hi);
                const unsigned mid = __float_as_uint(r1) & 0xFFFF0000u;
                const float r2 = r1 - __uint_as_float(mid);
                const u32x4 wb = {((hi >> 16) | mid) & on, (__float_as_uint(r2) >> 16) & on, 0u, 0u};
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    C[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wb), __builtin_bit_cast(bf16x8, ones), zero, 0, 0, 0);
            }
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                C[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[ni], fa[mi], C[ni][mi], 0, 0, 0);
    };
    // filler work of step `st` of a k-tile that carries drain unit U (-1: none) — see the header for the layout
    int m0n = a.M;                                       // next tile of this workgroup (decoded during unit 0..)
    auto filler = [&](f32x16 (&D)[2][4], auto utag, auto sttag, auto firsttag, int n0next) __attribute__((always_inline)) {
        constexpr int U = decltype(utag)::value, st = decltype(sttag)::value;
        constexpr bool first = decltype(firsttag)::value != 0;
        if constexpr (first && st == 1) load_bias(n0next);            // biasv is dead behind step 0's MFMAs
        if constexpr (U >= 0) {
            // rows of the next tile: SPREAD 8 -> two rows per k-tile in units 0..3; SPREAD 4 -> all eight in unit 0
            if constexpr (SPREAD == 8) {
                if constexpr (U < 4 && (st == 1 || st == 2)) {
                    constexpr int i = 2 * U + (st - 1);
                    decode_row(m0n, i, rsNv[i], rsNm[i]);
                }
            } else {
                if constexpr (U == 0) {
                    decode_row(m0n, 2 * st, rsNv[2 * st], rsNm[2 * st]);
                    decode_row(m0n, 2 * st + 1, rsNv[2 * st + 1], rsNm[2 * st + 1]);
                }
            }
            if constexpr (ABL != 1 && ABL != 3) {
                if constexpr (SPREAD == 8) {
                    constexpr int c = U >> 1, h = U & 1;
                    if constexpr (h == 0) {
                        if constexpr (st == 0) e_resload(IC<c>{});
                        e_act(D, IC<c>{}, IC<0>{}, IC<st>{}, IC<(c == 0 && st == 0) ? 1 : 0>{});
                    } else {
                        if constexpr (st == 0) { e_act(D, IC<c>{}, IC<1>{}, IC<0>{}, IC<0>{}); e_act(D, IC<c>{}, IC<1>{}, IC<1>{}, IC<0>{}); }
                        if constexpr (st == 1) { e_act(D, IC<c>{}, IC<1>{}, IC<2>{}, IC<0>{}); e_act(D, IC<c>{}, IC<1>{}, IC<3>{}, IC<0>{}); }
                        if constexpr (st == 2) e_read();
                        if constexpr (st == 3) e_store(IC<c>{}, IC<12>{});
                    }
                } else {
                    constexpr int c = U;
                    if constexpr (st == 0) {
                        e_resload(IC<c>{});
                        e_act(D, IC<c>{}, IC<0>{}, IC<0>{}, IC<c == 0 ? 1 : 0>{}); e_act(D, IC<c>{}, IC<0>{}, IC<1>{}, IC<0>{});
                        e_act(D, IC<c>{}, IC<0>{}, IC<2>{}, IC<0>{}); e_act(D, IC<c>{}, IC<0>{}, IC<3>{}, IC<0>{});
                    }
                    if constexpr (st == 1) {
                        e_act(D, IC<c>{}, IC<1>{}, IC<0>{}, IC<0>{}); e_act(D, IC<c>{}, IC<1>{}, IC<1>{}, IC<0>{});
                        e_act(D, IC<c>{}, IC<1>{}, IC<2>{}, IC<0>{}); e_act(D, IC<c>{}, IC<1>{}, IC<3>{}, IC<0>{});
                    }
                    if constexpr (st == 2) e_read();
                    if constexpr (st == 3) e_store(IC<c>{}, IC<6>{});
                }
            }
        }
    };
    // VMEM operations besides the DMA pieces that are certainly issued between the last piece of k-tile g+1 and the
    // barrier of k-tile g (steps 0..2 of this k-tile; under-counting is safe, over-counting is a race)
    auto ktile = [&](f32x16 (&C)[2][4], f32x16 (&D)[2][4], auto utag, auto firsttag, int n0next) __attribute__((always_inline)) {
        constexpr int U = decltype(utag)::value;
        constexpr bool first = decltype(firsttag)::value != 0;
        constexpr bool loads0 = RES && ABL != 1 && ABL != 3 && U >= 0 && (SPREAD == 4 || (U & 1) == 0);
        constexpr int NBAR = 12 + (loads0 ? 4 : 0) + (first ? 2 : 0);
        bf16x8 na[4], nw[2];
        // step 0
        fence();
        if constexpr (first)
            asm volatile("s_waitcnt vmcnt(12)" : "+v"(biasv[0]), "+v"(biasv[1]) :: "memory");
        read_frags(b0, IC<1>{}, na, nw);
        issue(IC<3>{}, b2); issue(IC<4>{}, b2); issue(IC<5>{}, b2);
        mma(C, firsttag);
        filler(D, utag, IC<0>{}, firsttag, n0next);
        if (PS_SCHED) step_schedule();
        fence();
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = na[i];
        fw[0] = nw[0]; fw[1] = nw[1];
        // step 1
        read_frags(b0, IC<2>{}, na, nw);
        issue(IC<6>{}, b2); issue(IC<7>{}, b2); issue(IC<8>{}, b2);
        mma(C, IC<0>{});
        filler(D, utag, IC<1>{}, firsttag, n0next);
        if (PS_SCHED) step_schedule();
        fence();
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = na[i];
        fw[0] = nw[0]; fw[1] = nw[1];
        // step 2
        read_frags(b0, IC<3>{}, na, nw);
        issue(IC<9>{}, b2); issue(IC<10>{}, b2); issue(IC<11>{}, b2);
        mma(C, IC<0>{});
        filler(D, utag, IC<2>{}, firsttag, n0next);
        if (PS_SCHED) step_schedule();
        fence();
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = na[i];
        fw[0] = nw[0]; fw[1] = nw[1];
        cursor_advance();                                 // -> k-tile g+3
        // step 3: this wave holds the last fragments of k-tile g; k-tile g+1 must be visible, buffer b0 is dead behind
        // the barrier
        fence();
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((ABL == 2 || ABL == 3) ? 0 : NBAR) : "memory");
        fence();
        read_frags(b1, IC<0>{}, na, nw);
        issue(IC<0>{}, b0); issue(IC<1>{}, b0); issue(IC<2>{}, b0);
        mma(C, IC<0>{});
        filler(D, utag, IC<3>{}, firsttag, n0next);
        if (PS_SCHED) step_schedule();
        fence();
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = na[i];
        fw[0] = nw[0]; fw[1] = nw[1];
        const int t0 = b0; b0 = b1; b1 = b2; b2 = t0;
    };
    auto tile_body = [&](f32x16 (&C)[2][4], f32x16 (&D)[2][4], int j) __attribute__((always_inline)) {
        int m0c, n0c, n0n = 0;
        tile_mn(j, m0c, n0c);
        if (j + 1 < ntl) { tile_mn(j + 1, m0n, n0n); n_dead = false; }
        else { m0n = a.M; n_dead = true; }
        n_wbase = 2u * (unsigned)n0n * (unsigned)Kw;
        drain_setup();
        // the first SPREAD k-tiles carry the drain of the previous tile (+ the row decode of the next)
        ktile(C, D, IC<0>{}, IC<1>{}, n0n);
        ktile(C, D, IC<1>{}, IC<0>{}, n0n);
        ktile(C, D, IC<2>{}, IC<0>{}, n0n);
        ktile(C, D, IC<3>{}, IC<0>{}, n0n);
        if constexpr (SPREAD == 8) {
            ktile(C, D, IC<4>{}, IC<0>{}, n0n);
            ktile(C, D, IC<5>{}, IC<0>{}, n0n);
            ktile(C, D, IC<6>{}, IC<0>{}, n0n);
            ktile(C, D, IC<7>{}, IC<0>{}, n0n);
        }
        for (int t = SPREAD; t < nK; ++t) ktile(C, D, IC<-1>{}, IC<0>{}, n0n);
        m0d = m0c; n0d = n0c;
    };
    auto final_drain = [&](f32x16 (&D)[2][4]) __attribute__((always_inline)) {
        drain_setup();
        auto chunk_out = [&](auto ctag) __attribute__((always_inline)) {
            e_resload(ctag);
            e_act(D, ctag, IC<0>{}, IC<0>{}, IC<1>{}); e_act(D, ctag, IC<0>{}, IC<1>{}, IC<0>{}); e_act(D, ctag, IC<0>{}, IC<2>{}, IC<0>{}); e_act(D, ctag, IC<0>{}, IC<3>{}, IC<0>{});
            e_act(D, ctag, IC<1>{}, IC<0>{}, IC<0>{}); e_act(D, ctag, IC<1>{}, IC<1>{}, IC<0>{}); e_act(D, ctag, IC<1>{}, IC<2>{}, IC<0>{}); e_act(D, ctag, IC<1>{}, IC<3>{}, IC<0>{});
            e_read();
            e_store(ctag, IC<0>{});
        };
        chunk_out(IC<0>{}); chunk_out(IC<1>{}); chunk_out(IC<2>{}); chunk_out(IC<3>{});
    };

    // ---- prologue of the workgroup: rows of tile 0, its bias, k-tiles 0 and 1 and the first three pieces of k-tile 2
    {
        int m0c, n0c;
        tile_mn(0, m0c, n0c);
#pragma unroll
        for (int i = 0; i < 8; ++i) decode_row(m0c, i, vsel[i], msel[i]);
        c_wbase = 2u * (unsigned)n0c * (unsigned)Kw;
        cursor_offsets();
        load_bias(n0c);
        auto all12 = [&](int dst) __attribute__((always_inline)) {
            issue(IC<0>{}, dst); issue(IC<1>{}, dst); issue(IC<2>{}, dst); issue(IC<3>{}, dst); issue(IC<4>{}, dst); issue(IC<5>{}, dst);
            issue(IC<6>{}, dst); issue(IC<7>{}, dst); issue(IC<8>{}, dst); issue(IC<9>{}, dst); issue(IC<10>{}, dst); issue(IC<11>{}, dst);
        };
        all12(b0);
        cursor_advance();
        all12(b1);
        cursor_advance();
        issue(IC<0>{}, b2); issue(IC<1>{}, b2); issue(IC<2>{}, b2);
        fence();
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((ABL == 2 || ABL == 3) ? 0 : 15) : "memory");   // bias + k-tile 0 landed (bias is older)
        fence();
        read_frags(b0, IC<0>{}, fa, fw);
    }
    for (int j = 0;;) {
        tile_body(accA, accB, j);
        if (++j == ntl) { final_drain(accA); break; }
        tile_body(accB, accA, j);
        if (++j == ntl) { final_drain(accB); break; }
    }
#ifdef ADAYOLO_MEASURE
    if (blockIdx.x == 0 && tid == 0) {
        g_dbg[0] = __builtin_readcyclecounter() - dbg_c0;
        g_dbg[1] = __builtin_amdgcn_s_memrealtime() - dbg_r0;
        g_dbg[2] = (unsigned long long)ntl;
        g_dbg[3] = (unsigned long long)nK;
    }
#endif
}

template <bool RES, int SPREAD, int ABL>
static hipError_t launch(const ConvArgs& a, hipStream_t s) {
    static_assert(kSmem <= 160 * 1024, "LDS budget");
    auto kern = k_conv_ps<RES, SPREAD, ABL>;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kSmem);
        if (e != hipSuccess) return e;
        configured = true;
    }
    const int total = a.mtiles * a.ntiles;
    hipLaunchKernelGGL(kern, dim3(total < 256 ? total : 256), dim3(256), kSmem, s, a, total);
    return hipGetLastError();
}

}  // namespace ps

// variant 70 = the kernel; with -DADAYOLO_MEASURE 71 / 72 = measurement builds. hipErrorInvalidValue -> not served.
hipError_t launch_conv_ps(ConvArgs a, hipStream_t s, int variant) {
    const long nK = (long)a.ks * a.ks * a.Cin / 64;
    if (a.Cin % 64 || a.Cout % 128 || a.act != ADAYOLO_ACT_SILU || nK < 4 || a.sh_hw < 0 || a.sh_w < 0) return hipErrorInvalidValue;
    // 32-bit buffer offsets: activations (+ guard + the largest tap offset), outputs, residual, weights
    const unsigned long long lim = 0xFFFFFF00ull - 64;
    const unsigned long long in_b = 2ull * a.B * a.H * a.W * a.in_cs + 4ull * (a.W + 1) * a.in_cs + 2ull * a.Cin;
    if (in_b > lim || 2ull * a.M * a.out_cs > lim || (a.res && 2ull * a.M * a.res_cs > lim) ||
        2ull * a.Cout * a.ks * a.ks * a.Cin > lim)
        return hipErrorInvalidValue;
    a.mtiles = (a.M + ps::BM - 1) / ps::BM;
    a.ntiles = a.Cout / ps::BN;
    const bool wide = nK >= 9;
#ifdef ADAYOLO_MEASURE
    if (variant == 71) return a.res ? ps::launch<true, 8, 1>(a, s) : ps::launch<false, 8, 1>(a, s);
    if (variant == 72) return a.res ? ps::launch<true, 8, 2>(a, s) : ps::launch<false, 8, 2>(a, s);
    if (variant == 73) return a.res ? ps::launch<true, 8, 3>(a, s) : ps::launch<false, 8, 3>(a, s);
#endif
    (void)variant;
    if (a.res) return wide ? ps::launch<true, 8, 0>(a, s) : ps::launch<true, 4, 0>(a, s);
    return wide ? ps::launch<false, 8, 0>(a, s) : ps::launch<false, 4, 0>(a, s);
}

#ifdef ADAYOLO_MEASURE
extern "C" int adayolo_debug_ps(unsigned long long* dst) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(ps::g_dbg), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : -1;
}
#endif

}  // namespace adayolo
