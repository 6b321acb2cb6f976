// Whole Bottleneck of the SHALLOW stages in ONE launch:   out = x + SiLU(b2 + W2 (3x3) * SiLU(b1 + W1 (1x1) * x))
// C = 128 (hidden 64: the two blocks on 184 x 320 maps) and C = 64 (hidden 32: the block on 368 x 640 maps)
// (yolov3/models/common.py:110-120, Bottleneck.forward = x + cv2(cv1(x)); yolov3/models/yolov3.yaml:13-27).
//
// Today these blocks are two launches each — a 1x1 on a ring kernel (or inside k_stem_down) that writes the hidden tensor h, and
// the weights-in-registers 3x3 of yolo_conv_ws.hip that reads it back: per C = 128 block 499 MB of fabric traffic against 241 MB
// algorithmic, the C = 64 block 753 against 482 (profiles/round5_pmc_traffic.json). Here h never leaves the CU. A persistent
// workgroup keeps the 3x3's weights in registers (as yolo_conv_ws.hip: 32 output channels x 9 taps per wave) and, per tile of
// TH x 16 output pixels:
//   A  its (TH + 2) x 18 patch of x arrives by LDS-DMA (buffer descriptor: pixels outside the image are out-of-range offsets,
//      the DMA writes zeros), requested one tile ahead: it lands under the previous tile's 3x3;
//   B  h = SiLU(b1 + W1 x) on the whole patch — W1 stays in LDS for the life of the workgroup — zeroed outside the image (the
//      3x3 pads h, not x), rounded to bf16 as the stand-alone layer stores it, written into the patch layout the 3x3 reads
//      (16-byte chunks keyed by the patch column: conflict-free ds_read_b128 for all nine taps);
//   C  the 3x3 from that patch: a tap is an address offset (yolo_conv_ws.hip's loop), bias + SiLU in the accumulator layout;
//   D  transposed through LDS, + x (re-read from memory: the lines this workgroup fetched for its patch a few microseconds
//      earlier), whole pixel rows stored non-temporally.
// ONE workgroup computes ALL output channels of its pixels (8 waves = NPG pixel groups of 4 tile rows x NCG channel groups of
// 32): the patch and h are staged once per pixel, not once per 64-channel chunk as the stand-alone 3x3 does.
// Both GEMMs accumulate in fp32 over k in the order the stand-alone kernels use.
#include "yolo_internal.h"
#include <type_traits>
#ifndef BWS_PRIO
#define BWS_PRIO 1          // waves 0-3 at priority 2 (yolo_conv_ws.hip's arrangement); 0: all waves equal (measurement)
#endif

namespace adayolo {
namespace bws {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr unsigned kOOB = 0xFFFFFFFFu;
constexpr unsigned kRecords = 0xFFFFFF00u;
constexpr unsigned kDescFlags = 0x00020000u;

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_pk{lo, hi}, bf16x2));
}
__device__ __forceinline__ void barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

struct BwsArgs {
    const unsigned short* x; int x_cs;
    const unsigned short* w1; const float* b1;       // [CM][CX], [CM]
    const unsigned short* w2; const float* b2;       // [CX][3][3][CM], [CX]
    unsigned short* out; int out_cs;
    int B, H, W, tiles_x, tiles_y;
};

// CM = hidden channels (the 3x3's input): 64 or 32; CX = 2 CM = the block's channels
template <int CM>
struct Geo {
    static constexpr int CX = 2 * CM;
    static constexpr int NCG = CX / 32;               // channel groups (waves across the output channels): 4 / 2
    static constexpr int NPG = 8 / NCG;               // pixel groups of 4 tile rows: 2 / 4
    static constexpr int TH = 4 * NPG, TW = 16;       // 8 x 16 / 16 x 16
    static constexpr int PH = TH + 2, PW = TW + 2, PROWS = PH * PW;       // 180 / 324 patch pixels
    static constexpr int NF = (PROWS + 31) / 32;      // pixel fragments of stage B: 6 / 11
    static constexpr int RBX = 2 * CX;                // bytes per x-patch row: 256 / 128
    static constexpr int CPRX = CX / 8;               // its 16-byte chunks: 16 / 8
    static constexpr int RPD = 1024 / RBX;            // x rows per DMA instruction: 4 / 8
    static constexpr int PINS = (PROWS + RPD - 1) / RPD;                  // 45 / 41
    static constexpr int NP = (PINS + 7) / 8;         // DMA instructions per wave: 6
    static constexpr int kXBytes = (NF * 32 > PINS * RPD ? NF * 32 : PINS * RPD) * RBX;   // 49 152 / 45 056 (stage B reads whole 16-pixel groups: <= NF * 32 rows)
    static constexpr int RBH = 2 * CM;                // bytes per h-patch row: 128 / 64
    static constexpr int kHBytes = (PROWS + 15) / 16 * 16 * RBH;          // 24 576 / 21 504 (whole 16-pixel groups: stage B writes its padding rows)
    static constexpr int kOutPitch = 2 * CX + 16;     // 272 / 144
    static constexpr int kOutBytes = TH * TW * kOutPitch;                 // 34 816 / 36 864
    static constexpr int kW1Bytes = CM * RBX;         // 16 384 / 4 096
    static constexpr int KK1 = CX / 16;               // k-steps of the 1x1: 8 / 4
    static constexpr int KK = CM / 16;                // k-steps per tap of the 3x3: 4 / 2
    static constexpr int NU = NF * (CM / 32);         // (pixel fragment, channel fragment) units of stage B: 12 / 11
    static constexpr int oX = 0, oH = oX + kXBytes, oO = oH + kHBytes, oW1 = oO + kOutBytes, oB = oW1 + kW1Bytes;
    static constexpr int kSmem = oB + (CM + CX) * 4;
    static constexpr int NST = TH * TW * (CX / 8) / 512;                  // 16-byte output pieces per thread: 4
};
// swizzle keys. Rows that are a multiple of 256 B apart share every bank: key = row & 15 over 16 chunks; 128-byte rows alternate
// between the two halves of the 64 banks: key = (row >> 1) & 7 over 8 chunks — both conflict-free for the 32 CONSECUTIVE rows of a
// fragment under ds_read_b128's lane groups ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}).
template <int CPR> __device__ __forceinline__ int row_key(int row) { return CPR == 16 ? (row & 15) : ((row >> 1) & 7); }
// the h patch is read by TAPS (a fragment = 2 tile rows x 16 columns + the tap's offset): keyed by the patch column (yolo_conv_ws.hip)
template <int CM> __device__ __forceinline__ int px_key(int x) { return (x >> 1) & (CM / 8 - 1); }

#ifdef ADAYOLO_MEASURE
__device__ unsigned long long g_bws_dbg[16];         // workgroup 0: cycles per phase summed over its tiles, tile count (tools/bneck_ws_stamps.py)
#define BWS_T(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); ph[i] += n_ - tprev; tprev = n_; } while (0)
#else
#define BWS_T(i) do { } while (0)
#endif

template <int CM>
__global__ __launch_bounds__(512) void k_bneck_ws(const BwsArgs a) {
    using G = Geo<CM>;
    constexpr int CX = G::CX, PW = G::PW, PH = G::PH, TW = G::TW, TH = G::TH;
    static_assert(PW == 18, "stage_piece divides by 18 with a multiply");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const xbuf = smem + G::oX;
    unsigned char* const hbuf = smem + G::oH;
    unsigned char* const obuf = smem + G::oO;
    unsigned char* const w1s = smem + G::oW1;
    float* const b1s = reinterpret_cast<float*>(smem + G::oB);
    float* const b2s = b1s + CM;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wave / G::NCG, cg = wave % G::NCG;
#if BWS_PRIO
    if (wave < 4) __builtin_amdgcn_s_setprio(2);          // waves w and w + 4 share a SIMD (same channels, other pixels): see stage C
#endif
    const int ntiles = a.B * a.tiles_y * a.tiles_x;
    const int fq = lane >> 5, fr = lane & 31;

    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, kRecords, kDescFlags);
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, kRecords, kDescFlags);

    // ---- x patch DMA: instruction g = wave + 8 i covers patch rows [g RPD, g RPD + RPD); lane -> (row, 16-byte slot)
    const int dslot0 = lane % G::CPRX, drow0 = lane / G::CPRX;
    // (a piece's (patch y, patch x, source chunk) is recomputed where it is issued — ~8 integer instructions — rather than kept in
    // NP registers across stage C: beside the 144 weight registers they spill, and a scratch reload in front of a DMA instruction
    // waits for every DMA instruction before it)
    auto tile_coords = [&](int t, int& b, int& oy0, int& ox0) {
        const int tx = t % a.tiles_x, r = t / a.tiles_x;
        ox0 = tx * TW; oy0 = (r % a.tiles_y) * TH; b = r / a.tiles_y;
    };
    // one DMA instruction of a patch: `i`-th piece of this wave, for the tile at (b, oy0, ox0); dead = past the last tile
    auto stage_piece = [&](int i, int b, int oy0, int ox0, bool live, int drow, int dslot) __attribute__((always_inline)) {
        if ((wave + 8 * i) < G::PINS) {                  // (uniform)
            const int rr = (wave + 8 * i) * G::RPD + drow;
            const int py = (int)(((unsigned)rr * 3641u) >> 16);            // rr / 18 for rr < 4096 (PW = 18)
            const int px = rr - py * PW;
            const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
            const bool ok = (int)live & (py < PH) & (iy >= 0) & (iy < a.H) & (ix >= 0) & (ix < a.W);
            const unsigned voff = ok ? 2u * (unsigned)(((b * a.H + iy) * a.W + ix) * a.x_cs) + 16u * (unsigned)(dslot ^ row_key<G::CPRX>(rr)) : kOOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr_t)(xbuf + (wave + 8 * i) * 1024), 16, voff, 0, 0, 0);
        }
    };
    auto stage_patch = [&](int t) {
        int b, oy0, ox0;
        tile_coords(t < ntiles ? t : ntiles - 1, b, oy0, ox0);
#pragma unroll
        for (int i = 0; i < G::NP; ++i) stage_piece(i, b, oy0, ox0, t < ntiles, drow0, dslot0);
    };
    int t = blockIdx.x;
    stage_patch(t);                                      // the first patch is in flight while the weights are fetched

    // ---- once per workgroup: W1 -> LDS (rows = hidden channel, chunks keyed by the row), biases -> LDS, 3x3 weights -> registers
    for (int i = tid; i < CM * G::CPRX; i += 512) {
        const int row = i / G::CPRX, c = i % G::CPRX;
        const u32x4 v = *reinterpret_cast<const u32x4*>(a.w1 + (long)row * CX + 8 * c);
        *reinterpret_cast<u32x4*>(w1s + row * G::RBX + ((c ^ row_key<G::CPRX>(row)) << 4)) = v;
    }
    if (tid < CM) b1s[tid] = a.b1[tid];
    else if (tid < CM + CX) b2s[tid - CM] = a.b2[tid - CM];
    bf16x8 wreg[9][G::KK];
    {
        const unsigned short* wp = a.w2 + (long)(cg * 32 + fr) * (9 * CM) + 8 * fq;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int kk = 0; kk < G::KK; ++kk) wreg[tap][kk] = *reinterpret_cast<const bf16x8*>(wp + tap * CM + kk * 16);
    }
    // stage C fragment addressing: MFMA column = pixel fr of a 32-pixel fragment = tile rows (2 pf, 2 pf + 1) of this wave's four
    int prow0[2];
#pragma unroll
    for (int pf = 0; pf < 2; ++pf) prow0[pf] = (pg * 4 + pf * 2 + (fr >> 4)) * PW + (lane & 15);

#ifdef ADAYOLO_MEASURE
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter(), ntl = 0;
#endif
    for (; t < ntiles; t += gridDim.x) {
        int b, oy0, ox0;
        tile_coords(t, b, oy0, ox0);
        // Every VMEM operation of a tile is issued unconditionally, in the order [patch(t + 1)][NST residual loads][NST stores]:
        // patch(t) — requested a tile ago, ahead of that tile's residual loads (consumed since) and stores — is complete when at
        // most the previous tile's NST stores are outstanding.
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::NST) : "memory");
        BWS_T(0);                                        // (stamps: 0 = wait for the patch, 1 = barrier A, 2 = stage B, 3 = barrier B, 4 = patch issue,
        barrier();                                       // A: patch(t) is in LDS (and, first tile: W1 / biases are)
        BWS_T(1);                                        //  5 = stage C, 6 = residual + SiLU + barrier C, 7 = rows + stores)

        // ---- B: h = SiLU(b1 + W1 x) on the patch, on v_mfma_f32_16x16x32_bf16: unit u = (16 patch pixels g, 16 hidden channels
        //      cq) — NG x NCQ = 48 / 42 units, u = wave + 8 uu: SIX per wave (with 32 x 32 MFMAs the 12 / 11 units were two for
        //      half of the waves and one for the others: the stage took two units' time, 3.9k cycles of a 14.6k-cycle tile —
        //      profiles/round6_bneck_ws_stamps.txt). NCQ divides 8, so a wave's units share their channel group: its W1
        //      fragments are read once per tile. Four (two) dependent MFMAs per unit; the units' chains are independent.
        // (lane-derived addressing of this stage: opaque per tile, so that it is recomputed — a dozen integer instructions — rather
        // than hoisted out of the tile loop and spilled beside the 144 weight registers)
        {
            constexpr int NCQ = CM / 16, NG = (G::PROWS + 15) / 16, NUQ = NG * NCQ, KQ = CX / 32;
            static_assert(8 % NCQ == 0, "a wave's units must share their channel group");
            int l15 = lane & 15, kq = lane >> 4;
            asm volatile("" : "+v"(l15), "+v"(kq));
            const int cq = wave % NCQ;
            const int wrow = 16 * cq + l15;               // this lane's row of W1 (a hidden channel)
            const unsigned char* wr = w1s + wrow * G::RBX;
            const int kw = row_key<G::CPRX>(wrow);
            bf16x8 wf[KQ];
#pragma unroll
            for (int k4 = 0; k4 < KQ; ++k4) wf[k4] = *reinterpret_cast<const bf16x8*>(wr + (((4 * k4 + kq) ^ kw) << 4));
            const float4 b4 = *reinterpret_cast<const float4*>(b1s + 16 * cq + 4 * kq);     // D rows 4 kq + (0..3) of the unit's 16 channels
            // Units go THREE at a time, without a branch between them: one unit alone is a latency chain — LDS reads, four dependent
            // MFMAs, exp -> rcp — of ~600 cycles with nothing beside it but the SIMD's other wave (six in a row: the 3.9k cycles
            // this stage took with either MFMA shape, profiles/round6_bneck_ws_stamps.txt); three interleave their reads, their
            // MFMA chains and their SiLUs. A slot past the last unit (C = 64: 42 units in 48 slots) redoes the wave's previous
            // unit — the same values to the same addresses; rows past the patch (the last group's padding) are written too:
            // the h buffer holds whole 16-pixel groups, stage C never reads them.
            constexpr int NB = 3, NSLOT = ((NUQ + 7) / 8 + NB - 1) / NB * NB;
#pragma unroll
            for (int u0 = 0; u0 < NSLOT; u0 += NB) {
                bf16x8 xf[NB][KQ];
                f32x4 hacc[NB];
                int pp[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    int u = wave + 8 * (u0 + j);
                    if (u >= NUQ) u -= 8;                 // (uniform)
                    pp[j] = 16 * (u / NCQ) + l15;        // patch pixel of this lane's MFMA column
                    const unsigned char* xr = xbuf + pp[j] * G::RBX;
                    const int kx = row_key<G::CPRX>(pp[j]);
#pragma unroll
                    for (int k4 = 0; k4 < KQ; ++k4) xf[j][k4] = *reinterpret_cast<const bf16x8*>(xr + (((4 * k4 + kq) ^ kx) << 4));
                    hacc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                }
#pragma unroll
                for (int k4 = 0; k4 < KQ; ++k4)
#pragma unroll
                    for (int j = 0; j < NB; ++j) hacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[k4], xf[j][k4], hacc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    // lane holds pixel p and hidden channels 16 cq + 4 kq + (0..3): 8 bytes — half (kq & 1) of chunk 2 cq + (kq >> 1) of row p
                    const int p = pp[j];
                    const int py = (int)(((unsigned)p * 3641u) >> 16), px = p - py * PW;      // p / 18
                    const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
                    const bool inside = (p < G::PROWS) & (iy >= 0) & (iy < a.H) & (ix >= 0) & (ix < a.W);
                    unsigned lo, hi;
                    bias_act_pack4<true>(hacc[j][0], hacc[j][1], hacc[j][2], hacc[j][3], b4, lo, hi);
                    if (!inside) { lo = 0u; hi = 0u; }
                    *reinterpret_cast<u32x2*>(hbuf + p * G::RBH + (((2 * cq + (kq >> 1)) ^ px_key<CM>(px)) << 4) + 8 * (kq & 1)) = u32x2{lo, hi};
                }
            }
        }
        BWS_T(2);
        barrier();                                       // B: h is complete; nobody reads the x patch any more
        BWS_T(3);
        // the next tile's patch is requested INSIDE stage C, one DMA instruction behind every NSTEP / NP-th MFMA step: issued
        // in one burst here the six instructions hold the wave for ~1.9k cycles (the CU's DMA path takes ~24 B/clk: stamps,
        // profiles/round6_bneck_ws_stamps.txt) with the matrix pipe idle; between MFMA steps the partner wave of the SIMD computes
        const int tn = t + (int)gridDim.x;
        int nb_, noy0, nox0;
        tile_coords(tn < ntiles ? tn : ntiles - 1, nb_, noy0, nox0);
        const bool nlive = tn < ntiles;
        int drowC = drow0, dslotC = dslot0;               // (opaque per tile: the pieces' lane terms are recomputed, not hoisted + spilled)
        asm volatile("" : "+v"(drowC), "+v"(dslotC));
        BWS_T(4);

        // ---- C: the 3x3 from the h patch (yolo_conv_ws.hip's loop). Accumulators start at the bias.
        f32x16 acc[2];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const float4 b4 = *reinterpret_cast<const float4*>(b2s + cg * 32 + 8 * qd + 4 * fq);
#pragma unroll
            for (int pf = 0; pf < 2; ++pf) {
                acc[pf][4 * qd] = b4.x; acc[pf][4 * qd + 1] = b4.y; acc[pf][4 * qd + 2] = b4.z; acc[pf][4 * qd + 3] = b4.w;
            }
        }
        int p0[2] = {prow0[0], prow0[1]}, x0 = lane & 15;
        asm volatile("" : "+v"(p0[0]), "+v"(p0[1]), "+v"(x0));          // (opaque per tile: no 72 hoisted + spilled addresses)
        constexpr int SPT = G::KK / 2, NSTEP = 9 * SPT;
        bf16x8 fr_[2][2][2];                                  // [step parity][k-chunk of the pair][pixel fragment]
        auto load_step = [&](auto stag) __attribute__((always_inline)) {
            constexpr int st = decltype(stag)::value, tap = st / SPT, kp = st % SPT;
            constexpr int toff = (tap / 3) * PW + (tap % 3);
#pragma unroll
            for (int pf = 0; pf < 2; ++pf) {
                const unsigned char* base = hbuf + (p0[pf] + toff) * G::RBH;
                const int key = px_key<CM>(x0 + tap % 3);
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2)
                    fr_[st & 1][k2][pf] = *reinterpret_cast<const bf16x8*>(base + ((((kp * 2 + k2) * 2 + fq) ^ key) << 4));
            }
        };
        auto run_steps = [&](auto self, auto stag) __attribute__((always_inline)) -> void {
            constexpr int st = decltype(stag)::value;
            if constexpr (st < NSTEP) {
                if constexpr (st + 1 < NSTEP) load_step(std::integral_constant<int, st + 1>{});
                constexpr int tap = st / SPT, kp = st % SPT;
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                    for (int pf = 0; pf < 2; ++pf)
                        acc[pf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[tap][kp * 2 + k2], fr_[st & 1][k2][pf], acc[pf], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < G::NP; ++i)
                    if (i * NSTEP / G::NP == st) { stage_piece(i, nb_, noy0, nox0, nlive, drowC, dslotC); __builtin_amdgcn_sched_barrier(0); }
                self(self, std::integral_constant<int, st + 1>{});
            }
        };
        load_step(std::integral_constant<int, 0>{});
        run_steps(run_steps, std::integral_constant<int, 0>{});
        BWS_T(5);
        // residual rows of this thread (16 B = 8 channels of NST pixels): requested now (the fragment registers are dead), they land
        // under the SiLU math — L2 hits, the lines came in with this tile's patch
        unsigned ovoff[G::NST];
        u32x4 rv[G::NST];
#pragma unroll
        for (int it = 0; it < G::NST; ++it) {
            const int idx = it * 512 + tid, pxl = idx / (CX / 8), chunk = idx % (CX / 8);
            const int oy = oy0 + pxl / TW, ox = ox0 + (pxl % TW);
            const bool ok = (oy < a.H) & (ox < a.W);
            const unsigned m = (unsigned)((b * a.H + oy) * a.W + ox);
            ovoff[it] = ok ? 2u * (m * (unsigned)a.out_cs + (unsigned)(chunk * 8)) : kOOB;
            const unsigned rvoff = ok ? 2u * (m * (unsigned)a.x_cs + (unsigned)(chunk * 8)) : kOOB;
            rv[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, rvoff, 0, 0));
        }

        // ---- epilogue math BEFORE the barrier (the prioritised wave of a SIMD computes its SiLUs beside its partner's MFMAs)
#pragma unroll
        for (int pf = 0; pf < 2; ++pf) {
            const int pxl = (pg * 4 + pf * 2 + (fr >> 4)) * TW + (lane & 15);              // pixel index in the tile
            unsigned char* const wr = obuf + pxl * G::kOutPitch + (cg * 32 + 4 * fq) * 2;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const f32x2_pk y0 = silu_pk(f32x2_pk{acc[pf][4 * qd], acc[pf][4 * qd + 1]});
                const f32x2_pk y1 = silu_pk(f32x2_pk{acc[pf][4 * qd + 2], acc[pf][4 * qd + 3]});
                *reinterpret_cast<u32x2*>(wr + 8 * qd * 2) = u32x2{pack_bf16x2(y0.x, y0.y), pack_bf16x2(y1.x, y1.y)};
            }
        }
        barrier();                                       // C: the output tile is complete (and every wave is done with h)
        BWS_T(6);
#pragma unroll
        for (int it = 0; it < G::NST; ++it) {
            const int idx = it * 512 + tid, pxl = idx / (CX / 8), chunk = idx % (CX / 8);
            u32x4 v = *reinterpret_cast<const u32x4*>(obuf + pxl * G::kOutPitch + chunk * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2_pk s = f32x2_pk{__uint_as_float(v[j] << 16), __uint_as_float(v[j] & 0xFFFF0000u)} +
                                   f32x2_pk{__uint_as_float(rv[it][j] << 16), __uint_as_float(rv[it][j] & 0xFFFF0000u)};
                v[j] = pack_bf16x2(s.x, s.y);
            }
            __builtin_amdgcn_raw_buffer_store_b128(v, rsO, ovoff[it], 0, 2);       // nt; a masked pixel is out of range: dropped
        }
        BWS_T(7);
#ifdef ADAYOLO_MEASURE
        ++ntl;
#endif
        // (barrier A / B of the next tile order these LDS reads before its writes to the output tile)
    }
#ifdef ADAYOLO_MEASURE
    if (blockIdx.x == 0 && tid == 0) {
        for (int i = 0; i < 8; ++i) g_bws_dbg[i] = ph[i];
        g_bws_dbg[8] = ntl;
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the tail's out-of-range patch request
}

template <int CM>
static hipError_t launch(BwsArgs a, hipStream_t s) {
    using G = Geo<CM>;
    static_assert(G::kSmem <= 160 * 1024, "LDS budget");
    auto kern = k_bneck_ws<CM>;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::kSmem);
        if (e != hipSuccess) return e;
        configured = true;
    }
    a.tiles_x = (a.W + G::TW - 1) / G::TW;
    a.tiles_y = (a.H + G::TH - 1) / G::TH;
    const long ntiles = (long)a.B * a.tiles_x * a.tiles_y;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0) cus = 256;
    const long grid = ntiles < cus ? ntiles : cus;       // one persistent workgroup per CU
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), G::kSmem, s, a);
    return hipGetLastError();
}

#ifdef ADAYOLO_MEASURE
extern "C" int adayolo_debug_bws(unsigned long long* dst) {           // measurement helper, not part of the ABI
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_bws_dbg), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#endif
}  // namespace bws

// C = 128 or 64 (hipErrorInvalidValue otherwise); 32-bit byte offsets into x and out
hipError_t launch_bottleneck_ws(const void* x, int x_cs, const void* w1, const float* b1, const void* w2, const float* b2,
                                void* out, int out_cs, int B, int H, int W, int C, hipStream_t s) {
    if (C != 128 && C != 64) return hipErrorInvalidValue;
    if (2ull * B * H * W * x_cs + 256 > 0xFFFFFF00ull || 2ull * B * H * W * out_cs + 256 > 0xFFFFFF00ull) return hipErrorInvalidValue;
    bws::BwsArgs a;
    a.x = static_cast<const unsigned short*>(x); a.x_cs = x_cs;
    a.w1 = static_cast<const unsigned short*>(w1); a.b1 = b1;
    a.w2 = static_cast<const unsigned short*>(w2); a.b2 = b2;
    a.out = static_cast<unsigned short*>(out); a.out_cs = out_cs;
    a.B = B; a.H = H; a.W = W; a.tiles_x = a.tiles_y = 0;
    return C == 128 ? bws::launch<64>(a, s) : bws::launch<32>(a, s);
}

}  // namespace adayolo
