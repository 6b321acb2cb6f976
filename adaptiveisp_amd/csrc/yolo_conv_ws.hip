// 3x3 stride-1 Conv + bias + SiLU (+ residual) for the SHALLOW layers (Cin = 32 or 64) — weights stationary in
// REGISTERS, activations as a patch in LDS, persistent workgroups.
//
// What bounds the implicit-GEMM kernels is the L2 -> LDS byte rate a CU sustains (~27 B/clk); at
// Cin = 64 a 256 x 128 tile moves 442 KB for 38 MFLOP — the early layers run at 430-640 TFLOP/s, 1.5-2x above their HBM
// floor. Here almost nothing crosses that path:
//   * K = 9 Cin <= 576: the weights of a wave's 32 output channels are 18 / 36 MFMA operand fragments = 72 / 144
//     registers, loaded ONCE per workgroup; a workgroup keeps its 64 output channels for every tile it processes;
//   * the activations of a 16 x 16 pixel tile are its 18 x 18 input patch (Cin = 64: 41 KB) staged once by LDS-DMA (buffer
//     descriptor: pixels outside the image are out-of-range offsets, the DMA writes zeros) and read by all nine taps —
//     a tap is a shifted patch row, i.e. address arithmetic; two patch buffers, the next tile's patch lands under this
//     tile's MFMAs;
//   * 8 waves = 4 pixel groups (4 tile rows = 64 px) x 2 channel groups (32 ch): per tile 72 MFMAs and 72 ds_read_b128
//     per wave (Cin = 64), two waves per SIMD; the epilogue goes through one LDS tile (bias + SiLU + bf16 in the
//     accumulator layout, then 128-byte pixel rows + residual), two barriers per tile.
// Restrictions (the launcher falls back otherwise): ksize 3, stride 1, Cin in {32, 64}, Cout % 64 == 0, 32-bit offsets.
// Measured and dropped (round 2, parity-green): the two channel groups skewed by a whole phase across tiles (group 1 one
// barrier late; in every barrier interval one wave of a SIMD is in its MFMA phase, the other in its SiLU phase; two output
// tiles; the rows of tile i - 1 leave in the SiLU phase of tile i). Per tile 9.8k cycles against 9.4k here (Cin = 64): side by
// side the two phases take 3.5k each (alone 3.0k and 2.8k), the patch issue (1.3k) lands in front of a group's MFMAs, and the
// within-tile skew below already overlaps one SiLU phase of two.
#include "yolo_internal.h"
#include <type_traits>

namespace adayolo {
namespace ws {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int TH = 16, TW = 16, PH = TH + 2, PW = TW + 2, PROWS = PH * PW;   // 18 x 18 = 324 patch pixels
constexpr int BN = 64;                               // output channels per workgroup
constexpr int kOutPitch = 144;                       // bytes per pixel row of the output tile (64 ch + pad)
constexpr int kOutBytes = TH * TW * kOutPitch;       // 36 KB
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
// swizzle key of a patch pixel: a function of its x coordinate in the 18-wide patch, (x >> 1) & (chunks per row - 1).
// A ds_read_b128 lane group holds pixels x in {0-3, 12-15} of one tile row and {4-11} of the next (+ the tap's kw): keyed
// by the linear patch row, rows 16 apart share a bank slot (2-way conflicts on every fragment read: the LDS time doubles
// and equals the MFMA time); keyed by x every group is conflict-free for all nine taps (checked exhaustively).
template <int CIN>
__device__ __forceinline__ int px_key(int x) { return (x >> 1) & (CIN / 8 - 1); }

#ifdef ADAYOLO_MEASURE
__device__ unsigned long long g_ws_dbg[16];          // workgroup 0: cycles per phase summed over its tiles, tile count
#define WS_T(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); ph[i] += n_ - tprev; tprev = n_; } while (0)
#else
#define WS_T(i) do { } while (0)
#endif
template <int CIN, bool RES>
__global__ __launch_bounds__(512) void k_conv_ws(const ConvArgs a, const int tiles_x, const int tiles_y, const int nchunks) {
    constexpr int RB = CIN * 2;                      // bytes per patch row (one pixel)
    constexpr int CH = CIN / 8;                      // 16-byte chunks per row
    constexpr int RPD = 64 / CH;                     // patch rows per DMA instruction (8 / 16)
    constexpr int PINS = (PROWS + RPD - 1) / RPD;    // DMA instructions per patch (41 / 21)
    constexpr int NP = (PINS + 7) / 8;               // ... per wave (6 / 3)
    constexpr int PBYTES = PINS * RPD * RB;          // 41 984 / 21 504
    constexpr int KK = CIN / 16;                     // MFMA k-steps per tap (4 / 2)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const pbuf0 = smem;
    unsigned char* const obuf = smem + 2 * PBYTES;
    float* const bias_s = reinterpret_cast<float*>(smem + 2 * PBYTES + kOutBytes);   // [64]: re-read per tile (16 registers)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wave & 3, cg = wave >> 2;         // pixel group (tile rows 4 pg .. 4 pg + 3), channel group (32 ch)
    if (cg == 0) __builtin_amdgcn_s_setprio(2);      // waves w and w + 4 share a SIMD: see the epilogue
    // the workgroup's channel chunk is fixed (weights stay in registers); it strides through the pixel tiles
    const int nwg_per_chunk = gridDim.x / nchunks;
    // the workgroups that take the same pixel tiles for different channel chunks sit on ONE XCD (block b runs on XCD b % 8):
    // the second one reads the patch from that XCD's L2
    int chunk, slot0;
    if ((nwg_per_chunk & 7) == 0) {
        const int q = blockIdx.x >> 3;
        chunk = q % nchunks;
        slot0 = (q / nchunks) * 8 + (blockIdx.x & 7);
    } else {
        chunk = blockIdx.x % nchunks;
        slot0 = blockIdx.x / nchunks;
    }
    const int n0 = chunk * BN;
    const int ntiles = a.B * tiles_y * tiles_x;

    // ---- weights: fragment (tap, kk) of channel 32 cg + (lane & 31): k-chunk (lane >> 5) -> 8 consecutive input channels
    bf16x8 wreg[9][KK];
    {
        const unsigned short* wp = a.w + (long)(n0 + cg * 32 + (lane & 31)) * (9 * CIN) + 8 * (lane >> 5);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) wreg[tap][kk] = *reinterpret_cast<const bf16x8*>(wp + tap * CIN + kk * 16);
    }
    if (tid < BN) bias_s[tid] = a.bias[n0 + tid];     // (visible behind the first barrier of the tile loop)

    // ---- patch DMA: piece i of this wave covers patch rows [(wave + 8 i) RPD, + RPD); lane -> row, 16-byte slot
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, kRecords, kDescFlags);
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, kRecords, kDescFlags);
    const unsigned long long resp = (unsigned long long)(RES ? (const void*)a.res : (const void*)a.out);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)resp, 0, kRecords, kDescFlags);
    constexpr int NPMIN = PINS / 8;                  // DMA pieces EVERY wave issues per patch
    const int dslot = lane % CH, drow = lane / CH;
    unsigned pinfo[NP];                              // this lane's row in piece i: patch y << 16 | patch x << 8 | source chunk
#pragma unroll                                       // (after the swizzle); y >= PH: past the end of the patch
    for (int i = 0; i < NP; ++i) {
        const int rr = (wave + 8 * i) * RPD + drow;
        const int py = rr / PW;
        pinfo[i] = ((unsigned)py << 16) | ((unsigned)(rr - py * PW) << 8) | (unsigned)(dslot ^ px_key<CIN>(rr - py * PW));
    }
    auto tile_coords = [&](int t, int& b, int& oy0, int& ox0) {
        const int tx = t % tiles_x, r = t / tiles_x;
        ox0 = tx * TW; oy0 = (r % tiles_y) * TH; b = r / tiles_y;
    };
    auto stage_patch = [&](int t, unsigned char* dst) {
        int b, oy0, ox0;
        tile_coords(t < ntiles ? t : ntiles - 1, b, oy0, ox0);
        const bool live = t < ntiles;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if ((wave + 8 * i) < PINS) {             // (uniform) the last instruction slot of some waves is past the patch
                const int py = (int)(pinfo[i] >> 16), px = (int)((pinfo[i] >> 8) & 0xFFu);
                const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
                const bool ok = (int)live & (py < PH) & (iy >= 0) & (iy < a.H) & (ix >= 0) & (ix < a.W);
                const unsigned voff = ok ? 2u * (unsigned)(((b * a.H + iy) * a.W + ix) * a.in_cs) + 16u * (pinfo[i] & 0xFFu) : kOOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(dst + (wave + 8 * i) * RPD * RB), 16, voff, 0, 0, 0);
            }
        }
    };

    // ---- fragment addressing: MFMA column = pixel (lane & 31) of a 32-pixel fragment = tile rows (2 pf, 2 pf + 1) of this
    //      wave's four, 16 px each; k-chunk fq = lane >> 5
    const int fq = lane >> 5;
    int prow0[2];                                    // patch row of the pixel's (kh = 0, kw = 0) tap
#pragma unroll
    for (int pf = 0; pf < 2; ++pf) prow0[pf] = (pg * 4 + pf * 2 + ((lane & 31) >> 4)) * PW + (lane & 15);

    int t = slot0;
    stage_patch(t, pbuf0);
    stage_patch(t + nwg_per_chunk, pbuf0 + PBYTES);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPMIN) : "memory");       // the first patch (the second one stays in flight)
    int cur = 0;
#ifdef ADAYOLO_MEASURE
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter(), ntl = 0;
#endif
    for (; t < ntiles; t += nwg_per_chunk, cur ^= 1) {
        unsigned char* const pb = pbuf0 + cur * PBYTES;
        int b, oy0, ox0;
        tile_coords(t, b, oy0, ox0);
        // Every VMEM operation of a tile is issued unconditionally (masked pixels are out-of-range buffer offsets), in the
        // order [4 residual loads][patch(t + 2): >= NPMIN pieces][4 stores], so the counted wait is exact: patch(t) — requested
        // a whole tile ago — is complete when at most this tile's predecessor's patch pieces and stores are outstanding. A
        // plain vmcnt(0) here waits for the previous tile's STORES to be acknowledged by HBM.
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPMIN + 4) : "memory");
        WS_T(0);
        barrier();
        WS_T(1);
        f32x16 acc[2];                                // start at the bias: channels 8 qd + 4 fq + (0..3) of the wave's 32
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const float4 b4 = *reinterpret_cast<const float4*>(bias_s + cg * 32 + 8 * qd + 4 * fq);
#pragma unroll
            for (int pf = 0; pf < 2; ++pf) {
                acc[pf][4 * qd] = b4.x; acc[pf][4 * qd + 1] = b4.y; acc[pf][4 * qd + 2] = b4.z; acc[pf][4 * qd + 3] = b4.w;
            }
        }
        // (the fragment addresses of all nine taps are loop-invariant: left visible, hipcc hoists 72 of them out of the tile
        // loop, spills them and reloads each one — scratch_load + vmcnt(0) — in front of its MFMA. Opaque per tile: ~10 VALU
        // per tap in the loop instead.)
        int p0[2] = {prow0[0], prow0[1]}, x0 = lane & 15;
        asm volatile("" : "+v"(p0[0]), "+v"(p0[1]), "+v"(x0));
        // steps of 4 MFMAs (one pair of k-chunks x the two pixel fragments); the four fragments of step s + 1 are requested
        // before the MFMAs of step s, and a fence per step keeps hipcc from hoisting more (all 72 reads next to the 144 weight
        // registers spill)
        constexpr int SPT = KK / 2, NSTEP = 9 * SPT;          // steps per tap, steps per tile
        bf16x8 fr[2][2][2];                                   // [step parity][k-chunk of the pair][pixel fragment]
        auto load_step = [&](auto stag) __attribute__((always_inline)) {
            constexpr int st = decltype(stag)::value, tap = st / SPT, kp = st % SPT;
            constexpr int toff = (tap / 3) * PW + (tap % 3);
#pragma unroll
            for (int pf = 0; pf < 2; ++pf) {
                const int rr = p0[pf] + toff;
                const unsigned char* base = pb + rr * RB;
                const int key = px_key<CIN>(x0 + tap % 3);
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2)
                    fr[st & 1][k2][pf] = *reinterpret_cast<const bf16x8*>(base + ((((kp * 2 + k2) * 2 + fq) ^ key) << 4));
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
                        acc[pf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[tap][kp * 2 + k2], fr[st & 1][k2][pf], acc[pf], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                self(self, std::integral_constant<int, st + 1>{});
            }
        };
        load_step(std::integral_constant<int, 0>{});
        run_steps(run_steps, std::integral_constant<int, 0>{});
        WS_T(2);
        // this thread's four output rows: 16 B (8 channels) of pixel (tid >> 3) + 64 it; the residual rows are requested now
        // (the fragment registers are dead) and land under the SiLU math. (Intrinsic loads, the compiler places the wait: an
        // inline-asm load with a hand-placed counted wait is only correct as long as the register allocator never moves the
        // destination registers in between — it did in a sibling of this kernel — and measured no faster: 98.2 vs 96.0 us.)
        unsigned ovoff[4];
        u32x4 rv[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int px = (tid >> 3) + 64 * it;
            const int oy = oy0 + px / TW, ox = ox0 + (px % TW);
            const bool ok = (oy < a.Ho) & (ox < a.Wo);
            const unsigned m = (unsigned)((b * a.Ho + oy) * a.Wo + ox);
            ovoff[it] = ok ? 2u * (m * (unsigned)a.out_cs + (unsigned)(n0 + (tid & 7) * 8)) : kOOB;
            if (RES) {
                const unsigned rvoff = ok ? 2u * (m * (unsigned)a.res_cs + (unsigned)(n0 + (tid & 7) * 8)) : kOOB;
                rv[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, rvoff, 0, 0));
            }
        }
        // ---- epilogue math BEFORE the barrier: the two waves of a SIMD are the two channel groups of the same pixels and the
        //      first group has priority (s_setprio above), so it finishes its MFMA steps first and computes its SiLUs beside
        //      the second group's MFMAs instead of behind the barrier.
        //      D[row = channel][col = pixel]; lane holds pixel (lane & 31) and channels 8 qd + 4 fq + (0..3)
#pragma unroll
        for (int pf = 0; pf < 2; ++pf) {
            const int px = (pg * 4 + pf * 2 + ((lane & 31) >> 4)) * TW + (lane & 15);          // pixel index in the tile
            unsigned char* const wr = obuf + px * kOutPitch + (cg * 32 + 4 * fq) * 2;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const f32x2_pk x0 = silu_pk(f32x2_pk{acc[pf][4 * qd], acc[pf][4 * qd + 1]});
                const f32x2_pk x1 = silu_pk(f32x2_pk{acc[pf][4 * qd + 2], acc[pf][4 * qd + 3]});
                *reinterpret_cast<u32x2*>(wr + 8 * qd * 2) = u32x2{pack_bf16x2(x0.x, x0.y), pack_bf16x2(x1.x, x1.y)};
            }
        }
        WS_T(3);
        barrier();                                   // every wave has read patch(t) (its buffer takes patch(t + 2)) and written its
        stage_patch(t + 2 * nwg_per_chunk, pb);      // part of the output tile
        WS_T(4);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int px = (tid >> 3) + 64 * it;
            u32x4 v = *reinterpret_cast<const u32x4*>(obuf + px * kOutPitch + (tid & 7) * 16);
            if (RES) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x2_pk x = f32x2_pk{__uint_as_float(v[j] << 16), __uint_as_float(v[j] & 0xFFFF0000u)} +
                                       f32x2_pk{__uint_as_float(rv[it][j] << 16), __uint_as_float(rv[it][j] & 0xFFFF0000u)};
                    v[j] = pack_bf16x2(x.x, x.y);
                }
            }
            __builtin_amdgcn_raw_buffer_store_b128(v, rsO, ovoff[it], 0, 2);       // nt; a masked pixel is out of range: dropped
        }
        WS_T(5);
#ifdef ADAYOLO_MEASURE
        ++ntl;
#endif
        // (the next iteration's first barrier orders these LDS reads before the next tile's output writes)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tail's out-of-range patch requests
#ifdef ADAYOLO_MEASURE
    if (blockIdx.x == 0 && tid == 0) {
        for (int i = 0; i < 6; ++i) g_ws_dbg[i] = ph[i];
        g_ws_dbg[6] = ntl;
    }
#endif
}

template <int CIN, bool RES>
static hipError_t launch(ConvArgs a, hipStream_t s) {
    constexpr int RPD = 64 / (CIN / 8), PINS = (PROWS + RPD - 1) / RPD, PBYTES = PINS * RPD * CIN * 2;
    constexpr int smem = 2 * PBYTES + kOutBytes + BN * 4;
    static_assert(smem <= 160 * 1024, "LDS budget");
    auto kern = k_conv_ws<CIN, RES>;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        configured = true;
    }
    const int tiles_x = (a.Wo + TW - 1) / TW, tiles_y = (a.Ho + TH - 1) / TH;
    const int nchunks = a.Cout / BN;
    const long ntiles = (long)a.B * tiles_y * tiles_x;
    // one workgroup per CU (two for the 32-channel form would fit; the layer is HBM-bound either way); every channel chunk
    // gets the same number of workgroups
    long per_chunk = 256 / nchunks;
    if (per_chunk < 1) per_chunk = 1;
    if (per_chunk > ntiles) per_chunk = ntiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)(per_chunk * nchunks)), dim3(512), smem, s, a, tiles_x, tiles_y, nchunks);
    return hipGetLastError();
}

#ifdef ADAYOLO_MEASURE
extern "C" int adayolo_debug_ws(unsigned long long* dst) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_ws_dbg), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#endif
}  // namespace ws

// variant 90 (hipErrorInvalidValue -> the shape is not served, the caller falls back)
hipError_t launch_conv_ws(ConvArgs a, hipStream_t s, int variant) {
    (void)variant;
    if (a.ks != 3 || a.stride != 1 || (a.Cin != 32 && a.Cin != 64) || a.Cout % 64 || a.act != ADAYOLO_ACT_SILU)
        return hipErrorInvalidValue;
    if (2ull * a.B * a.H * a.W * a.in_cs + 256 > 0xFFFFFF00ull || 2ull * a.M * a.out_cs + 256 > 0xFFFFFF00ull ||
        (a.res && 2ull * a.M * a.res_cs + 256 > 0xFFFFFF00ull))
        return hipErrorInvalidValue;
    if (a.Cin == 32) return a.res ? ws::launch<32, true>(a, s) : ws::launch<32, false>(a, s);
    return a.res ? ws::launch<64, true>(a, s) : ws::launch<64, false>(a, s);
}

}  // namespace adayolo
