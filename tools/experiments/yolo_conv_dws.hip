// 3x3 STRIDE-2 Conv 64 -> 128 + bias + SiLU (the down-sampling conv between the C = 64 and the C = 128 stage: yolov3.yaml layer 3,
// Conv.forward_fuse, yolov3/models/common.py:45-59) — weights stationary in REGISTERS, the input as a patch in LDS, persistent
// workgroups: yolo_conv_ws.hip's scheme for stride 2 (variant 95).
//
// The two-workgroup ring kernel runs this layer at 93 us: 361 MB at 3.9 TB/s, 400 TFLOP/s — K = 576 is nine k-tiles of ring
// structure per 256 x 128 tile. Here, per tile of 4 x 16 output pixels:
//   * the 9 x 33 input patch arrives by LDS-DMA (buffer descriptor: out-of-image pixels are out-of-range offsets, the DMA writes
//     zeros), DE-INTERLEAVED by column parity: a tap's columns 2 ox + kw are then CONSECUTIVE entries of one parity plane
//     (kw = 0 -> even plane entry ox, kw = 1 -> odd plane entry ox, kw = 2 -> even plane entry ox + 1), i.e. the stride-2 walk is a
//     stride-1 walk in LDS; layout [patch row][parity][20 entries] x 128 B with 16-byte chunks XOR-keyed by (row >> 1) & 7: the two
//     output rows of a fragment are 80 LDS rows apart (= 0 mod 16), so ds_read_b128's lane groups meet 16 distinct bank slots for
//     every tap, even or odd start entry (checked by hand in DESIGN 4.2);
//   * two patch buffers: the next tile's DMA instructions are issued between the MFMA steps of this one (as in yolo_bneck_ws.hip);
//   * 8 waves = 2 pixel groups (2 output rows = 32 px = one 32 x 32 fragment) x 4 channel groups (32 ch: 36 weight fragments = 144
//     registers, loaded once per workgroup); 36 MFMAs per wave and tile; bias + SiLU in the accumulator layout, transposed through
//     LDS, whole 256-byte pixel rows stored non-temporally.
// Restrictions (hipErrorInvalidValue otherwise: the caller's default kernel runs): ksize 3, stride 2, Cin 64, Cout 128, SiLU, no
// residual, 32-bit byte offsets.
#include "yolo_internal.h"
#include <type_traits>

namespace adayolo {
namespace dws {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
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

constexpr int CIN = 64, COUT = 128;
constexpr int TH = 4, TW = 16;                       // output tile
constexpr int PR = 2 * TH + 1;                      // input patch: 9 rows x (2 TW + 1 =) 33 columns
constexpr int ES = 20;                               // entries per (patch row, parity): 17 / 16 used; 2 * ES * 2 = 80 = 0 mod 16
constexpr int LROWS = PR * 2 * ES;                   // 360 LDS rows of 128 B
constexpr int RB = CIN * 2;                          // 128 B per row (one pixel)
constexpr int PINS = LROWS / 8;                      // 45 DMA instructions per patch (8 rows each)
constexpr int NP = (PINS + 7) / 8;                   // per wave: 6
constexpr int kPBytes = LROWS * RB;                  // 46 080
constexpr int kOutPitch = COUT * 2 + 16;             // 272
constexpr int kOutBytes = TH * TW * kOutPitch;       // 17 408
constexpr int oP0 = 0, oP1 = kPBytes, oO = 2 * kPBytes, oB = oO + kOutBytes;
constexpr int kSmem = oB + COUT * 4;                 // 110 080
constexpr int KK = CIN / 16;                         // 4 k-steps per tap
constexpr int NST = TH * TW * (COUT / 8) / 512;      // 16-byte output pieces per thread: 2

__device__ __forceinline__ int row_key(int row) { return (row >> 1) & 7; }

__global__ __launch_bounds__(512) void k_conv_dws(const ConvArgs a, const int tiles_x, const int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const obuf = smem + oO;
    float* const bias_s = reinterpret_cast<float*>(smem + oB);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wave >> 2, cg = wave & 3;             // pixel group (output rows 2 pg, 2 pg + 1), channel group (32 ch)
    if (wave < 4) __builtin_amdgcn_s_setprio(2);
    const int ntiles = a.B * tiles_y * tiles_x;
    const int fq = lane >> 5, fr = lane & 31;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, kRecords, kDescFlags);
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, kRecords, kDescFlags);

    auto tile_coords = [&](int t, int& b, int& oy0, int& ox0) {
        const int tx = t % tiles_x, r = t / tiles_x;
        ox0 = tx * TW; oy0 = (r % tiles_y) * TH; b = r / tiles_y;
    };
    // one DMA instruction: LDS rows [8 g, 8 g + 8) of the patch buffer, g = wave + 8 i; lane -> (row, 16-byte slot). A row is
    // (patch row pr, parity pl, entry e): input pixel (2 oy0 - 1 + pr, 2 ox0 - 1 + 2 e + pl); entries past the patch (e >= 17 / 16)
    // and pixels outside the image are out-of-range offsets (zeros)
    auto stage_piece = [&](int i, unsigned char* dst, int b, int oy0, int ox0, bool live, int drow, int dslot) __attribute__((always_inline)) {
        const int g = wave + 8 * i;
        if (g < PINS) {                                  // (uniform)
            const int R = g * 8 + drow;
            const int pr = (int)(((unsigned)R * 1639u) >> 16);               // R / 40 for R < 4096
            const int rem = R - pr * 40;
            const int pl = rem >= ES ? 1 : 0, e = rem - pl * ES;
            const int iy = 2 * oy0 - 1 + pr, ix = 2 * ox0 - 1 + 2 * e + pl;
            const bool ok = (int)live & (e < 17 - pl) & (iy >= 0) & (iy < a.H) & (ix >= 0) & (ix < a.W);
            const unsigned voff = ok ? 2u * (unsigned)(((b * a.H + iy) * a.W + ix) * a.in_cs) + 16u * (unsigned)(dslot ^ row_key(R)) : kOOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(dst + g * 1024), 16, voff, 0, 0, 0);
        }
    };
    const int dslot0 = lane & 7, drow0 = lane >> 3;
    int t = blockIdx.x;
    {
        int b, oy0, ox0;
        tile_coords(t < ntiles ? t : ntiles - 1, b, oy0, ox0);
#pragma unroll
        for (int i = 0; i < NP; ++i) stage_piece(i, smem + oP0, b, oy0, ox0, t < ntiles, drow0, dslot0);
    }
    // ---- once per workgroup: bias -> LDS, weights -> registers (fragment (tap, kk) of channel 32 cg + fr, k-chunk fq)
    if (tid < COUT) bias_s[tid] = a.bias[tid];
    bf16x8 wreg[9][KK];
    {
        const unsigned short* wp = a.w + (long)(cg * 32 + fr) * (9 * CIN) + 8 * fq;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) wreg[tap][kk] = *reinterpret_cast<const bf16x8*>(wp + tap * CIN + kk * 16);
    }
    // fragment addressing: MFMA column = pixel fr = (output row 2 pg + (fr >> 4), output column fr & 15); its tap (kh, kw) sits
    // in LDS row ((2 oyl + kh) * 2 + (kw & 1)) * ES + oxl + (kw >> 1)
    const int oyl = 2 * pg + (fr >> 4);
    const int row00 = (2 * oyl) * 2 * ES + (fr & 15);

    int cur = 0;
    for (; t < ntiles; t += gridDim.x, cur ^= 1) {
        unsigned char* const pb = smem + (cur ? oP1 : oP0);
        unsigned char* const pn = smem + (cur ? oP0 : oP1);
        int b, oy0, ox0;
        tile_coords(t, b, oy0, ox0);
        // VMEM order per tile: [patch(t + 1) pieces, inside the MFMA loop][NST stores]: patch(t) — requested a tile ago, ahead of that
        // tile's stores — is complete when at most the previous tile's NST stores are outstanding
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
        barrier();                                       // A: patch(t) is in LDS; the output tile of t - 1 has been read
        const int tn = t + (int)gridDim.x;
        int nb_, noy0, nox0;
        tile_coords(tn < ntiles ? tn : ntiles - 1, nb_, noy0, nox0);
        const bool nlive = tn < ntiles;
        int drowC = drow0, dslotC = dslot0, r00 = row00;   // (opaque per tile: recomputed, not hoisted + spilled beside the weights)
        asm volatile("" : "+v"(drowC), "+v"(dslotC), "+v"(r00));

        f32x16 acc;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const float4 b4 = *reinterpret_cast<const float4*>(bias_s + cg * 32 + 8 * qd + 4 * fq);
            acc[4 * qd] = b4.x; acc[4 * qd + 1] = b4.y; acc[4 * qd + 2] = b4.z; acc[4 * qd + 3] = b4.w;
        }
        // steps of 4 MFMAs (one tap: its four k-chunks); the fragments of step s + 1 are requested before the MFMAs of step s
        constexpr int NSTEP = 9;
        bf16x8 frg[2][KK];
        auto load_step = [&](auto stag) __attribute__((always_inline)) {
            constexpr int tap = decltype(stag)::value, kh = tap / 3, kw = tap % 3;
            const int R = r00 + (kh * 2 + (kw & 1)) * ES + (kw >> 1);
            const unsigned char* base = pb + R * RB;
            const int key = row_key(R);
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) frg[tap & 1][kk] = *reinterpret_cast<const bf16x8*>(base + (((2 * kk + fq) ^ key) << 4));
        };
        auto run_steps = [&](auto self, auto stag) __attribute__((always_inline)) -> void {
            constexpr int st = decltype(stag)::value;
            if constexpr (st < NSTEP) {
                if constexpr (st + 1 < NSTEP) load_step(std::integral_constant<int, st + 1>{});
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[st][kk], frg[st & 1][kk], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NP; ++i)
                    if (i * NSTEP / NP == st) { stage_piece(i, pn, nb_, noy0, nox0, nlive, drowC, dslotC); __builtin_amdgcn_sched_barrier(0); }
                self(self, std::integral_constant<int, st + 1>{});
            }
        };
        load_step(std::integral_constant<int, 0>{});
        run_steps(run_steps, std::integral_constant<int, 0>{});

        // ---- bias (in the accumulator) + SiLU -> output tile in LDS: lane holds pixel fr and channels 32 cg + 8 qd + 4 fq + (0..3)
        {
            unsigned char* const wr = obuf + (16 * oyl + (fr & 15)) * kOutPitch + (cg * 32 + 4 * fq) * 2;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const f32x2_pk y0 = silu_pk(f32x2_pk{acc[4 * qd], acc[4 * qd + 1]});
                const f32x2_pk y1 = silu_pk(f32x2_pk{acc[4 * qd + 2], acc[4 * qd + 3]});
                *reinterpret_cast<u32x2*>(wr + 8 * qd * 2) = u32x2{pack_bf16x2(y0.x, y0.y), pack_bf16x2(y1.x, y1.y)};
            }
        }
        barrier();                                       // B: the output tile is complete (and every wave is done with patch(t))
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int idx = it * 512 + tid, pxl = idx >> 4, chunk = idx & 15;
            const int oy = oy0 + (pxl >> 4), ox = ox0 + (pxl & 15);
            const bool ok = (oy < a.Ho) & (ox < a.Wo);
            const unsigned voff = ok ? 2u * ((unsigned)((b * a.Ho + oy) * a.Wo + ox) * (unsigned)a.out_cs + (unsigned)(chunk * 8)) : kOOB;
            const u32x4 v = *reinterpret_cast<const u32x4*>(obuf + pxl * kOutPitch + chunk * 16);
            __builtin_amdgcn_raw_buffer_store_b128(v, rsO, voff, 0, 2);           // nt; a masked pixel is out of range: dropped
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the tail's out-of-range patch request
}

static hipError_t launch(ConvArgs a, hipStream_t s) {
    static_assert(kSmem <= 160 * 1024, "LDS budget");
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_dws), hipFuncAttributeMaxDynamicSharedMemorySize, kSmem);
        if (e != hipSuccess) return e;
        configured = true;
    }
    const int tiles_x = (a.Wo + TW - 1) / TW, tiles_y = (a.Ho + TH - 1) / TH;
    const long ntiles = (long)a.B * tiles_y * tiles_x;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0) cus = 256;
    const long grid = ntiles < cus ? ntiles : cus;
    hipLaunchKernelGGL(k_conv_dws, dim3((unsigned)grid), dim3(512), kSmem, s, a, tiles_x, tiles_y);
    return hipGetLastError();
}

}  // namespace dws

// variant 95 (hipErrorInvalidValue -> the shape is not served, the caller falls back)
hipError_t launch_conv_dws(ConvArgs a, hipStream_t s, int variant) {
    (void)variant;
    if (a.ks != 3 || a.stride != 2 || a.Cin != 64 || a.Cout != 128 || a.act != ADAYOLO_ACT_SILU || a.res) return hipErrorInvalidValue;
    if (2ull * a.B * a.H * a.W * a.in_cs + 256 > 0xFFFFFF00ull || 2ull * a.M * a.out_cs + 256 > 0xFFFFFF00ull) return hipErrorInvalidValue;
    return dws::launch(a, s);
}

}  // namespace adayolo
