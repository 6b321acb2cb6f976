// Fused detector head: letterbox + stem Conv(3->32, k3 s1) + SiLU + first down-sampling Conv(32->64, k3 s2) + SiLU.
//
// The stem's output is the largest tensor of the network (8 x 736 x 1280 x 32 bf16 = 482 MB): written once and read once,
// it makes the stem (170 us) and the first conv (240 us) HBM-bound at ~3 TB/s. Here it never leaves the CU: a
// workgroup owns an 8 x 16 tile of the SECOND conv's output, computes the 17 x 33 stem pixels under it into LDS
// (10 % of them are recomputed by a neighbour) and runs the 3x3 stride-2 conv from there.
//   A  image patch 19 x 35 -> LDS as (R, G, B, 0) bf16 per pixel (letterbox value / zero padding resolved here, like k_stem)
//   B  stem conv on the patch: per kernel row K = 3 columns x 4 = 12 (16 with the zero k-group) = one
//      v_mfma_f32_16x16x16_bf16 per 16 pixels, channel half and kernel row — the operand is ONE aligned 8-byte LDS read, no
//      gather / select / pack (weights in registers, same row permutation as k_stem: a lane owns 8 consecutive channels), bias + SiLU,
//      bf16, 16-byte chunks XOR-swizzled by the pixel's column; stem pixels outside the frame are the second conv's zero pad
//   C  second conv: K = 9 taps x 32 ch = 18 steps of v_mfma_f32_32x32x16_bf16; a wave owns 64 px x 32 ch, its 72
//      weight registers are loaded once; activation fragments are gathered from the stem patch (stride-2 pixel walk)
//   D  bias + SiLU + bf16, transposed through LDS, 128 contiguous bytes per pixel
//   E  (optional) the following 1x1 conv (Bottleneck.cv1, 64 -> 32) + SiLU from the output tile while it is in LDS
// The kernel is bound by VALU issue and needs its four workgroups per CU (278 / 300 / 340 us at 4 / 3 / 2). Round 2 took it
// from 278 to 238 us on issue slots alone (tools/stem_down_stamps.py timeline before: loads + image patch 12.7k, stem conv +
// SiLU 12.4k, second conv 5.0k, its SiLU + tile 2.5k, stores 1.1k, 1x1 stage 2.4k of 39k cycles per workgroup): the stem
// conv's operand as one aligned 8-byte read per kernel row instead of eight 2-byte gathers + selects + packs (254 us), the
// second conv's 36 fragment addresses as base registers + immediates (column-keyed swizzle, 248 us), every prologue load
// unconditional — 18 stem weights behind a `g < 3 ?` had become six branch regions, each its own memory round trip (238 us).
// What is left is 120 SiLUs per lane (8 per 16-pixel group of phase B are 36 of its ~76 VALU instructions + 16
// transcendentals). A persistent form that requests the next tile's patch under phase B was measured SLOWER (340 vs 282 us;
// tools/experiments/stem_down_persistent.patch).
// Numerics: the same roundings as the two separate kernels (image and stem output rounded to bf16, fp32 accumulation).
#include "yolo_internal.h"
#include <cstdlib>

namespace adayolo {
namespace sd {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
__device__ __forceinline__ float silu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}

constexpr int TY = 8, TX = 16;                       // output tile of the second conv
constexpr int SH = 2 * TY + 1, SW = 2 * TX + 1;      // stem pixels under it: 17 x 33
constexpr int IH = SH + 2, IW = SW + 2;              // image pixels under those: 19 x 35
constexpr int NSP = SH * SW;                         // 561
// LDS: the stem patch (32 ch bf16 per stem pixel; the 128 px x 128 B output tile later overlays it) and the image patch as
// [y][x][4] bf16 (R, G, B, 0: one aligned 8-byte read = the three channels of a pixel = one kernel column of the stem's
// MFMA operand). The image patch ALIASES the stem pixels 512 .. 560: those are written by the last round of phase B
// (groups 32 .. 35, behind a barrier), which reads image rows 15 .. 18 only, and their 3136 bytes end inside image row 11,
// dead since group 24. 37.2 KB per workgroup: four per CU (the kernel needs them: 278 / 300 / 340 us at 4 / 3 / 2).
constexpr int kPatchBytes = NSP * 64;
constexpr int kImgOff = 512 * 64;                    // byte offset of the image patch = stem pixel 512
constexpr int kImgBytes = IH * IW * 8;               // 5320
constexpr int kSmem = kImgOff + kImgBytes + 8;
static_assert(kImgOff + kImgBytes >= kPatchBytes, "the image patch must cover the tail of the stem patch it aliases");
static_assert(kPatchBytes - kImgOff <= 11 * IW * 8 + 7 * 8 && (35 * 16 + 15) / SW >= 15, "aliased bytes must be dead image rows");

#ifdef ADAYOLO_MEASURE
__device__ unsigned long long g_sd_stamp[4096 * 12];      // per-workgroup phase stamps (measurement build, tools/stem_down_stamps.py)
#define SD_STAMP(k) do { if (threadIdx.x == 0) { const int wg_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x; \
        if (wg_ % 3 == 0 && wg_ / 3 < 4096) g_sd_stamp[(wg_ / 3) * 12 + (k)] = __builtin_readcyclecounter(); } } while (0)
#else
#define SD_STAMP(k) do { } while (0)
#endif

__global__ __launch_bounds__(256, 4) void k_stem_down(const float* __restrict__ img, const float* __restrict__ w0,
                                                   const float* __restrict__ b0, const unsigned short* __restrict__ w1,
                                                   const float* __restrict__ b1, unsigned short* __restrict__ out,
                                                   int out_cs, int H, int W, int Hp, int pad_top, float pad_value,
                                                   const unsigned short* __restrict__ w2, const float* __restrict__ b2,
                                                   unsigned short* __restrict__ out2, int out2_cs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* patch = smem;
    unsigned char* img4 = smem + kImgOff;                  // [IH][IW] x (R, G, B, 0) bf16
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, ox0 = blockIdx.x * TX, oy0 = blockIdx.y * TY;
    const int Ho = Hp >> 1, Wo = W >> 1;
    const int fy0 = 2 * oy0 - 1, fx0 = 2 * ox0 - 1;              // frame coordinates of stem-patch pixel (0, 0)
    SD_STAMP(0);

    // ---- second-conv weights of this wave's 32 channels: 18 k-steps x 16 B per lane, issued first (L2 hits)
    const int chf = wave & 1, pxh = wave >> 1;
    bf16x8 wfr[18];
    {
        const unsigned short* wr = w1 + (long)(32 * chf + (lane & 31)) * 288 + (lane >> 5) * 8;
#pragma unroll
        for (int j = 0; j < 18; ++j) wfr[j] = *reinterpret_cast<const bf16x8*>(wr + (j >> 1) * 32 + (j & 1) * 16);
    }

    // ---- A: image patch (frame rows fy0-1 .. , cols fx0-1 ..)
    const long plane = (long)H * W;
    const float* src = img + (long)b * 3 * plane;
    // a thread takes whole pixels (index math once per pixel, three plane loads each): all 9 loads of a thread are issued
    // before the first LDS write — clamped (always legal) addresses, pinned by an empty asm, then the letterbox value /
    // zero padding by select (a select straight on the load is compiled into a branch around it and serialises the
    // memory round trips) — and a pixel goes to LDS as one 8-byte (R, G, B, 0) write
    {
        constexpr int NPX = IH * IW, NI = (NPX + 255) / 256;
        float v[NI][3], padc[NI];
        bool inimg[NI];
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int i = min(tid + 256 * it, NPX - 1);
            const int ly = i / IW, lx = i - ly * IW;
            const int gy = fy0 - 1 + ly, gx = fx0 - 1 + lx;
            const int sy = gy - pad_top;
            const bool inframe = gy >= 0 && gy < Hp && gx >= 0 && gx < W;
            inimg[it] = inframe && sy >= 0 && sy < H;
            padc[it] = (inframe && !inimg[it]) ? pad_value : 0.0f;   // zero outside the letterboxed frame (the conv's padding)
            const float* q = src + (long)min(max(sy, 0), H - 1) * W + min(max(gx, 0), W - 1);
#pragma unroll
            for (int c = 0; c < 3; ++c) v[it][c] = q[c * plane];
        }
#pragma unroll
        for (int it = 0; it < NI; ++it)
#pragma unroll
            for (int c = 0; c < 3; ++c) asm volatile("" : "+v"(v[it][c]));      // the loads above stay unconditional
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const float r = inimg[it] ? v[it][0] : padc[it], gg = inimg[it] ? v[it][1] : padc[it], bb = inimg[it] ? v[it][2] : padc[it];
            if (tid + 256 * it < NPX)
                *reinterpret_cast<u32x2*>(img4 + (tid + 256 * it) * 8) = u32x2{pack2(r, gg), pack2(bb, 0.0f)};
        }
    }

    // ---- stem weights: per kernel row kh and channel half t one v_mfma_f32_16x16x16_bf16 operand (rows permuted: row
    //      4g+i of fragment t = channel 8g + 4t + i); lane (row p, k-group g) holds k = 4g .. 4g+3 = kernel column kw = g,
    //      channels (R, G, B, pad): zero for g = 3 and for the pad channel
    const int g = lane >> 4, p = lane & 15;
    s16x4 wfk[3][2];
    {
        float h[3][2][3];                                  // all 18 loads unconditional and in flight together (see phase A)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int ch = 8 * (p >> 2) + 4 * t + (p & 3);
#pragma unroll
                for (int c = 0; c < 3; ++c) h[kh][t][c] = w0[ch * 27 + (kh * 3 + min(g, 2)) * 3 + c];
            }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    asm volatile("" : "+v"(h[kh][t][c]));
                    h[kh][t][c] = g < 3 ? h[kh][t][c] : 0.0f;
                }
                const u32x2 pk = {pack2(h[kh][t][0], h[kh][t][1]), pack2(h[kh][t][2], 0.0f)};
                wfk[kh][t] = __builtin_bit_cast(s16x4, pk);
            }
    }
    float bv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bv[i] = b0[8 * g + i];
    SD_STAMP(1);
    __syncthreads();
    SD_STAMP(2);

    // ---- B: stem conv on the patch, 16 pixels per step: per kernel row one aligned 8-byte read (pixel column kw = g; the
    //      k-group g = 3 meets zero weights and re-reads column 2 so that it never sees foreign bits) and one MFMA per
    //      channel half. The last round (groups 32 .. 35) writes the stem pixels the image patch aliases: barrier first.
    const int gk = min(g, 2);
    for (int grp = wave; grp < (NSP + 15) / 16; grp += 4) {
        if (grp >= 32) __syncthreads();                      // (uniform: every wave's ninth round)
        const int P = grp * 16 + p;
        const int Pc = P < NSP ? P : NSP - 1;
        const int sy = Pc / SW, sx = Pc - sy * SW;
        const unsigned char* t0 = img4 + (sy * IW + sx + gk) * 8;
        s16x4 af[3];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) af[kh] = *reinterpret_cast<const s16x4*>(t0 + kh * IW * 8);
        f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            d0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wfk[kh][0], af[kh], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wfk[kh][1], af[kh], d1, 0, 0, 0);
        }
        const int fy = fy0 + sy, fx = fx0 + sx;
        const bool inside = fy >= 0 && fy < Hp && fx >= 0 && fx < W;       // else: the second conv's zero padding
        u32x4 o = {0u, 0u, 0u, 0u};
        if (inside) {
            unsigned o0, o1, o2, o3;
            bias_act_pack4<true>(d0[0], d0[1], d0[2], d0[3], float4{bv[0], bv[1], bv[2], bv[3]}, o0, o1);
            bias_act_pack4<true>(d1[0], d1[1], d1[2], d1[3], float4{bv[4], bv[5], bv[6], bv[7]}, o2, o3);
            o = u32x4{o0, o1, o2, o3};
        }
        if (P < NSP) *reinterpret_cast<u32x4*>(patch + P * 64 + ((g ^ ((sx >> 1) & 3)) << 4)) = o;   // chunks keyed by the pixel's COLUMN
    }
    SD_STAMP(3);
    __syncthreads();
    SD_STAMP(4);

    // ---- C: second conv from the patch. Wave: channel fragment chf (32 ch), pixel half pxh (4 output rows x 16)
    f32x16 acc[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][e] = 0.0f;
    const int fr = lane & 31, fq = lane >> 5;
    // The 16-byte chunks of a stem pixel are XOR-keyed by (its column >> 1) & 3: a tap's column is 2 ox + kw, so the key is
    // ox & 3 for kw = 0, 1 and (ox + 1) & 3 for kw = 2 whatever the row — the 36 fragment addresses of a lane are eight
    // base registers plus immediates (keyed by the linear pixel index they cost ~5 VALU instructions each, in a kernel
    // that is bound by VALU issue).
    const unsigned char* fbase[2][2][2];                   // [pixel mi][kw == 2][k half]
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int oyl = 4 * pxh + 2 * mi + (fr >> 4), oxl = fr & 15;
        const unsigned char* pb = patch + ((2 * oyl) * SW + 2 * oxl) * 64;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int half = 0; half < 2; ++half) fbase[mi][k2][half] = pb + (((2 * half + fq) ^ ((oxl + k2) & 3)) << 4);
    }
#pragma unroll
    for (int j = 0; j < 18; ++j) {
        const int tap = j >> 1, half = j & 1, kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(fbase[mi][kw == 2][half] + (kh * SW + kw) * 64);
            acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfr[j], af, acc[mi], 0, 0, 0);
        }
    }
    SD_STAMP(5);
    __syncthreads();                                       // every wave is done with the patch: overlay the output tile
    SD_STAMP(6);

    // ---- D: bias + SiLU + bf16 -> LDS [128 px][64 ch] (16-byte chunks swizzled by the pixel) -> 128-byte rows
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int q = (4 * pxh + 2 * mi + (fr >> 4)) * 16 + (fr & 15);
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int cl = 32 * chf + 8 * qd + 4 * fq;                     // 4 consecutive channels
            const float4 b4 = *reinterpret_cast<const float4*>(b1 + cl);
            unsigned lo, hi;
            bias_act_pack4<true>(acc[mi][4 * qd], acc[mi][4 * qd + 1], acc[mi][4 * qd + 2], acc[mi][4 * qd + 3], b4, lo, hi);
            *reinterpret_cast<u32x2*>(patch + q * 128 + ((((cl >> 3)) ^ (q & 7)) << 4) + (cl & 4) * 2) = u32x2{lo, hi};
        }
    }
    SD_STAMP(7);
    __syncthreads();
    SD_STAMP(8);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int idx = it * 256 + tid, q = idx >> 3, chunk = idx & 7;
        const int oy = oy0 + (q >> 4), ox = ox0 + (q & 15);
        if (oy < Ho && ox < Wo) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(patch + q * 128 + ((chunk ^ (q & 7)) << 4));
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(out + (((long)b * Ho + oy) * Wo + ox) * out_cs + chunk * 8));
        }
    }
    SD_STAMP(9);
    if (!w2) return;

    // ---- E (optional): the 1x1 conv that follows in yolov3.yaml (Bottleneck.cv1: 64 -> 32) + SiLU, straight from the
    //      output tile in LDS: wave w takes pixels 32w .. 32w+31, K = 64 = 4 MFMA steps; its result goes through the
    //      8 KB behind the tile and leaves as 64-byte pixel rows.
    f32x16 h;
#pragma unroll
    for (int e = 0; e < 16; ++e) h[e] = 0.0f;
    {
        const int q = 32 * wave + fr;
        const unsigned short* wr = w2 + (long)fr * 64 + fq * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const bf16x8 wv = *reinterpret_cast<const bf16x8*>(wr + kk * 16);
            const bf16x8 av = *reinterpret_cast<const bf16x8*>(patch + q * 128 + (((2 * kk + fq) ^ (q & 7)) << 4));
            h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, av, h, 0, 0, 0);
        }
        unsigned char* t2 = patch + 128 * 128;                             // [128 px][32 ch] bf16, chunks swizzled by the pixel
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int cl = 8 * qd + 4 * fq;
            const float4 b4 = *reinterpret_cast<const float4*>(b2 + cl);
            unsigned lo, hi;
            bias_act_pack4<true>(h[4 * qd], h[4 * qd + 1], h[4 * qd + 2], h[4 * qd + 3], b4, lo, hi);
            *reinterpret_cast<u32x2*>(t2 + q * 64 + (((cl >> 3) ^ (q & 3)) << 4) + (cl & 4) * 2) = u32x2{lo, hi};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // the wave reads back only what it wrote
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int q2 = 32 * wave + it * 16 + (lane >> 2), chunk = lane & 3;
            const int oy = oy0 + (q2 >> 4), ox = ox0 + (q2 & 15);
            if (oy < Ho && ox < Wo) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(t2 + q2 * 64 + ((chunk ^ (q2 & 3)) << 4));
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(out2 + (((long)b * Ho + oy) * Wo + ox) * out2_cs + chunk * 8));
            }
        }
    }
    SD_STAMP(10);
}

}  // namespace sd

hipError_t launch_stem_down(const float* img, const float* w0, const float* b0, const void* w1, const float* b1, void* out,
                            int out_cs, int B, int H, int W, int Hp, int pad_top, float pad_value, const void* w2,
                            const float* b2, void* out2, int out2_cs, hipStream_t s) {
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sd::k_stem_down),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, sd::kSmem);
        if (e != hipSuccess) return e;
        configured = true;
    }
    const int Ho = Hp / 2, Wo = W / 2;
    dim3 grid((Wo + sd::TX - 1) / sd::TX, (Ho + sd::TY - 1) / sd::TY, B);
    int smem_bytes = sd::kSmem;
#ifdef ADAYOLO_MEASURE
    static const int extra = getenv("ADAYOLO_SD_EXTRA_SMEM") ? atoi(getenv("ADAYOLO_SD_EXTRA_SMEM")) : 0;   // > 1 KB: three workgroups per CU
    smem_bytes += extra;
    if (extra) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sd::k_stem_down), hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
#endif
    hipLaunchKernelGGL(sd::k_stem_down, grid, dim3(256), smem_bytes, s, img, w0, b0, static_cast<const unsigned short*>(w1), b1,
                       static_cast<unsigned short*>(out), out_cs, H, W, Hp, pad_top, pad_value,
                       static_cast<const unsigned short*>(w2), b2, static_cast<unsigned short*>(out2), out2_cs);
    return hipGetLastError();
}

#ifdef ADAYOLO_MEASURE
extern "C" int adayolo_debug_stem_down(unsigned long long* dst, int n) {       // measurement helper, not part of the ABI
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(sd::g_sd_stamp), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif

}  // namespace adayolo
