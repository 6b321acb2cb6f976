// Detector-side kernels that are not the generic conv: the fused ISP->detector stem, nearest 2x
// up-sampling into a concat slice, and the Detect-head decode. gfx950 only.
#include "yolo_internal.h"

namespace adayolo {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ unsigned short f2bf(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
typedef __attribute__((ext_vector_type(2))) __bf16 hw_bf16x2;
typedef __attribute__((ext_vector_type(2))) float hw_f32x2;
// round-to-nearest-even pair conversion on the hardware unit (v_cvt_pk_bf16_f32) instead of ~8 integer VALU ops
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(hw_f32x2{lo, hi}, hw_bf16x2));
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ float silu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}

// ---------------------------------------------------------------------------------------------------
// Stem: letterbox + Conv(3->32, k3 s1 p1) + SiLU, planar fp32 in, NHWC bf16 out.
// K = 27 taps*channels padded to 32 = ONE v_mfma_f32_16x16x32_bf16 step. Weights (the MFMA A operand) sit
// in registers for the whole kernel; the activation fragment of 16 neighbouring pixels is gathered from
// an fp32 LDS tile. The MFMA row -> channel map is permuted (row 4g+i of tile t = channel 8g+4t+i) so that
// each lane ends up with 8 CONSECUTIVE channels of its pixel and stores them as one 16-byte vector.
// ---------------------------------------------------------------------------------------------------
constexpr int S_TW = 64, S_TH = 16, S_LW = S_TW + 2, S_LH = S_TH + 2;

__global__ __launch_bounds__(256) void k_stem(const float* __restrict__ img, const float* __restrict__ w,
                                              const float* __restrict__ bias, unsigned short* __restrict__ out,
                                              int out_cs, int H, int W, int Hp, int pad_top, float pad_value, int act,
                                              unsigned short* __restrict__ pre, int pre_cs) {
    __shared__ float tile[3 * S_LH * S_LW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, x0 = blockIdx.x * S_TW, y0 = blockIdx.y * S_TH;
    const long plane = (long)H * W;
    const float* src = img + (long)b * 3 * plane;
    for (int i = tid; i < 3 * S_LH * S_LW; i += 256) {
        const int c = i / (S_LH * S_LW), r = i - c * (S_LH * S_LW);
        const int ly = r / S_LW, lx = r - ly * S_LW;
        const int gy = y0 - 1 + ly, gx = x0 - 1 + lx;
        float v = 0.0f;                                   // conv zero padding outside the letterboxed frame
        if (gy >= 0 && gy < Hp && gx >= 0 && gx < W) {
            const int sy = gy - pad_top;
            v = (sy >= 0 && sy < H) ? src[c * plane + (long)sy * W + gx] : pad_value;
        }
        tile[i] = v;
    }
    const int g = lane >> 4, p = lane & 15;
    // weight fragments: tile t, MFMA row j = lane&15 -> channel 8*(j>>2) + 4t + (j&3); k = 8g + e
    bf16x8 wf[2];
    int off[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * g + e;
        const int tap = k / 3, c = k - tap * 3, kh = tap / 3, kw = tap - kh * 3;
        off[e] = (k < 27) ? (c * S_LH + kh) * S_LW + kw : 0;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ch = 8 * (p >> 2) + 4 * t + (p & 3);
        unsigned short h[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 8 * g + e;
            h[e] = (k < 27) ? f2bf(w[ch * 27 + k]) : (unsigned short)0;
        }
        u32x4 pk = {(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16),
                    (unsigned)h[4] | ((unsigned)h[5] << 16), (unsigned)h[6] | ((unsigned)h[7] << 16)};
        wf[t] = __builtin_bit_cast(bf16x8, pk);
    }
    float bv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bv[i] = bias[8 * g + i];
    __syncthreads();

    // each wave: 4 rows x 64 cols = 16 groups of 16 pixels
    for (int grp = 0; grp < 16; ++grp) {
        const int ly = wave * 4 + (grp >> 2), lx = (grp & 3) * 16 + p;
        const int gy = y0 + ly, gx = x0 + lx;
        const float* t0 = tile + ly * S_LW + lx;
        float a[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = t0[off[e]];
        u32x4 pk = {pack2(a[0], a[1]), pack2(a[2], a[3]), pack2(a[4], a[5]), pack2(a[6], a[7])};
        const bf16x8 af = __builtin_bit_cast(bf16x8, pk);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0], af, z, 0, 0, 0);
        const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1], af, z, 0, 0, 0);
        if (gy < Hp && gx < W) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = d0[i] + bv[i]; v[4 + i] = d1[i] + bv[4 + i];
                if (act && !pre) { v[i] = silu(v[i]); v[4 + i] = silu(v[4 + i]); }
            }
            u32x4 o = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
            if (pre) {      // training forward: keep the bf16 pre-activation, activate THAT value (adayolo_silu_fwd's arithmetic)
                *reinterpret_cast<u32x4*>(pre + (((long)b * Hp + gy) * W + gx) * pre_cs + 8 * g) = o;
                if (act) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = silu_bf16x2(o[j]);
                }
            }
            *reinterpret_cast<u32x4*>(out + (((long)b * Hp + gy) * W + gx) * out_cs + 8 * g) = o;
        }
    }
}

hipError_t launch_stem(const float* img, const float* w, const float* bias, void* out, int out_cs, int B, int H,
                       int W, int Hp, int pad_top, float pad_value, int act, hipStream_t s, void* pre, int pre_cs) {
    dim3 grid((W + S_TW - 1) / S_TW, (Hp + S_TH - 1) / S_TH, B);
    hipLaunchKernelGGL(k_stem, grid, dim3(256), 0, s, img, w, bias, static_cast<unsigned short*>(out), out_cs, H, W,
                       Hp, pad_top, pad_value, act, static_cast<unsigned short*>(pre), pre_cs);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Letterbox + layout change only: planar fp32 [B,3,H,W] -> NHWC bf16 [B,Hp,W,8] (channels 3..7 zero), for detectors
// whose first conv is not the 3->32 stem above (width-scaled checkpoints): the generic conv then runs with Cin = 8
// on zero-padded weights. One lane = one pixel = one 16-byte store.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_letterbox_pack(const float* __restrict__ img, unsigned short* __restrict__ out,
                                                        int out_cs, int H, int W, int Hp, int pad_top, float pad_value) {
    const int b = blockIdx.z, gy = blockIdx.y, gx = blockIdx.x * 256 + threadIdx.x;
    if (gx >= W) return;
    const long plane = (long)H * W;
    const int sy = gy - pad_top;
    float r = pad_value, g = pad_value, bl = pad_value;
    if (sy >= 0 && sy < H) {
        const float* src = img + (long)b * 3 * plane + (long)sy * W + gx;
        r = src[0]; g = src[plane]; bl = src[2 * plane];
    }
    const u32x4 o = {pack2(r, g), pack2(bl, 0.0f), 0u, 0u};
    *reinterpret_cast<u32x4*>(out + (((long)b * Hp + gy) * W + gx) * out_cs) = o;
}

hipError_t launch_letterbox_pack(const float* img, void* out, int out_cs, int B, int H, int W, int Hp, int pad_top,
                                 float pad_value, hipStream_t s) {
    hipLaunchKernelGGL(k_letterbox_pack, dim3((W + 255) / 256, Hp, B), dim3(256), 0, s, img,
                       static_cast<unsigned short*>(out), out_cs, H, W, Hp, pad_top, pad_value);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Nearest 2x up-sampling into a channel slice of the concat buffer (nn.Upsample + Concat).
// One lane moves one 16-byte channel chunk of an input pixel to its 4 output pixels.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_upsample2x(const unsigned short* __restrict__ in, int in_cs,
                                                    unsigned short* __restrict__ out, int out_cs, int B, int H, int W,
                                                    int C) {
    const int cpp = C / 8;
    const long total = (long)B * H * W * cpp;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % cpp) * 8;
        const long pix = i / cpp;
        const int x = (int)(pix % W);
        const long by = pix / W;             // b*H + y
        const int y = (int)(by % H);
        const long b = by / H;
        const u32x4 v = *reinterpret_cast<const u32x4*>(in + pix * in_cs + ch);
        unsigned short* o = out + ((b * 2 * H + 2 * y) * (2L * W) + 2 * x) * out_cs + ch;
        *reinterpret_cast<u32x4*>(o) = v;
        *reinterpret_cast<u32x4*>(o + out_cs) = v;
        *reinterpret_cast<u32x4*>(o + 2L * W * out_cs) = v;
        *reinterpret_cast<u32x4*>(o + (2L * W + 1) * out_cs) = v;
    }
}

hipError_t launch_upsample2x(const void* in, int in_cs, void* out, int out_cs, int B, int H, int W, int C,
                             hipStream_t s) {
    const long total = (long)B * H * W * (C / 8);
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_upsample2x, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const unsigned short*>(in),
                       in_cs, static_cast<unsigned short*>(out), out_cs, B, H, W, C);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Detect decode (eval branch of Detect.forward, yolov3/models/yolo.py:56-76).
// ---------------------------------------------------------------------------------------------------
// grid: x over the (cell, channel) pairs of one anchor map, y = anchor, z = image. Consecutive lanes walk the 85
// channels of a cell (contiguous bf16 reads, contiguous fp32 writes); one constant division per element.
__global__ __launch_bounds__(256) void k_detect_decode(const unsigned short* __restrict__ raw, int raw_cs,
                                                       float* __restrict__ pred, int pred_rows, int row_offset,
                                                       const float* __restrict__ anchors_px, float det_stride, int ny,
                                                       int nx, int na, int no) {
    const int an = blockIdx.y;
    const long b = blockIdx.z;
    const int per_map = ny * nx * no;
    const float aw = anchors_px[2 * an], ah = anchors_px[2 * an + 1];
    const unsigned short* src = raw + b * ny * nx * raw_cs + an * no;
    float* dst = pred + (b * pred_rows + row_offset + (long)an * ny * nx) * no;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < per_map; i += gridDim.x * 256) {
        const int cell = i / no, j = i - cell * no;
        const float t = bf2f(src[(long)cell * raw_cs + j]);
        const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * t));
        float v = s;
        if (j < 4) {
            const int y = cell / nx, x = cell - y * nx;
            const float s2 = s * 2.0f;
            v = (j == 0) ? (s2 + ((float)x - 0.5f)) * det_stride
              : (j == 1) ? (s2 + ((float)y - 0.5f)) * det_stride
              : (j == 2) ? s2 * s2 * aw : s2 * s2 * ah;
        }
        dst[i] = v;
    }
}

// Tiled form (default when the layout allows it). The element-per-lane kernel above spends ~40 of its ~70
// instructions per element on two integer divisions and stores 4 bytes per lane: 90 us for the 92x160 map. Here a
// workgroup takes 64 cells of one image. Their raw rows (all anchors, raw_cs bf16 each) are one contiguous block: read
// with 16-byte loads and scattered into LDS in OUTPUT order ([anchor][cell][channel], bf16), so that each anchor's
// 64 x no outputs — one contiguous run of the prediction tensor — are produced four at a time from one ds_read_b64
// with no index arithmetic: sigmoid, float4 store. The four box channels of every cell are computed first (one thread
// per cell and anchor, fp32 into LDS) and patched into the 7-in-85 runs that touch them.
constexpr int DC = 32;                          // cells per workgroup (18 KB of LDS: 8 workgroups per CU)
__global__ __launch_bounds__(256) void k_detect_decode_tiled(const unsigned short* __restrict__ raw, int raw_cs,
                                                             float* __restrict__ pred, int pred_rows, int row_offset,
                                                             const float* __restrict__ anchors_px, float det_stride,
                                                             int ny, int nx, int na, int no) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    unsigned short* l16 = reinterpret_cast<unsigned short*>(dsm);          // [na][DC][no]
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    const int ncell = ny * nx;
    const int cell0 = blockIdx.x * DC;
    const int nc = min(DC, ncell - cell0);
    const long b = blockIdx.y;
    const int used = na * no;                                  // bf16 channels per cell that matter
    const int cpr = (used + 7) / 8;                            // 16-byte chunks per row
    const float inv_cpr = 1.0f / (float)cpr, inv_no = 1.0f / (float)no, inv_nx = 1.0f / (float)nx;
    const unsigned char* src = reinterpret_cast<const unsigned char*>(raw + (b * ncell + cell0) * raw_cs);
    // four 16-byte loads in flight per thread before the first one is used (the raw maps were streamed out by the
    // head convs: every load is an HBM round trip, and one per loop trip made this phase 8 serial round trips)
    for (int q0 = threadIdx.x; q0 < nc * cpr; q0 += 4 * 256) {
        uint4 vv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = min(q0 + i * 256, nc * cpr - 1);
            const int r = (int)(((float)q + 0.5f) * inv_cpr), c0 = (q - r * cpr) * 8;    // q / cpr by reciprocal (q < 2^16)
            vv[i] = *reinterpret_cast<const uint4*>(src + (long)r * raw_cs * 2 + c0 * 2);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = q0 + i * 256;
            if (q >= nc * cpr) break;
            const int r = (int)(((float)q + 0.5f) * inv_cpr), c0 = (q - r * cpr) * 8;
            const unsigned w[4] = {vv[i].x, vv[i].y, vv[i].z, vv[i].w};
            int an = (int)(((float)c0 + 0.5f) * inv_no);
            int j = c0 - an * no;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (c0 + u < used) l16[(an * DC + r) * no + j] = (unsigned short)(w[u >> 1] >> ((u & 1) * 16));
                ++j;
                const int wr = j == no;
                j = wr ? 0 : j; an += wr;
            }
        }
    }
    __syncthreads();
    // box channels (x, y, w, h) of every cell and anchor -> fp32 in LDS, one thread each; the main pass patches them in
    float* boxf = reinterpret_cast<float*>(dsm + (size_t)na * DC * no * 2);        // [na][DC][4]
    for (int idx = threadIdx.x; idx < na * nc; idx += 256) {
        const int an = (int)(((float)idx + 0.5f) / (float)nc), cl = idx - an * nc;
        const int cell = cell0 + cl;
        const int y = (int)(((float)cell + 0.5f) * inv_nx), x = cell - y * nx;
        const unsigned short* lp = l16 + (an * DC + cl) * no;
        float sg[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) sg[u] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * bf2f(lp[u])));
        float* bo = boxf + (an * DC + cl) * 4;
        bo[0] = (sg[0] * 2.0f + ((float)x - 0.5f)) * det_stride;
        bo[1] = (sg[1] * 2.0f + ((float)y - 0.5f)) * det_stride;
        bo[2] = (sg[2] * 2.0f) * (sg[2] * 2.0f) * anchors_px[2 * an];
        bo[3] = (sg[3] * 2.0f) * (sg[3] * 2.0f) * anchors_px[2 * an + 1];
    }
    __syncthreads();
    const int total = nc * no;                                 // a multiple of 4 (launcher: DC * no and ny * nx * no are)
    for (int an = 0; an < na; ++an) {
        float* dst = pred + (b * pred_rows + row_offset + (long)an * ncell + cell0) * no;     // 16-byte aligned (launcher)
        const unsigned char* la = dsm + (long)an * DC * no * 2;
        for (int e4 = threadIdx.x; 4 * e4 < total; e4 += 256) {
            const uint2 p = *reinterpret_cast<const uint2*>(la + 8 * e4);
            const float t[4] = {__uint_as_float(p.x << 16), __uint_as_float(p.x & 0xFFFF0000u),
                                __uint_as_float(p.y << 16), __uint_as_float(p.y & 0xFFFF0000u)};
            f32x4v o;
#pragma unroll
            for (int u = 0; u < 4; ++u) o[u] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * t[u]));
            const int cl = (int)(((float)(4 * e4) + 0.5f) * inv_no), j0 = 4 * e4 - cl * no;
            if (j0 < 4 || j0 + 3 >= no) {                      // 7 of every `no` lanes: the run touches box channels
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int wrp = j0 + u >= no;
                    const int j = j0 + u - (wrp ? no : 0);
                    if (j < 4) o[u] = boxf[(an * DC + cl + wrp) * 4 + j];
                }
            }
            __builtin_nontemporal_store(o, reinterpret_cast<f32x4v*>(dst + 4 * e4));
        }
    }
}

hipError_t launch_detect_decode(const void* raw, int raw_cs, float* pred, int pred_rows, int row_offset,
                                const float* anchors_px, float det_stride, int B, int ny, int nx, int na, int no,
                                hipStream_t s) {
    {
        // tiled form: raw rows 16-byte aligned, every anchor's output run 16-byte aligned (no is odd for COCO, so the
        // row counts in front of it have to be multiples of 4), tile within the static LDS limit
        const int cpr = (na * no + 7) / 8;
        const size_t lds = (size_t)na * DC * no * 2 + (size_t)na * DC * 16;     // bf16 tile + fp32 box values
        const bool rows_ok = raw_cs % 8 == 0 && cpr * 8 <= raw_cs && (reinterpret_cast<uintptr_t>(raw) & 15) == 0;
        const bool out_ok = (reinterpret_cast<uintptr_t>(pred) & 15) == 0 && ((long)pred_rows * no) % 4 == 0 &&
                            ((long)row_offset * no) % 4 == 0 && ((long)ny * nx * no) % 4 == 0 && (DC * no) % 4 == 0;
        if (rows_ok && out_ok && lds <= 64 * 1024 && DC * cpr < 65536 && na * DC < 65536) {
            hipLaunchKernelGGL(k_detect_decode_tiled, dim3((ny * nx + DC - 1) / DC, B), dim3(256), lds, s,
                               static_cast<const unsigned short*>(raw), raw_cs, pred, pred_rows, row_offset, anchors_px,
                               det_stride, ny, nx, na, no);
            return hipGetLastError();
        }
    }
    const int per_map = ny * nx * no;
    int bx = (per_map + 255) / 256;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(k_detect_decode, dim3(bx, na, B), dim3(256), 0, s, static_cast<const unsigned short*>(raw),
                       raw_cs, pred, pred_rows, row_offset, anchors_px, det_stride, ny, nx, na, no);
    return hipGetLastError();
}

}  // namespace adayolo
