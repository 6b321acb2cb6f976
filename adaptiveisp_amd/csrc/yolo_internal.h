// Shared between the translation units of libadayolo.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/adayolo.h"

namespace adayolo {

struct ConvArgs {
    const unsigned short* in; int in_cs;
    const unsigned short* w; const float* bias;
    const unsigned short* res; int res_cs;
    unsigned short* out; int out_cs;
    int B, H, W, Cin, Cout, Ho, Wo, ks, stride, pad, act;
    int M;            // B*Ho*Wo
    int mtiles, ntiles;
    // division by Ho*Wo and by Wo as multiply-high + shift (exact for 0 <= n < 2^31; sh < 0: divisor is 1)
    unsigned magic_hw, magic_w;
    int sh_hw, sh_w;
    // fused second layer (yolo_conv_pp.hip, Cout == 256 only): a 1x1 conv 256 -> 128 + bias + SiLU applied to this conv's
    // OUTPUT tile while it is in LDS (Bottleneck.cv1 of the next block); null = not fused
    const unsigned short* w2; const float* bias2;
    unsigned short* out2; int out2_cs;
    // training forward (yolo_conv_dma2.hip / yolo_conv_pp128.hip): when set, the bf16 pre-activation (conv + bias) is stored
    // here and the activation is applied to that ROUNDED value — bit for bit what adayolo_silu_fwd makes of it
    unsigned short* pre; int pre_cs;
    // split-K (yolo_conv_pp128.hip, variants 100 + S): S workgroups per output tile, fp32 partial tiles + one ticket per tile
    // in the caller's workspace; 1 = every other kernel
    float* partial = nullptr; int* tickets = nullptr; int ksplit = 1;
    // backward of the frozen detector (variants of the keep set): when gpre is set, `pre` is an INPUT (the layer's saved
    // pre-activation), the conv result (+ residual) is the gradient of that layer's output — stored to `out` only when `out`
    // is not null — and gpre = bf16(that bf16 value * silu'(pre)): adayolo_silu_bwd inside the producing launch
    unsigned short* gpre = nullptr; int gpre_cs = 0;
    // data gradient of a stride-2 conv as a 2x2 stride-1 conv over the (small) output-gradient grid whose 4*C output channels
    // are the four pixel parities of the (large) input-gradient tensor: d2s_c = C > 0 makes the epilogue address out / res /
    // pre / gpre depth-to-space — channel group p of pixel (b, i, j) is pixel (b, 2i + p/2, 2j + p%2) of [B, 2Ho, 2Wo, C]
    int d2s_c = 0;
    // persistent chain (yolo_conv_chain.hip): which tile body runs this layer's items — 0: 256 x 256 (yolo_conv_pp.hip),
    // 1: 256 x 128 (yolo_conv_pp128.hip)
    int chain_tile = 0;
};

// (pixel, channel) of an epilogue element in the tensors it addresses: the identity, or the depth-to-space map above
__device__ __forceinline__ void epilogue_pos(const ConvArgs& a, int m, int n, long& pix, int& nn) {
    pix = m; nn = n;
    if (a.d2s_c) {
        const int b = a.sh_hw < 0 ? m : (int)(__umulhi((unsigned)m, a.magic_hw) >> a.sh_hw);
        const int rem = m - b * (a.Ho * a.Wo);
        const int i = a.sh_w < 0 ? rem : (int)(__umulhi((unsigned)rem, a.magic_w) >> a.sh_w);
        const int j = rem - i * a.Wo;
        const int p = n / a.d2s_c;
        nn = n - p * a.d2s_c;
        pix = ((long)(b * 2 * a.Ho + 2 * i + (p >> 1))) * (2 * a.Wo) + 2 * j + (p & 1);
    }
}

// Epilogue math on channel pairs: packed fp32 (v_pk_add/mul_f32 do two channels per issue slot; the two transcendentals
// stay per element) — the conv epilogues are VALU-bound on exactly this (128 SiLUs per lane in the 256x256 kernel).
typedef float f32x2_pk __attribute__((ext_vector_type(2)));
// eight bf16 values of an epilogue row: silu of each, rounded back to bf16 (the training forward's second output)
__device__ __forceinline__ f32x2_pk silu_pk(f32x2_pk x);
__device__ __forceinline__ unsigned silu_bf16x2(unsigned v) {
    typedef __bf16 bf16x2_pk __attribute__((ext_vector_type(2)));
    const f32x2_pk y = silu_pk(f32x2_pk{__uint_as_float(v << 16), __uint_as_float(v & 0xFFFF0000u)});
    return __builtin_bit_cast(unsigned, __builtin_convertvector(y, bf16x2_pk));
}
// g * silu'(p) on a bf16 pair, rounded back to bf16 — THE formula of the backward (k_silu_bwd and the conv epilogues that
// absorb it must agree bit for bit, hence the explicit fma: nothing is left to contraction).
// silu'(p) = s + p s (1 - s), s = sigmoid(p)
__device__ __forceinline__ float dsilu_f32(float g, float p) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * p));
    return g * __builtin_fmaf(p * s, 1.0f - s, s);
}
__device__ __forceinline__ unsigned dsilu_bf16x2(unsigned g, unsigned p) {
    typedef __bf16 bf16x2_pk __attribute__((ext_vector_type(2)));
    const f32x2_pk y = {dsilu_f32(__uint_as_float(g << 16), __uint_as_float(p << 16)),
                        dsilu_f32(__uint_as_float(g & 0xFFFF0000u), __uint_as_float(p & 0xFFFF0000u))};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(y, bf16x2_pk));
}
__device__ __forceinline__ f32x2_pk silu_pk(f32x2_pk x) {
    const f32x2_pk u = x * -1.44269504088896341f;
    f32x2_pk e = {__builtin_amdgcn_exp2f(u.x), __builtin_amdgcn_exp2f(u.y)};
    e = e + 1.0f;
    const f32x2_pk r = {__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
    return x * r;
}
// four consecutive channels: + bias, SiLU (compile-time), round to bf16 (v_cvt_pk_bf16_f32) -> two packed words
template <bool SILU>
__device__ __forceinline__ void bias_act_pack4(float a0, float a1, float a2, float a3, const float4 b, unsigned& lo, unsigned& hi) {
    typedef __bf16 bf16x2_pk __attribute__((ext_vector_type(2)));
    f32x2_pk x0 = f32x2_pk{a0, a1} + f32x2_pk{b.x, b.y};
    f32x2_pk x1 = f32x2_pk{a2, a3} + f32x2_pk{b.z, b.w};
    if (SILU) { x0 = silu_pk(x0); x1 = silu_pk(x1); }
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(x0, bf16x2_pk));
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(x1, bf16x2_pk));
}

// ---- persistent chain (yolo_conv_pp.hip: k_conv_chain): several consecutive layers of the 256 x 256 kernel's tiles in ONE launch.
// One work item = one tile of one layer; the tables are built on the host (yolo_api.hip: adayolo_conv_chain_prepare).
// Workspace (device, caller-owned; the byte offsets are multiples of 64):
//   [0, 64)            int head (next item to hand out), int err (!= 0: a bounded wait of THIS launch gave up: item + 1), int exit
//                      (workgroups that have left), int sticky err (the last launch that gave up, until the next prepare),
//                      padding — head, err, exit and done[] are ZERO between launches
//   [64, ...)          int done[ndone]: arrival counters, one per (layer, m-tile)
//   [off_layers, ...)  ConvArgs[nlayers]
//   [off_heads, ...)   ChainHead[total]   layer-major: an item only ever waits for items BEFORE it in this order
//   [off_deps, ...)    ChainDeps[total]
// Everything behind the counters is constant during a launch and read through the SCALAR path (uniform indices).
struct ChainHead { int layer, lid, flag, pad; };   // tile `lid` (m-tile * ntiles + n-tile) of `layer`; done[flag] += 1 when it is stored
struct ChainDeps {                                 // done[lo .. lo + n) each >= need before the input window / the residual rows are read
    int in_lo, in_n_need;                          // (n << 16) | need; n = 0: produced before the launch
    int res_lo, res_n_need;
};
struct ChainArgs {
    unsigned char* ws;
    int* host_err;               // pinned HOST word (device-visible): the launch's last workgroup mirrors a give-up code there, so
                                 // that adayolo_conv_chain_poll reads it without touching the device (nullptr: not mirrored)
    int off_layers, off_heads, off_deps, total, ndone;
    int stagger;                 // cycles: workgroup b starts ((b >> 3) & 7) * stagger late (0: all at once). ADAYOLO_CHAIN_STAGGER
};
hipError_t launch_conv_chain(const ChainArgs& c, int grid, hipStream_t s);
hipError_t launch_conv_dma(ConvArgs a, hipStream_t s, int variant);   // LDS-DMA ring (yolo_conv_dma.hip)
hipError_t launch_conv_dma2(ConvArgs a, hipStream_t s, int variant);  // lean-address 32x32 MFMA ring (yolo_conv_dma2.hip)
hipError_t launch_conv_small(ConvArgs a, hipStream_t s, int variant); // 3x3, Cin 32/64, whole K resident (yolo_conv_small.hip)
hipError_t launch_conv_pp(ConvArgs a, hipStream_t s, int variant);    // 256x256 ping-pong wave groups (yolo_conv_pp.hip)
hipError_t launch_conv_pp128(ConvArgs a, hipStream_t s, int variant); // 256x128 ping-pong, 3-deep ring (yolo_conv_pp128.hip)
size_t conv_pp128_splitk_bytes(const ConvArgs& a, int S);               // workspace of the split-K form, 0 = shape not served
hipError_t launch_conv_pp128_splitk(ConvArgs a, hipStream_t s, int S, void* workspace, size_t workspace_bytes);
// one Bottleneck (1x1 256 -> 128 + SiLU, 3x3 128 -> 256 + SiLU, + x) per launch, hidden tensor in LDS (yolo_bneck.hip)
hipError_t launch_bottleneck256(const void* x, int x_cs, const void* w1, const float* b1, const void* w2, const float* b2,
                                void* out, int out_cs, int B, int H, int W, hipStream_t s);
// one Bottleneck of the shallow stages (C = 64 / 128) per launch: weights in registers, x patch by LDS-DMA, h in LDS (yolo_bneck_ws.hip)
hipError_t launch_bottleneck_ws(const void* x, int x_cs, const void* w1, const float* b1, const void* w2, const float* b2,
                                void* out, int out_cs, int B, int H, int W, int C, hipStream_t s);
hipError_t launch_conv_k1(ConvArgs a, hipStream_t s);                 // 1x1, whole K at once: weights in registers (fragment-major copy), activation tile in LDS (yolo_conv_k1.hip)
hipError_t launch_conv_pq(ConvArgs a, hipStream_t s, int variant);    // 256x128, 4 waves, two workgroups per CU (yolo_conv_pq.hip)
hipError_t launch_conv_ws(ConvArgs a, hipStream_t s, int variant);    // 3x3 s1, Cin 32 / 64: weights in registers, patch in LDS, persistent (yolo_conv_ws.hip)
hipError_t launch_stem(const float* img, const float* w, const float* bias, void* out, int out_cs, int B, int H,
                       int W, int Hp, int pad_top, float pad_value, int act, hipStream_t s, void* pre = nullptr, int pre_cs = 0);
hipError_t launch_letterbox_pack(const float* img, void* out, int out_cs, int B, int H, int W, int Hp, int pad_top,
                                 float pad_value, hipStream_t s);
hipError_t launch_stem_down(const float* img, const float* w0, const float* b0, const void* w1, const float* b1, void* out,
                            int out_cs, int B, int H, int W, int Hp, int pad_top, float pad_value, const void* w2,
                            const float* b2, void* out2, int out2_cs, hipStream_t s);
hipError_t launch_upsample2x(const void* in, int in_cs, void* out, int out_cs, int B, int H, int W, int C,
                             hipStream_t s);
hipError_t launch_detect_decode(const void* raw, int raw_cs, float* pred, int pred_rows, int row_offset,
                                const float* anchors_px, float det_stride, int B, int ny, int nx, int na, int no,
                                hipStream_t s);
hipError_t launch_silu_fwd(const void* pre, int pre_cs, const void* res, int res_cs, void* out, int out_cs, long npix,
                           int C, hipStream_t s);
hipError_t launch_silu_bwd(const void* gy, int gy_cs, const void* pre, int pre_cs, void* gp, int gp_cs, void* gres,
                           int gres_cs, int accumulate, long npix, int C, hipStream_t s);
hipError_t launch_zero_insert(const void* in, int in_cs, void* out, int out_cs, int B, int Ho, int Wo, int H, int W, int C,
                              hipStream_t s);
hipError_t launch_upsample_bwd(const void* gy, int gy_cs, void* gx, int gx_cs, int accumulate, int B, int H, int W, int C,
                               hipStream_t s);
hipError_t launch_image_grad(const void* g, int g_cs, float* grad_img, int B, int H, int W, int Hp, int pad_top,
                             hipStream_t s);
hipError_t launch_detloss_fwd(const adayolo_loss_args& a, hipStream_t s);   // yolo_loss.hip
hipError_t launch_detloss_bwd(const adayolo_loss_args& a, hipStream_t s);
hipError_t launch_nms(const float* boxes, int n, float thr, int max_det, unsigned long long* mask_ws, int* keep,
                      int* num_keep, hipStream_t s);

}  // namespace adayolo
