/*
 * adayolo.h — C-ABI of the MI355X-native YOLOv3 reward-model forward (libadayolo.so).
 *
 * Replaces, for the detector half of the hot path, the reference's PyTorch modules
 *   Conv       = SiLU(BN(conv2d(x)))            yolov3/models/common.py:45-59   (BN folded: forward_fuse :58-59,
 *                                                yolov3/utils/torch_utils.py:248-269)
 *   Bottleneck = x + cv2(cv1(x))                 yolov3/models/common.py:110-120
 *   nn.Upsample(2,'nearest') + Concat            yolov3/models/yolov3.yaml:34-36,41-43, common.py:305-312
 *   Detect (eval decode)                         yolov3/models/yolo.py:56-76
 * The reference has no FFI here either; these entry points are what a binding of those modules calls.
 *
 * Conventions: DEVICE pointers, caller-owned. Activations are NHWC bf16 ("pixel-major"): element
 * (b,y,x,c) of a tensor with channel stride `cs` lives at ((b*H + y)*W + x)*cs + c — `cs` may exceed the
 * channel count so that a conv can read or write a channel slice of a wider (concatenated) tensor.
 * Weights are bf16 [Cout][KH][KW][Cin] (BN folded in), bias fp32 [Cout]. Channel counts and strides are
 * multiples of 8 (16-byte vectors). All calls enqueue on `stream`, never allocate or synchronise.
 * Return 0 on success, negative on error (adayolo_strerror).
 */
#ifndef ADAYOLO_H_
#define ADAYOLO_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADAYOLO_ABI_VERSION 9

#define ADAYOLO_ACT_NONE 0
#define ADAYOLO_ACT_SILU 1

#define ADAYOLO_OK       0
#define ADAYOLO_EINVAL  -1
#define ADAYOLO_ESHAPE  -2
#define ADAYOLO_ELAUNCH -3

/*
 * out[b,ho,wo,co] = act( bias[co] + sum_{kh,kw,ci} in[b, ho*stride-pad+kh, wo*stride-pad+kw, ci] * w[co,kh,kw,ci] )
 *                   (+ residual[b,ho,wo,co] if residual != NULL, added AFTER the activation: Bottleneck shortcut)
 * Implicit GEMM on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16), fp32 accumulate, fused epilogue.
 * ksize in {1,3}; pad = ksize/2; stride in {1,2}. Ho = (H + 2*pad - ksize)/stride + 1 (same for Wo).
 */
int adayolo_conv_fwd(const void* in, int in_cstride,
                     const void* weight, const float* bias,
                     const void* residual, int res_cstride,
                     void* out, int out_cstride,
                     int B, int H, int W, int Cin, int Cout,
                     int ksize, int stride, int act, void* stream);

/*
 * A CHAIN of consecutive conv layers in ONE persistent launch (csrc/yolo_conv_pp.hip: k_conv_chain). Each layer is what
 * adayolo_conv_fwd (weight2 == NULL) or adayolo_conv_fused1x1_fwd computes — Conv/Bottleneck of yolov3/models/common.py:45-59,
 * 110-120 — on the 256 px x 256 ch kernel (tile 0: Cin % 64 == 0, Cout % 256 == 0; fused: Cout == 256, Cout2 == 128, weight2
 * fragment-major as below) or the 256 px x 128 ch kernel (tile 1: Cin % 64 == 0, Cout % 128 == 0, not fused). One workgroup per
 * CU draws tiles of ALL the layers from a work counter in layer order; a tile waits, through per-m-tile arrival counters, for exactly the tiles of the producing layer that its input
 * window and its residual rows lie in. The results are bit-identical to launching the layers one by one (same tile code);
 * what is saved is the ramp / tail of every launch, the partly empty last round of every layer and the lock-step of the
 * CUs' memory phases. Rules (ADAYOLO_ESHAPE otherwise): a layer's `in` / `residual` is either EXACTLY the `out` / `out2` of
 * an earlier layer of the chain (same pointer and stride, M equal) or a tensor no layer of the chain writes (complete
 * before the launch); no output overlaps any other tensor of the chain; every output tensor is < 2 GB; n <= 64.
 *   workspace_bytes  size of the device workspace for this chain (0: not served)
 *   prepare          builds the work-item tables in `workspace` (host work + one blocking copy: a SET-UP call, not
 *                    capturable); again whenever the layer list changes
 *   fwd              one forward of the chain on `stream` (ONE kernel: capturable; no allocation, no sync, no table work: a
 *                    checked launch — ADAYOLO_EINVAL unless `workspace` was prepared for exactly this layer list). The counters in
 *                    the workspace are zero between launches: `prepare` zeroes them, every launch's last workgroup leaves them zero
 *   A dependency wait is bounded (1 s of wall time, s_memrealtime). A wait that gives up lets the launch finish — on incomplete
 *   inputs — and records its item: in the workspace (sticky until the next prepare) and in a pinned HOST word.
 *   poll             that host word: 0, or item + 1 of the last wait that gave up. No device call, no sync — the production
 *                    path's check (YoloEngine: before every forward and at every host sync point; a non-zero value raises)
 *   status           the same from the workspace; BLOCKS (hipDeviceSynchronize + copy): tests, and end-of-run checks
 */
typedef struct adayolo_chain_layer {
    const void* in; int in_cstride;
    const void* weight; const float* bias;
    const void* residual; int res_cstride;
    void* out; int out_cstride;
    int B, H, W, Cin, Cout, ksize, stride, act;
    const void* weight2; const float* bias2; void* out2; int out2_cstride; int Cout2;     /* weight2 == NULL: not fused */
    int tile;       /* which kernel's tile runs the layer: 0 = 256 px x 256 ch (variant 50; the only one that fuses), 1 = 256 px x
                       128 ch (variant 60: Cin % 64 == 0, Cout % 128 == 0) — the results are those kernels' bit for bit */
} adayolo_chain_layer;
size_t adayolo_conv_chain_workspace_bytes(const adayolo_chain_layer* layers, int n);
int adayolo_conv_chain_prepare(const adayolo_chain_layer* layers, int n, void* workspace, size_t workspace_bytes);
int adayolo_conv_chain_fwd(const adayolo_chain_layer* layers, int n, void* workspace, size_t workspace_bytes, void* stream);
int adayolo_conv_chain_status(const void* workspace);
int adayolo_conv_chain_poll(const void* workspace);
/* The workspace image `prepare` uploads, written to HOST memory instead (no device needed: how the -m "not gpu" tests check the
 * work-item order and every tile's dependency window). Layout: 64 bytes of counters (head, err, exit), int done[ndone], then at the
 * 64-byte-aligned offsets returned in info = {items, ndone, off_layers, off_items, off_deps, sizeof(layer record)}: the
 * per-layer argument records, items {layer, tile, arrival counter, 0} and dependencies {in_lo, (in_n << 16) | in_need, res_lo,
 * (res_n << 16) | res_need} (indices into done[]). */
int adayolo_conv_chain_tables(const adayolo_chain_layer* layers, int n, void* host_image, size_t bytes, int32_t* info);

/*
 * Conv(Cin -> Cout, k1, s1) + bias + act — what adayolo_conv_fwd computes for ksize 1 (Bottleneck.cv1,
 * yolov3/models/common.py:45-59,110-120) — on the "whole K at once" kernel (csrc/yolo_conv_k1.hip): a workgroup takes 128
 * pixels x 256 output channels, its whole activation tile arrives in LDS in one LDS-DMA burst and each wave's weights stay in
 * registers. For the deep, narrow-map layers (512 -> 256 on 46 x 80 maps at the benchmark's size: one round of tiles on 256
 * CUs) where the ring kernels spend their time in prologue / barrier / epilogue structure. Same arithmetic as the ring kernels
 * (bf16 operands, fp32 accumulation in k order, bias, SiLU, one rounding to bf16).
 * weight_fragments: the bf16 [Cout][Cin] matrix stored FRAGMENT-MAJOR, [Cout/32][Cin/16][2][32][8]: element [c][kk][f][r][j] =
 * w[32 c + r][16 kk + 8 f + j] (a wave's MFMA operand load is 1 KB contiguous; YoloEngine packs it once:
 * view(Cout/32, 32, Cin/16, 2, 8).permute(0, 2, 3, 1, 4)).
 * Shapes: Cin in {256, 512}, Cout % 256 == 0 (ADAYOLO_ESHAPE otherwise — the caller keeps adayolo_conv_fwd).
 */
int adayolo_conv1x1_stream_fwd(const void* in, int in_cstride, const void* weight_fragments, const float* bias,
                               void* out, int out_cstride, int B, int H, int W, int Cin, int Cout, int act, void* stream);

/*
 * Two layers in one launch: Conv(Cin -> 256, k, stride) + bias + act (+ residual) -> `out`, and on that output tile, while
 * it is in LDS, Conv(256 -> 128, k1) + bias2 + SiLU -> `out2`. This is Bottleneck.cv2 of one block followed by
 * Bottleneck.cv1 of the next (yolov3/models/common.py:110-120): the 1x1 conv is HBM-bound on its own (it re-reads what
 * was just written); fused, its input never comes back from memory. `out2` equals what adayolo_conv_fwd would produce
 * from `out` (the second layer reads the bf16-rounded rows of `out`). bias2: fp32 [128]. weight2: the bf16 [128][256] matrix
 * w2 stored FRAGMENT-MAJOR, [4][16][2][32][8]: element [c][kk][f][r][j] = w2[32 c + r][16 kk + 8 f + j] (a wave's MFMA
 * operand load is then 1 KB contiguous; YoloEngine packs it once: view(4,32,16,2,8).permute(0,2,3,1,4)).
 * Shapes: Cout == 256, Cout2 == 128, Cin % 64 == 0 (ADAYOLO_ESHAPE otherwise — the caller runs the two layers separately).
 */
int adayolo_conv_fused1x1_fwd(const void* in, int in_cstride,
                              const void* weight, const float* bias,
                              const void* residual, int res_cstride,
                              void* out, int out_cstride,
                              int B, int H, int W, int Cin, int Cout,
                              int ksize, int stride, int act,
                              const void* weight2, const float* bias2,
                              void* out2, int out2_cstride, int Cout2, void* stream);

/*
 * A whole Bottleneck block of the C = 256 stage in one launch (yolov3/models/common.py:110-120, `x + cv2(cv1(x))` with
 * cv1 = Conv(256 -> 128, k1) + SiLU and cv2 = Conv(128 -> 256, k3, s1) + SiLU, BatchNorm folded into both):
 *     out = x + SiLU(bias2 + W2 (3x3) * SiLU(bias1 + W1 (1x1) * x))
 * The hidden tensor never leaves the CU: a workgroup computes it on the 18 x 18 patch its 16 x 16 output tile needs,
 * bf16-rounded exactly as the stand-alone 1x1 layer would store it, into LDS; the 3x3's activation operand is read from
 * there (only its weights stream), and the shortcut re-reads the x tile the same workgroup fetched a moment earlier.
 * x / out: NHWC bf16 [B,H,W,256] with channel strides; weight1 bf16 [128][256], weight2 bf16 [256][3][3][128], biases fp32.
 * `out` must not overlap `x` (tiles read their neighbours' x while others write). ADAYOLO_ESHAPE for strides that are not
 * multiples of 8 or < 256. Same result as adayolo_conv_fwd (k1) followed by adayolo_conv_fwd (k3, residual = x).
 */
int adayolo_bottleneck256_fwd(const void* x, int x_cstride,
                              const void* weight1, const float* bias1,
                              const void* weight2, const float* bias2,
                              void* out, int out_cstride,
                              int B, int H, int W, void* stream);

/*
 * The same for the SHALLOW stages — C = 128 (hidden 64: the two blocks on 184 x 320 maps at the benchmark's size) and C = 64
 * (hidden 32: the block on 368 x 640 maps), yolov3/models/common.py:110-120, yolov3/models/yolov3.yaml:13-27 —
 *     out = x + SiLU(bias2 + W2 (3x3, pad 1) * SiLU(bias1 + W1 (1x1) * x))
 * on the weights-in-registers kernel (csrc/yolo_bneck_ws.hip): persistent workgroups, the 3x3's weights in registers, the x patch
 * of a tile by LDS-DMA, the hidden tensor only ever in LDS (rounded to bf16 exactly as the stand-alone 1x1 stores it), ALL
 * output channels of a pixel from one workgroup. weight1: bf16 [C/2][C]; weight2: bf16 [C][3][3][C/2]; biases fp32; x and out
 * NHWC bf16 with channel strides (multiples of 8, >= C), x != out, each < 4 GB.
 * ADAYOLO_ESHAPE: C not in {64, 128}, or a tensor beyond 32-bit byte offsets — the caller runs the two layers separately.
 */
int adayolo_bottleneck_ws_fwd(const void* x, int x_cstride, const void* weight1, const float* bias1,
                              const void* weight2, const float* bias2, void* out, int out_cstride,
                              int B, int H, int W, int C, void* stream);


/*
 * Same as adayolo_conv_fwd with an explicit kernel (what YoloEngine.autotune picks per layer; results agree to the
 * bf16 rounding of the output, tests/test_gpu_yolo_variants.py): 0 = library default (= 2); 2 = LDS-DMA ring, 16x16x32
 * MFMA; 5 / 22 / 26 / 27 = lean-address LDS-DMA ring on 32x32x16 MFMA with tiles 128x128 (by shape) / 128x64 /
 * 128x256 / 256x128 px x ch; 40 = whole-K-resident 3x3 for Cin 32 / 64; 50 = 256x256 ping-pong wave groups
 * (Cin % 64 == 0, Cout % 256 == 0); 60 = 256x128 ping-pong (Cin % 64 == 0, Cout % 128 == 0); 80 = 256x128 with four
 * waves and two workgroups per CU (Cin % 32 == 0, Cout % 128 == 0, K >= 96), 85 = the same with a 128-pixel tile; 90 = 3x3 stride 1 with Cin 32 / 64 and SiLU:
 * weights stationary in registers, activation patch in LDS, persistent workgroups (Cout % 64 == 0). A specialised kernel
 * asked for a shape it does not serve runs the default instead; any other number is ADAYOLO_EINVAL.
 */
int adayolo_conv_fwd_variant(const void* in, int in_cstride,
                             const void* weight, const float* bias,
                             const void* residual, int res_cstride,
                             void* out, int out_cstride,
                             int B, int H, int W, int Cin, int Cout,
                             int ksize, int stride, int act, int variant, void* stream);

/*
 * Detector stem fused with the ISP->detector hand-over: reads the ISP output as planar fp32
 * [B,3,H,W] (values in [0,1]), letterboxes it vertically to Hp rows (pad_top rows of `pad_value` above,
 * the rest below: yolov3/utils/augmentations.py:111-141 uses 114/255), and applies the first
 * Conv(3->Cout, k3 s1) + SiLU, writing NHWC bf16 [B,Hp,W,Cout]. weight is fp32 [Cout][3][3][3]
 * (co,kh,kw,ci), bias fp32.
 */
int adayolo_stem_fwd(const float* img, const float* weight, const float* bias, void* out, int out_cstride,
                     int B, int H, int W, int Hp, int pad_top, float pad_value, int Cout, void* stream);

/*
 * The hand-over alone, for detectors whose first conv is not 3->32 (width-scaled checkpoints,
 * yolov3/models/yolo.py:301-336 `width_multiple`): letterbox as above and write NHWC bf16 [B,Hp,W,8] with
 * channels 3..7 zero; the first conv then runs through adayolo_conv_fwd with Cin = 8 on zero-padded weights.
 */
int adayolo_letterbox_pack(const float* img, void* out, int out_cstride, int B, int H, int W, int Hp, int pad_top,
                           float pad_value, void* stream);

/*
 * adayolo_stem_fwd fused with the first down-sampling block of yolov3.yaml (layer 1: Conv(32->64, k3 s2) + SiLU):
 * planar fp32 image in, NHWC bf16 [B, Hp/2, W/2, 64] out; the 32-channel stem output stays in LDS.
 * w_stem fp32 [32][3][3][3], w_down bf16 [64][3][3][32] (BN folded), biases fp32. Hp and W even.
 * Optionally (w_next != NULL) also the 1x1 conv that follows (layer 2's Bottleneck.cv1: 64 -> 32, w_next bf16 [32][64])
 * + SiLU, computed from the output tile while it is in LDS: out_next NHWC bf16 [B, Hp/2, W/2, 32].
 */
int adayolo_stem_down_fwd(const float* img, const float* w_stem, const float* b_stem, const void* w_down,
                          const float* b_down, void* out, int out_cstride, int B, int H, int W, int Hp, int pad_top,
                          float pad_value, const void* w_next, const float* b_next, void* out_next, int out_next_cstride,
                          void* stream);

/*
 * Training forward of `Conv` (common.py:45-59) in ONE launch: pre = bf16(conv + bias) is stored (the backward needs
 * silu'(pre)), out = act(pre) (+ residual) is computed from that ROUNDED value — exactly adayolo_conv_fwd_variant with
 * ADAYOLO_ACT_NONE into `pre` followed by adayolo_silu_fwd(pre, residual, out), bit for bit. Served by the kernels whose
 * epilogue has the second output: variants 5 / 22 / 26 / 27, 60 and 80 / 85 (EINVAL for any other; ESHAPE when the named kernel
 * does not take the shape — the caller then keeps the two launches).
 */
int adayolo_conv_keep_fwd(const void* in, int in_cstride, const void* weight, const float* bias, const void* residual,
                          int res_cstride, void* out, int out_cstride, void* pre, int pre_cstride, int B, int H, int W, int Cin,
                          int Cout, int ksize, int stride, int act, int variant, void* stream);

/*
 * Split-K form of variant 60 for layers with few output tiles and a long reduction (e.g. Conv 512 -> 1024 k3 on 8 x 16 x 16
 * pixels: 64 tiles of 72 k-tiles on 256 CUs — the shapes of the RL training step, config 4): variant =
 * ADAYOLO_SPLITK_BASE + S, 2 <= S <= 16. The k-tiles (64 input channels of one tap) of every 256 px x 128 ch output tile
 * are cut into S equal ranges, one workgroup each; fp32 partial tiles go through `workspace`, the workgroup that finishes
 * a tile last adds them in range order (deterministic) and applies bias / activation / residual (and stores `pre` when
 * given, as adayolo_conv_keep_fwd). Same layouts and checks as adayolo_conv_fwd_variant; Cin % 64 == 0, Cout % 128 == 0,
 * the k-tile count divisible by S with at least 4 per range, tiles x S <= 512 — ESHAPE otherwise.
 * `workspace`: adayolo_conv_splitk_workspace_bytes(...) bytes (0 = this split does not serve the shape), ZEROED ONCE by
 * the caller before the first launch (every launch leaves its tickets zero); one workspace serves the launches of one
 * stream, launches that may overlap need their own.
 */
#define ADAYOLO_SPLITK_BASE 100
size_t adayolo_conv_splitk_workspace_bytes(int B, int H, int W, int Cin, int Cout, int ksize, int stride, int variant);
int adayolo_conv_splitk_fwd(const void* in, int in_cstride, const void* weight, const float* bias, const void* residual,
                            int res_cstride, void* out, int out_cstride, void* pre, int pre_cstride, int B, int H, int W,
                            int Cin, int Cout, int ksize, int stride, int act, int variant, void* workspace,
                            size_t workspace_bytes, void* stream);

/*
 * Backward of `Conv` + SiLU through the FROZEN detector, one layer's SiLU' inside the launch that completes its gradient:
 * g = conv(in, weight) + bias (+ residual) is dL/d(output of the layer whose pre-activation is `pre`), rounded to bf16;
 * `out` (may be NULL: not needed again) receives g, `grad_pre` receives bf16(g * silu'(pre)) computed from that ROUNDED g —
 * bit for bit adayolo_conv_fwd_variant(..., ADAYOLO_ACT_NONE) into a buffer followed by adayolo_silu_bwd on it.
 * `in` / `weight` are what the data-gradient conv takes anyway (dL/d(pre) of the consumer layer, its weights transposed
 * and flipped). Variants 5 / 22 / 26 / 27 / 60 / 80 / 85 and the split-K ones (then with their workspace, see above; NULL / 0
 * otherwise); EINVAL for any other, ESHAPE when the named kernel does not take the shape.
 */
int adayolo_conv_dsilu_fwd(const void* in, int in_cstride, const void* weight, const float* bias, const void* residual,
                           int res_cstride, void* out, int out_cstride, const void* pre, int pre_cstride, void* grad_pre,
                           int gp_cstride, int B, int H, int W, int Cin, int Cout, int ksize, int stride, int variant,
                           void* workspace, size_t workspace_bytes, void* stream);

/*
 * Data gradient of a STRIDE-2 `Conv` (k3, p1; yolov3.yaml's down-sampling layers) without zero insertion: over the
 * Ho x Wo grid of grad_out = dL/d(pre-activation of that layer) [B,Ho,Wo,Cout], a 2x2 stride-1 convolution (taps (dh,dw) in
 * {0,1}^2 read grad_out[i+dh, j+dw], zero beyond the far edges) with 4*Cin output channels, group p = 2a+b of which is the
 * gradient of input pixel (2i+a, 2j+b): g [B,2Ho,2Wo,Cin] (+ residual, same shape) rounded to bf16 -> grad_in (may be NULL
 * when grad_pre is given); with pre / grad_pre (both or neither) also grad_pre = bf16(g * silu'(pre)) as in
 * adayolo_conv_dsilu_fwd. 16/9 of the multiply-adds of the exact form, 4/9 of adayolo_zero_insert2x + the 3x3 conv.
 * weight4: bf16 [4*Cin][2][2][Cout] with
 *     weight4[(2a+b)*Cin + c][dh][dw][co] = W[co][KH(a,dh)][KH(b,dw)][c],  KH(0,0) = 1, KH(1,0) = 2, KH(1,1) = 0,
 * and zero for (parity 0, offset 1) (W = the layer's [Cout][3][3][Cin] weights); bias4: fp32 zeros [4*Cin].
 * Variants 5 / 22 / 26 / 27 / 60 and the split-K ones (workspace as above, sized with ksize = 2, H = Ho, W = Wo,
 * Cin = Cout, Cout = 4*Cin of this call); ESHAPE where the named kernel does not take the shape.
 */
int adayolo_conv_s2grad_fwd(const void* grad_out, int go_cstride, const void* weight4, const float* bias4, const void* residual,
                            int res_cstride, void* grad_in, int gi_cstride, const void* pre, int pre_cstride, void* grad_pre,
                            int gp_cstride, int B, int Ho, int Wo, int Cout, int Cin, int variant, void* workspace,
                            size_t workspace_bytes, void* stream);

/* adayolo_stem_fwd with the activation selectable (ADAYOLO_ACT_NONE keeps the pre-activation for training). */
int adayolo_stem_fwd_act(const float* img, const float* weight, const float* bias, void* out, int out_cstride,
                         int B, int H, int W, int Hp, int pad_top, float pad_value, int Cout, int act, void* stream);

/*
 * Training forward of the stem in ONE launch: pre = bf16(conv + bias) is stored, out = SiLU of that ROUNDED value — bit for bit
 * adayolo_stem_fwd_act(ADAYOLO_ACT_NONE) into `pre` followed by adayolo_silu_fwd(pre, NULL, out).
 */
int adayolo_stem_keep_fwd(const float* img, const float* weight, const float* bias, void* out, int out_cstride, void* pre,
                          int pre_cstride, int B, int H, int W, int Hp, int pad_top, float pad_value, int Cout, void* stream);

/*
 * Training side of the frozen reward model (train.py:239-243,262-271,341-342: weights have requires_grad False, the
 * loss back-propagates THROUGH the detector to the retouched image). The backward convolutions are adayolo_conv_fwd
 * on transposed / spatially flipped weights (stride-2 layers: on the zero-inserted output gradient); the rest:
 *   adayolo_silu_fwd        out = silu(pre) (+ residual)            Conv.act / Bottleneck shortcut, common.py:45-59,110-120
 *   adayolo_silu_bwd        grad_pre = grad_out * silu'(pre) (may be NULL); grad_res (=|+=) grad_out (may be NULL)
 *   adayolo_zero_insert2x   out[b,2y,2x] = in[b,y,x], 0 elsewhere; out is [B,H,W,C] with (H+1)/2 == Ho
 *   adayolo_upsample2x_bwd  grad_in[b,y,x] (=|+=) sum of the 2x2 block of grad_out [B,2H,2W,C]
 *   adayolo_image_grad      NHWC bf16 [B,Hp,W,>=3] -> planar fp32 [B,3,H,W] (rows pad_top .. pad_top+H: letterbox removed)
 * NHWC bf16 tensors with explicit channel strides, C % 8 == 0, npix = B*H*W.
 */
int adayolo_silu_fwd(const void* pre, int pre_cstride, const void* residual, int res_cstride, void* out,
                     int out_cstride, long npix, int C, void* stream);
int adayolo_silu_bwd(const void* grad_out, int go_cstride, const void* pre, int pre_cstride, void* grad_pre,
                     int gp_cstride, void* grad_res, int gr_cstride, int accumulate_res, long npix, int C, void* stream);
int adayolo_zero_insert2x(const void* in, int in_cstride, void* out, int out_cstride, int B, int Ho, int Wo, int H, int W,
                          int C, void* stream);
int adayolo_upsample2x_bwd(const void* grad_out, int go_cstride, void* grad_in, int gi_cstride, int accumulate, int B,
                           int H, int W, int C, void* stream);
int adayolo_image_grad(const void* grad_nhwc, int g_cstride, float* grad_img, int B, int H, int W, int Hp, int pad_top,
                       void* stream);

/* Nearest 2x up-sampling of in[B,H,W,C] into a channel slice of out[B,2H,2W,*] (Upsample + Concat). */
int adayolo_upsample2x(const void* in, int in_cstride, void* out, int out_cstride,
                       int B, int H, int W, int C, void* stream);

/*
 * Detect head decode (eval): raw[b,y,x, a*no + j] (NHWC bf16, channel stride raw_cstride, na*no valid
 * channels) -> pred[b, row_offset + (a*ny + y)*nx + x, j] fp32 with
 *   s = sigmoid(raw); xy = (2 s - 0.5 + grid) * det_stride; wh = (2 s)^2 * anchor_px; others = s.
 * anchors_px: fp32 [na][2] (pixel units). pred has `pred_rows` rows of `no` floats per image.
 */
int adayolo_detect_decode(const void* raw, int raw_cstride, float* pred, int pred_rows, int row_offset,
                          const float* anchors_px, float det_stride,
                          int B, int ny, int nx, int na, int no, void* stream);

/*
 * Greedy IoU NMS (eval harness; replaces the reference's call of torchvision.ops.nms at
 * yolov3/utils/general.py:949 — torchvision is an un-vendored dependency, pinned 0.15.2 in requirements.txt).
 * boxes_xyxy: fp32 [n][4], ALREADY sorted by descending score and offset by class (general.py:945-948).
 * A box is kept unless an earlier kept box has IoU > iou_thres with it. keep[0..*num_keep) receives the kept
 * indices in score order (at most max_det; the rest of keep[] is set to -1). workspace: device memory of
 * adayolo_nms_workspace_bytes(n) bytes. No host synchronisation.
 */
size_t adayolo_nms_workspace_bytes(int n);
int adayolo_nms(const float* boxes_xyxy, int n, float iou_thres, int max_det, void* workspace,
                int32_t* keep, int32_t* num_keep, void* stream);

/*
 * Per-image detection loss of the RL reward on the RAW head maps, with its gradient (training path). Replaces the
 * caller-side Python of train.py:175-197 — `ComputeLossBatch` called once per sample with the image index of its targets
 * set to 0 — i.e. ComputeLoss.__call__ (yolov3/utils/loss.py:115-170 / :262-318) and bbox_iou(CIoU=True)
 * (yolov3/utils/metrics.py:222-260). The target assignment (build_targets, loss.py:320-380) stays with the caller:
 * per layer it hands over n matches, idx[j] = (image, anchor, gj, gi, class) and box[j] = (tx, ty, tw, th, anchor_w,
 * anchor_h) in cell units (tx, ty relative to the cell).
 *   loss[b] = hyp_box * sum_l mean_{matches of b in l}(1 - CIoU)  +  hyp_cls * sum_l mean_{matches x classes} BCE(cls)
 *           + hyp_obj * sum_l balance_l * mean_{anchors x cells of b} BCE(obj, tobj),  tobj = clamp(CIoU, 0) at matched cells
 *   (a cell matched more than once keeps its LAST match's value, the result of a sequential index assignment).
 * raw: the detector's head map, NHWC bf16 [B][ny][nx][cs], channel = anchor * no + (x, y, w, h, obj, classes...).
 * Scratch (device, caller-owned, written by _fwd and read by _bwd) per layer: part fp32 [B][3], tobj fp32 [B][na][ny][nx],
 * cnt fp32 [B]; per call: ticket int32 [B], ZERO before the first call (each call leaves it zero). nc <= 128.
 * adayolo_detloss_bwd writes EVERY element of grad (NHWC bf16 [B][ny][nx][grad_cs], grad_cs % 8 == 0,
 * >= na*no; channels past na*no are zeroed): d (sum_b grad_loss[b] * loss[b]) / d raw. Both are bit-reproducible.
 */
typedef struct adayolo_loss_layer {
    const void* raw; int cs; int ny, nx; float balance;
    const int32_t* idx; const float* box; int n;
    float* part; float* tobj; float* cnt;
    void* grad; int grad_cs;
} adayolo_loss_layer;
typedef struct adayolo_loss_args {
    adayolo_loss_layer layer[4];
    int nl, B, na, nc, no;
    float hyp_box, hyp_obj, hyp_cls, cp, cn, cls_pw, obj_pw;
    float* loss;                 /* [B] */
    int32_t* ticket;             /* [B], see above */
    const float* grad_loss;      /* [B], _bwd only */
} adayolo_loss_args;
int adayolo_detloss_fwd(const adayolo_loss_args* args, void* stream);
int adayolo_detloss_bwd(const adayolo_loss_args* args, void* stream);

const char* adayolo_strerror(int code);
int adayolo_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ADAYOLO_H_ */
