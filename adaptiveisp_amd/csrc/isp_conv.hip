// Small-stencil ISP filters for gfx950: the 3x3 sharpen pair (adjust_sharpness / sharpness) and
// the 5x5 gaussian unsharp mask. LDS-staged 2-D tiles: a 256-thread workgroup owns a 64x32 output
// tile; the tile plus halo of all three planes is staged once into LDS with 16-B global loads
// (quad-aligned columns x0-4 .. x0+67), then every lane produces 4 px x 2 rows per plane and
// stores 16 B per plane per row. HBM traffic stays at the algorithmic 24 B/px: halo re-reads of
// neighbouring tiles are served by L2.
//
// Reference: isp/sharpen.py:105-142 (adjust_sharpness), :145-182 (sharpness), :63-102 (unsharp_mask).
#include "isp_internal.h"

#ifndef ISP_CONV_NT_LD
#define ISP_CONV_NT_LD 0
#endif

namespace adaisp {
namespace {

constexpr int kThreads = 256;
constexpr int TW = 64, TH = 32;
constexpr int PITCH = TW + 8;           // quad-aligned columns x0-4 .. x0+TW+3
constexpr int MAXR = 2;
constexpr int ROWS_MAX = TH + 2 * MAXR;

enum Mode { kAdjust = 0, kSharpness = 1, kUSM = 2 };

// torch 'reflect' padding index (edge not repeated); valid for -n < i < 2n-1.
__device__ __forceinline__ int reflect(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

template <int R, int MODE, bool VEC>
__device__ void conv_tile(float* __restrict__ lds, const float* __restrict__ in, float* __restrict__ out,
                          const float* __restrict__ p, int H, int W) {
    constexpr int ROWS = TH + 2 * R;
    constexpr int QPR = PITCH / 4;  // quads per staged row
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const long plane = (long)H * W;

    // ---- stage tile + halo of the 3 planes ------------------------------------------------------
    for (int q = tid; q < 3 * ROWS * QPR; q += kThreads) {
        const int c = q / (ROWS * QPR);
        const int rem = q - c * (ROWS * QPR);
        const int ly = rem / QPR, lq = rem - ly * QPR;
        int gy = y0 - R + ly;
        const int gx = x0 - 4 + 4 * lq;
        const float* src = in + c * plane;
        float4 v;
        if (MODE == kUSM) gy = reflect(gy, H);
        const bool row_ok = (gy >= 0) && (gy < H);
        if (VEC && row_ok && gx >= 0 && gx + 3 < W) {
            v = *reinterpret_cast<const float4*>(src + (long)gy * W + gx);
        } else {
            float t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int xx = gx + k;
                if (MODE == kUSM) {
                    // columns further than the reflect range (only in the unused part of the quad margin
                    // or beyond the image's right edge of a partial tile) are never consumed
                    xx = (xx > -W && xx < 2 * W - 1) ? reflect(xx, W) : 0;
                    t[k] = row_ok ? src[(long)gy * W + xx] : 0.0f;
                } else {
                    t[k] = (row_ok && xx >= 0 && xx < W) ? src[(long)gy * W + xx] : 0.0f;
                }
            }
            v = make_float4(t[0], t[1], t[2], t[3]);
        }
        *reinterpret_cast<float4*>(lds + (c * ROWS + ly) * PITCH + 4 * lq) = v;
    }

    // ---- per-image weights ------------------------------------------------------------------------
    float w[2 * R + 1][2 * R + 1];
    float amount;
    if (MODE == kUSM) {
        // _get_gaussian_kernel1d (isp/sharpen.py:15-23): pdf = exp(-0.5 * (x / sigma)^2), x = -2..2
        const float sigma = p[0];
        amount = p[1];
        float g1[5], sum = 0.0f;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const float t = (float)(i - 2) / sigma;
            g1[i] = expf(-0.5f * (t * t));
            sum += g1[i];
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) g1[i] = g1[i] / sum;
#pragma unroll
        for (int i = 0; i < 2 * R + 1; ++i)
#pragma unroll
            for (int j = 0; j < 2 * R + 1; ++j) w[i][j] = g1[i] * g1[j];
    } else {
        amount = p[0];
        const float a = 1.0f / 13.0f, c5 = 5.0f / 13.0f;  // ones(3,3) with centre 5, divided by its sum
#pragma unroll
        for (int i = 0; i < 2 * R + 1; ++i)
#pragma unroll
            for (int j = 0; j < 2 * R + 1; ++j) w[i][j] = (i == R && j == R) ? c5 : a;
    }
    __syncthreads();

    // ---- 4 px x 2 rows per lane per plane ---------------------------------------------------------
    const int tx = tid & 15, ty = tid >> 4;
    const int gx = x0 + 4 * tx;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int ly = ty + 16 * half;          // output row inside the tile
        const int gy = y0 + ly;
        if (gy >= H || gx >= W) continue;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v[2 * R + 1][4 + 2 * R];
#pragma unroll
            for (int i = 0; i < 2 * R + 1; ++i)
#pragma unroll
                for (int j = 0; j < 4 + 2 * R; ++j)
                    v[i][j] = lds[(c * ROWS + ly + i) * PITCH + 4 + 4 * tx - R + j];
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float blur = 0.0f;
#pragma unroll
                for (int i = 0; i < 2 * R + 1; ++i)
#pragma unroll
                    for (int j = 0; j < 2 * R + 1; ++j) blur = fmaf(w[i][j], v[i][k + j], blur);
                const float ctr = v[R][k + R];
                if (MODE != kUSM) {
                    // the 1-px frame keeps the image (valid conv + zero-padded mask + where, sharpen.py:133-138)
                    const int xx = gx + k;
                    if (gy == 0 || gy == H - 1 || xx == 0 || xx >= W - 1) blur = ctr;
                }
                float r;
                if (MODE == kAdjust) r = ctr * amount + blur * (1.0f - amount);
                else r = ctr + (ctr - blur) * amount;
                o[k] = clamp01(r);
            }
            float* dst = out + c * plane + (long)gy * W + gx;
            if (VEC && gx + 3 < W) {
                *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (gx + k < W) dst[k] = o[k];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Row-sliding variant (used whenever rows are 16-byte aligned): no LDS at all. A wave owns a 256-px wide strip of
// ONE plane and walks down `rs` output rows; each lane holds 4 consecutive pixels of the current 3 (5) input rows
// in registers and gets its horizontal neighbours from the adjacent lanes with DPP wave shifts
// (v_mov_b32_dpp wave_shl/shr). Every input row is loaded once per strip with one coalesced 16-B load per lane
// ((rs+2R)/rs re-read factor along y only), every output row is one 16-B store per lane.
//
// The walk is software-pipelined in groups of G = 2R+1 rows: the 16-B loads of the NEXT group (and the two strip-edge
// lanes' outside neighbours) are all issued before the current group is computed, so a wave keeps G rows in flight
// instead of one (one load per trip = one full memory round trip per row: 4.4 TB/s at 4K; the kernel is HBM-bound).
// With G rows per group the 2R+1-row window is a register ring with compile-time slots: no row shifting moves.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dpp_from_prev(float v) {   // lane i <- lane i-1
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ float dpp_from_next(float v) {   // lane i <- lane i+1
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x130, 0xF, 0xF, false));
}

struct RawRow {          // what a lane fetches for one input row: its 4 pixels + (strip-edge lanes only) 2 outside pixels
    float4 c;
    float e0, e1;        // left edge lane: (x-2, x-1); right edge lane: (x+4, x+5); R = 1 uses e1 (left) / e0 (right)
};

// issue the loads of input row y; nothing is consumed here, so consecutive calls put all their loads in flight
template <int R, int MODE>
__device__ __forceinline__ void issue_row(const float* __restrict__ src, int y, int H, int W, int gx, bool active,
                                          bool edge_l, bool edge_r, RawRow& r) {
    int yy = y;
    if (MODE == kUSM) yy = reflect(y, H);
    const bool row_ok = yy >= 0 && yy < H;
    const float* rowp = src + (long)(row_ok ? yy : 0) * W;
    r.c = make_float4(0.f, 0.f, 0.f, 0.f);
    r.e0 = 0.f; r.e1 = 0.f;
    if (active && row_ok) {
#if ISP_CONV_NT_LD
        r.c = ld4_nt(reinterpret_cast<const float4*>(rowp + gx));
#else
        r.c = *reinterpret_cast<const float4*>(rowp + gx);
#endif
        if (edge_l) {
            if (MODE == kUSM) { r.e0 = rowp[reflect(gx - 2, W)]; r.e1 = rowp[reflect(gx - 1, W)]; }
            else if (gx > 0) r.e1 = rowp[gx - 1];
        } else if (edge_r) {                               // lane 63 of a strip that is not the image's last: gx + 5 < W
            r.e0 = rowp[gx + 4];
            if (MODE == kUSM) r.e1 = rowp[gx + 5];
        }
    }
}

// the lane's 4 pixels plus R neighbours on each side -> v[0 .. 4+2R)
template <int R, int MODE>
__device__ __forceinline__ void finish_row(const RawRow& r, bool active, bool edge_l, bool edge_r, bool last_col,
                                           float* v) {
    const float4 c = r.c;
    v[R + 0] = c.x; v[R + 1] = c.y; v[R + 2] = c.z; v[R + 3] = c.w;
    float l1 = dpp_from_prev(c.w), r1 = dpp_from_next(c.x);
    float l2 = 0.f, r2 = 0.f;
    if (R == 2) { l2 = dpp_from_prev(c.z); r2 = dpp_from_next(c.y); }
    if (edge_l || !active) { l2 = r.e0; l1 = r.e1; }        // inactive lanes hold zeros
    if (edge_r) { r1 = r.e0; r2 = r.e1; }
    if (last_col) {                                          // the image's right border (W % 4 == 0: gx + 4 == W)
        // reflect(W) = W - 2, reflect(W + 1) = W - 3: the lane's own pixels; the 3x3 filters see zero (and keep the frame)
        r1 = (MODE == kUSM) ? c.z : 0.f;
        r2 = (MODE == kUSM) ? c.y : 0.f;
    }
    if (R == 1) { v[0] = l1; v[5] = r1; }
    else { v[0] = l2; v[1] = l1; v[6] = r1; v[7] = r2; }
}

// The walk itself. The wave's lanes cover pixels gx = x_first + 4 * lane < x_read_end of plane `src`; output rows
// [y_begin, y_end) are computed, rows < y_store_end and pixels < x_store_end are stored (the fused-pooling cut computes
// the row / quad it shares with the next pool window without storing it; a walk over TWO pool windows passes the first
// window's end, the second one's begin and `colsum2` — a row inside both adds to both). `colsum` accumulates the lane's four output
// columns over the computed rows, ascending y from its initial value (the pooled planes' column sums).
template <int R, int MODE, int NG>
__device__ void conv_rows_core(const float* __restrict__ src, float* __restrict__ dst, const float* __restrict__ p, int H,
                               int W, int x_first, int x_read_end, int x_store_end, int y_begin, int y_end,
                               int y_store_end, float (&colsum)[4], int y_first_end = 1 << 30, int y_second_begin = 1 << 30,
                               float* colsum2 = nullptr) {
    constexpr int G = 2 * R + 1, GS = G * NG;        // ring period, rows per pipeline group
    const int lane = threadIdx.x & 63;
    const int gx = x_first + 4 * lane;
    const bool active = gx < x_read_end;             // W % 4 == 0 on this path
    const bool last_col = active && gx + 4 >= W;
    const bool edge_l = lane == 0, edge_r = active && gx + 4 >= x_read_end && !last_col;
    const bool store_x = active && gx < x_store_end;

    float w[G][G];
    float amount;
    if (MODE == kUSM) {
        const float sigma = p[0];
        amount = p[1];
        float g1[5], sum = 0.0f;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const float t = (float)(i - 2) / sigma;
            g1[i] = expf(-0.5f * (t * t));
            sum += g1[i];
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) g1[i] = g1[i] / sum;
#pragma unroll
        for (int i = 0; i < G; ++i)
#pragma unroll
            for (int j = 0; j < G; ++j) w[i][j] = g1[i] * g1[j];
    } else {
        amount = p[0];
        const float a = 1.0f / 13.0f, c5 = 5.0f / 13.0f;
#pragma unroll
        for (int i = 0; i < G; ++i)
#pragma unroll
            for (int j = 0; j < G; ++j) w[i][j] = (i == R && j == R) ? c5 : a;
    }

    // ring of input rows: row (y_begin - R + n) lives in slot n % G
    float win[G][4 + 2 * R];
    RawRow cur[GS], nxt[GS];
    {
        RawRow head[2 * R];
#pragma unroll
        for (int i = 0; i < 2 * R; ++i) issue_row<R, MODE>(src, y_begin - R + i, H, W, gx, active, edge_l, edge_r, head[i]);
#pragma unroll
        for (int u = 0; u < GS; ++u)
            issue_row<R, MODE>(src, y_begin + R + u, H, W, gx, active, edge_l, edge_r, cur[u]);
#pragma unroll
        for (int i = 0; i < 2 * R; ++i) finish_row<R, MODE>(head[i], active, edge_l, edge_r, last_col, win[i]);
    }
    for (int y0 = y_begin; y0 < y_end; y0 += GS) {
        // loads of the next group first (rows past the band's last input row are never consumed: skip them)
#pragma unroll
        for (int u = 0; u < GS; ++u) {
            const int yn = y0 + GS + u;                  // output row that would need input row yn + R
            if (yn < y_end) issue_row<R, MODE>(src, yn + R, H, W, gx, active, edge_l, edge_r, nxt[u]);
            else { nxt[u].c = make_float4(0.f, 0.f, 0.f, 0.f); nxt[u].e0 = 0.f; nxt[u].e1 = 0.f; }
        }
#pragma unroll
        for (int u = 0; u < GS; ++u) {
            const int y = y0 + u;
            finish_row<R, MODE>(cur[u], active, edge_l, edge_r, last_col, win[(u + 2 * R) % G]);   // input row y + R
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float blur = 0.0f;
#pragma unroll
                for (int i = 0; i < G; ++i)
#pragma unroll
                    for (int j = 0; j < G; ++j) blur = fmaf(w[i][j], win[(u + i) % G][k + j], blur);
                const float ctr = win[(u + R) % G][k + R];
                if (MODE != kUSM) {
                    const int xx = gx + k;
                    if (y == 0 || y == H - 1 || xx == 0 || xx >= W - 1) blur = ctr;
                }
                const float r = (MODE == kAdjust) ? ctr * amount + blur * (1.0f - amount) : ctr + (ctr - blur) * amount;
                o[k] = clamp01(r);
            }
            if (y < y_end) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (y < y_first_end) colsum[k] += o[k];
                    if (y >= y_second_begin) colsum2[k] += o[k];
                }
                if (store_x && y < y_store_end) st4(reinterpret_cast<float4*>(dst + (long)y * W + gx), make_float4(o[0], o[1], o[2], o[3]));
            }
        }
#pragma unroll
        for (int u = 0; u < GS; ++u) cur[u] = nxt[u];
    }
}

template <int R, int MODE, int NG>
__device__ __forceinline__ void conv_rows(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ p,
                                          int H, int W, int c, int strip, int band, int rs) {
    const long plane = (long)H * W;
    const int x_end = min(W, strip * 256 + 256), y_begin = band * rs, y_end = min(H, y_begin + rs);
    float unused[4] = {0.f, 0.f, 0.f, 0.f};
    conv_rows_core<R, MODE, NG>(in + c * plane, out + c * plane, p, H, W, strip * 256, x_end, x_end, y_begin, y_end, y_end,
                                unused);
}

// R = 1: the two 3x3 sharpeners; R = 2: the 5x5 unsharp mask (its own kernel: the 5-row pipeline needs 134 VGPRs, which
// would cost the 3x3 filters two of their five waves per SIMD).
template <int R, int NG>
__global__ __launch_bounds__(kThreads) void k_conv_rows(const float* __restrict__ img, float* __restrict__ out,
                                                        const int32_t* __restrict__ ids, int uniform_op,
                                                        const float* __restrict__ params, int pstride, int H, int W,
                                                        int strips, int rs) {
    const int b = blockIdx.z;
    const int op = ids ? ids[b] : uniform_op;
    if (R == 1 && op != ADAISP_OP_SHARPEN && op != ADAISP_OP_SHARPEN_V2) return;
    if (R == 2 && op != ADAISP_OP_USM) return;
    const int wave = threadIdx.x >> 6;
    const int band = blockIdx.y * 4 + wave;          // 4 waves = 4 consecutive row bands
    if (band * rs >= H) return;
    const int c = blockIdx.x / strips, strip = blockIdx.x - c * strips;
    const long off = (long)b * 3 * H * W;
    const float* p = params + (long)b * pstride;
    if (R == 2) conv_rows<2, kUSM, NG>(img + off, out + off, p, H, W, c, strip, band, rs);
    else if (op == ADAISP_OP_SHARPEN) conv_rows<1, kAdjust, NG>(img + off, out + off, p, H, W, c, strip, band, rs);
    else conv_rows<1, kSharpness, NG>(img + off, out + off, p, H, W, c, strip, band, rs);
}

// The same walk cut along the pool windows, with the next step's 64x64 pooling fused: workgroup = (pool row, plane) over
// the whole width, wave w = the 256-pixel strip [256 w, 256 w + 256) — line-aligned like the pointwise family's cut
// (isp_pointwise.hip; round 3 cut along the pool columns: 220-px strips whose edge lines two waves fetched) — or, beyond
// 2048 px, strips w, w + 8, ... The lane's four output columns are summed down the window's rows (ascending y from 0.0f),
// go to LDS ([W] floats) and thread ox adds its pool cell's columns in ascending x: k_pool64's order, bit-identical.
// NW = pool windows per wave: with two, the 2R halo rows and the pipeline's head are paid once per ~22 rows instead of once
// per ~11 (reads 1.13 x the plane instead of 1.24 x at 720 rows): 8x720x1280 sharpen 46.1 -> 44.3 us, USM 62.3 -> 54.3; at
// 2160 rows a window is already 34 rows tall and two per wave leave too few, too long waves (4x2160x3840: 164 -> 201 us), so
// the launcher takes two only while a window is shorter than 16 rows (tools/isp_step_ab.py, profiles/round4_isp_step_ab.txt).
constexpr int kPoolWaves = 8;
template <int R, int NW>
__global__ __launch_bounds__(64 * kPoolWaves) void k_conv_rows_pool(const float* __restrict__ img, float* __restrict__ out,
                                                                    float* __restrict__ pooled, const int32_t* __restrict__ ids,
                                                                    int uniform_op, const float* __restrict__ params, int pstride,
                                                                    int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float colsum_lds[];   // [NW][256 x strips]
    const int b = blockIdx.z;
    const int op = ids ? ids[b] : uniform_op;
    if (R == 1 && op != ADAISP_OP_SHARPEN && op != ADAISP_OP_SHARPEN_V2) return;
    if (R == 2 && op != ADAISP_OP_USM) return;
    const int oy = blockIdx.x * NW, c = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = (int)blockDim.x >> 6;
    const int strips = (W + 255) >> 8, Wp = strips << 8;
    const long plane = (long)H * W;
    const float* src = img + ((long)b * 3 + c) * plane;
    float* dst = out + ((long)b * 3 + c) * plane;
    const float* p = params + (long)b * pstride;
    const int oy_last = min(oy + NW - 1, 63);
    const int ys = win_lo(oy, H), ye = win_hi(oy_last, H), y_own_end = oy_last == 63 ? H : win_lo(oy_last + 1, H);
    const int y_first_end = win_hi(oy, H), y_second_begin = NW > 1 && oy + 1 <= 63 ? win_lo(oy + 1, H) : (1 << 30);
    for (int sw = wave; sw < strips; sw += nwaves) {
        const int x_lo = 256 * sw, x_end = min(W, x_lo + 256);
        float acc[4] = {0.f, 0.f, 0.f, 0.f}, acc2[4] = {0.f, 0.f, 0.f, 0.f};
        if (R == 2) conv_rows_core<2, kUSM, 1>(src, dst, p, H, W, x_lo, x_end, x_end, ys, ye, y_own_end, acc, y_first_end, y_second_begin, acc2);
        else if (op == ADAISP_OP_SHARPEN) conv_rows_core<1, kAdjust, 1>(src, dst, p, H, W, x_lo, x_end, x_end, ys, ye, y_own_end, acc, y_first_end, y_second_begin, acc2);
        else conv_rows_core<1, kSharpness, 1>(src, dst, p, H, W, x_lo, x_end, x_end, ys, ye, y_own_end, acc, y_first_end, y_second_begin, acc2);
        *reinterpret_cast<float4*>(colsum_lds + x_lo + 4 * lane) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        if (NW > 1) *reinterpret_cast<float4*>(colsum_lds + Wp + x_lo + 4 * lane) = make_float4(acc2[0], acc2[1], acc2[2], acc2[3]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * NW; i += blockDim.x) {
        const int ox = i & 63, w = i >> 6, oyw = oy + w;
        if (oyw > 63) break;
        const int xs = win_lo(ox, W), xe = win_hi(ox, W);
        const float a = window_sum(colsum_lds + w * Wp, xs, xe);
        pooled[(((long)b * 3 + c) * 64 + oyw) * 64 + ox] = a / (float)(win_hi(oyw, H) - win_lo(oyw, H)) / (float)(xe - xs);
    }
}

template <bool VEC>
__global__ __launch_bounds__(kThreads) void k_conv(const float* __restrict__ img, float* __restrict__ out,
                                                   const int32_t* __restrict__ ids, int uniform_op,
                                                   const float* __restrict__ params, int pstride, int H, int W) {
    __shared__ __attribute__((aligned(16))) float lds[3 * ROWS_MAX * PITCH];
    const int b = blockIdx.z;
    const int op = ids ? ids[b] : uniform_op;
    const long off = (long)b * 3 * H * W;
    const float* p = params + (long)b * pstride;
    switch (op) {
        case ADAISP_OP_SHARPEN:    conv_tile<1, kAdjust, VEC>(lds, img + off, out + off, p, H, W); break;
        case ADAISP_OP_SHARPEN_V2: conv_tile<1, kSharpness, VEC>(lds, img + off, out + off, p, H, W); break;
        case ADAISP_OP_USM:        conv_tile<2, kUSM, VEC>(lds, img + off, out + off, p, H, W); break;
        default: break;
    }
}

}  // namespace

hipError_t launch_conv_pool(const Batch& a, float* pooled, const PoolGeom& g, hipStream_t s) {
    const bool want3 = a.ids ? true : (a.uniform_op == ADAISP_OP_SHARPEN || a.uniform_op == ADAISP_OP_SHARPEN_V2);
    const bool want5 = a.ids ? !(a.flags & ADAISP_NO_USM) : (a.uniform_op == ADAISP_OP_USM);
    (void)g;
    const int strips = (a.W + 255) / 256;
    const int nw = a.H < 16 * 64 ? 2 : 1;
    const dim3 grid((64 + nw - 1) / nw, 3, a.B), block(64 * (strips < kPoolWaves ? strips : kPoolWaves));
    const size_t smem = (size_t)nw * strips * 256 * sizeof(float);
#define CONV_POOL_LAUNCH(R, NW) hipLaunchKernelGGL((k_conv_rows_pool<R, NW>), grid, block, smem, s, a.img, a.out, pooled, a.ids, \
                                                   a.uniform_op, a.params, a.pstride, a.H, a.W)
    if (want3) { if (nw == 2) CONV_POOL_LAUNCH(1, 2); else CONV_POOL_LAUNCH(1, 1); }
    if (want5) { if (nw == 2) CONV_POOL_LAUNCH(2, 2); else CONV_POOL_LAUNCH(2, 1); }
#undef CONV_POOL_LAUNCH
    return hipGetLastError();
}

hipError_t launch_conv(const Batch& a, hipStream_t s) {
    const bool vec = (a.W % 4 == 0) && (a.W >= 8) && ((reinterpret_cast<uintptr_t>(a.img) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(a.out) & 15) == 0);
    dim3 grid((a.W + TW - 1) / TW, (a.H + TH - 1) / TH, a.B);
    if (vec) {
        // rows per wave, measured (profiles/round2_stencil_sweep.txt, 8x720x1280 and 4x2160x3840): taller bands re-read fewer
        // halo rows ((rs + 2R) / rs) but leave fewer waves; 15 is best for the 3x3 filters at both sizes (30: -7 % at 4K,
        // 60: -35 %), 30 for the 5x5 (its halo is twice as tall). Multiples of 3 and 5, the pipeline's group sizes.
        const int strips = (a.W + 255) / 256;
        const int rs3 = 15, rs5 = 30;
        // per-image ids live on the device: both families are enqueued (workgroups of the other family's images return at
        // once) unless the caller rules the unsharp mask out (ADAISP_NO_USM: it is not in the policy's filter list)
        const bool want3 = a.ids ? true : (a.uniform_op != ADAISP_OP_USM);
        const bool want5 = a.ids ? !(a.flags & ADAISP_NO_USM) : (a.uniform_op == ADAISP_OP_USM);
        if (want3)
            hipLaunchKernelGGL((k_conv_rows<1, 1>), dim3(3 * strips, ((a.H + rs3 - 1) / rs3 + 3) / 4, a.B), dim3(kThreads), 0, s,
                               a.img, a.out, a.ids, a.uniform_op, a.params, a.pstride, a.H, a.W, strips, rs3);
        if (want5)
            hipLaunchKernelGGL((k_conv_rows<2, 1>), dim3(3 * strips, ((a.H + rs5 - 1) / rs5 + 3) / 4, a.B), dim3(kThreads), 0, s,
                               a.img, a.out, a.ids, a.uniform_op, a.params, a.pstride, a.H, a.W, strips, rs5);
    } else
        hipLaunchKernelGGL(k_conv<false>, grid, dim3(kThreads), 0, s, a.img, a.out, a.ids, a.uniform_op, a.params,
                           a.pstride, a.H, a.W);
    return hipGetLastError();
}

}  // namespace adaisp
