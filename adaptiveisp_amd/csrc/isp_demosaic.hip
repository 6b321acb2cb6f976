// Bayer demosaic front-end — an EXTENSION: the reference's ISP starts from a 3-channel linear
// image and has no demosaic; only the inverse packing exists (`mosaic`, isp/unprocess_np.py:82-98,
// `reconstruct_bayer` :111-128). north_star asks for a raw-Bayer entry to the path, so this kernel is defined by its
// own oracle (oracle_demosaic in oracle/isp_oracle.c) and checked for consistency with that packing.
//
// raw: uint16 [B,H,W], one colour sample per pixel (pattern gives the position of the red sample in the 2x2 cell);
// out: planar fp32 [B,3,H,W] = bilinear interpolation of the normalised samples s = (raw - black) * 1/(white - black):
//   at a sampled colour: the sample;  green at red/blue sites: ((N + S) + (W + E)) / 4;
//   red/blue at green sites: (W + E) / 2 or (N + S) / 2;  red at blue sites (and v.v.): ((NW + NE) + (SW + SE)) / 4;
// image borders mirror without repeating the edge sample (index -1 -> 1, H -> H-2), which preserves the Bayer phase.
//
// Memory-bound: 2 B/px in, 12 B/px out. One workgroup = a 128 x 32 pixel tile: the tile plus a one-pixel ring is
// staged in LDS as fp32 samples (34 x 130), then each lane produces 2x2 cells — two adjacent pixels per row, so every
// plane is written with 8-byte stores, 512 contiguous bytes per wave and row.
#include "isp_internal.h"

namespace adaisp {
namespace {

constexpr int TW = 128, TH = 32, LW = TW + 2, LH = TH + 2;

// one reflection is all a valid output ever needs; the clamp only keeps the staging of rows / columns beyond the image
// (tiles that overhang it) inside the allocation
__device__ __forceinline__ int mirror(int i, int n) {
    const int m = i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i);
    return min(max(m, 0), n - 1);
}

__global__ __launch_bounds__(256) void k_demosaic(const unsigned short* __restrict__ raw, float* __restrict__ out,
                                                  int H, int W, int ry, int rx, float black, float inv_range) {
    __shared__ float s[LH][LW + 2];
    const int b = blockIdx.z, x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const unsigned short* __restrict__ src = raw + (long)b * H * W;
    for (int i = threadIdx.x; i < LH * LW; i += 256) {
        const int ly = i / LW, lx = i - ly * LW;
        const int y = mirror(y0 + ly - 1, H), x = mirror(x0 + lx - 1, W);
        s[ly][lx] = ((float)src[(long)y * W + x] - black) * inv_range;
    }
    __syncthreads();
    const long plane = (long)H * W;
    float* __restrict__ o = out + (long)b * 3 * plane;
    // 64 x 16 cells per tile, 4 per thread: thread -> cell column (tid & 63), cell rows (tid >> 6) + 4*k
    const int cx = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int cy = (threadIdx.x >> 6) + 4 * k;
        const int gx = x0 + 2 * cx, gy = y0 + 2 * cy;
        if (gx >= W || gy >= H) continue;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            float r[2], g[2], bl[2];
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int ly = 2 * cy + dy + 1, lx = 2 * cx + dx + 1;
                const float c = s[ly][lx];
                const float n = s[ly - 1][lx], so = s[ly + 1][lx], w = s[ly][lx - 1], e = s[ly][lx + 1];
                const float cross = ((n + so) + (w + e)) * 0.25f;
                const float diag = ((s[ly - 1][lx - 1] + s[ly - 1][lx + 1]) + (s[ly + 1][lx - 1] + s[ly + 1][lx + 1])) * 0.25f;
                const float horiz = (w + e) * 0.5f, vert = (n + so) * 0.5f;
                const int py = (gy + dy - ry) & 1, px = (gx + dx - rx) & 1;      // 0,0 = red site; 1,1 = blue site
                if (py == 0 && px == 0) { r[dx] = c; g[dx] = cross; bl[dx] = diag; }
                else if (py == 0) { r[dx] = horiz; g[dx] = c; bl[dx] = vert; }
                else if (px == 0) { r[dx] = vert; g[dx] = c; bl[dx] = horiz; }
                else { r[dx] = diag; g[dx] = cross; bl[dx] = c; }
            }
            const long off = (long)(gy + dy) * W + gx;
            *reinterpret_cast<float2*>(o + off) = make_float2(r[0], r[1]);
            *reinterpret_cast<float2*>(o + plane + off) = make_float2(g[0], g[1]);
            *reinterpret_cast<float2*>(o + 2 * plane + off) = make_float2(bl[0], bl[1]);
        }
    }
}

}  // namespace

hipError_t launch_demosaic(const uint16_t* raw, float* out, int B, int H, int W, int pattern, float black, float white,
                           hipStream_t s) {
    const int ry = pattern >> 1, rx = pattern & 1;
    dim3 grid((W + TW - 1) / TW, (H + TH - 1) / TH, B);
    hipLaunchKernelGGL(k_demosaic, grid, dim3(256), 0, s, raw, out, H, W, ry, rx, black, 1.0f / (white - black));
    return hipGetLastError();
}

}  // namespace adaisp
