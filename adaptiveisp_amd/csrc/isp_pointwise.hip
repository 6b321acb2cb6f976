// Pointwise ISP filters for gfx950: exposure, gamma, white balance, CCM, tone curve, colour curve,
// contrast, black&white, saturation+ — plus the "zero image" case of an all-zero one-hot.
//
// One launch covers the whole batch: blockIdx.y = image, the op is block-uniform (read from the
// device-side id array or given by the host), so the switch below is a scalar branch and every
// op gets its own tight streaming loop. Planar CHW fp32; each lane moves 16 B per plane per access
// (global_load_dwordx4), two accesses in flight per plane. HBM-bound: 24 B/px algorithmic.
//
// Arithmetic follows the reference op order (compiled with -ffp-contract=off so that a*b+c stays
// two roundings like the ATen mul/add chains it replaces). Reference: isp/filters.py.
#include "isp_internal.h"

namespace adaisp {
namespace {

constexpr int kThreads = 256;

// ---- per-op functors: init(p) reads the image's regressed params, apply() maps one pixel ----------

struct OpZero {  // agent.py:154 with an all-zero one-hot row
    __device__ void init(const float*) {}
    __device__ __forceinline__ void apply(float& r, float& g, float& b) const { r = g = b = 0.0f; }
};

struct OpExposure {  // isp/filters.py:223-224   img * exp(p * ln2)
    float s;
    __device__ void init(const float* p) { s = expf(p[0] * 0.6931471805599453f); }
    __device__ __forceinline__ void apply(float& r, float& g, float& b) const { r *= s; g *= s; b *= s; }
};

// x^g for x >= 0.001 (normal, positive) and moderate g, ~2 ulp, on the hardware log2/exp2 units.
// libm's powf costs ~360 VALU slots (it also serves negative bases, integer exponents, subnormals) and made
// the gamma filter VALU-bound at 22 % of HBM; this form needs ~25. log2(x) is split into the v_log_f32
// result `hi` plus a first-order correction `lo` recovered from 2^hi, and the rounding error of g*hi is
// carried by an fma, so the absolute error of the exponent stays ~1e-7 even for |g*log2 x| ~ 30.
__device__ __forceinline__ float pow_pos(float x, float g) {
    const float hi = __builtin_amdgcn_logf(x);
    const float t = __builtin_amdgcn_exp2f(hi);
    const float lo = (x - t) * __builtin_amdgcn_rcpf(t) * 1.44269504088896341f;
    const float yh = g * hi;
    const float yl = fmaf(g, hi, -yh) + g * lo;
    const float r = __builtin_amdgcn_exp2f(yh);
    return fmaf(r, yl * 0.693147180559945309f, r);
}

struct OpGamma {  // isp/filters.py:244-245   pow(max(img, 0.001), gamma)
    float gm;
    __device__ void init(const float* p) { gm = p[0]; }
    __device__ __forceinline__ void apply(float& r, float& g, float& b) const {
        r = pow_pos(fmaxf(r, 0.001f), gm);
        g = pow_pos(fmaxf(g, 0.001f), gm);
        b = pow_pos(fmaxf(b, 0.001f), gm);
    }
};

struct OpWB {  // isp/filters.py:271-272   img * gains
    float s0, s1, s2;
    __device__ void init(const float* p) { s0 = p[0]; s1 = p[1]; s2 = p[2]; }
    __device__ __forceinline__ void apply(float& r, float& g, float& b) const { r *= s0; g *= s1; b *= s2; }
};

struct OpCCM {  // isp/filters.py:703-708,666-672   rows normalised by their sum, out[c] = sum_k img[k]*M[c][k]
    float m[9];
    __device__ void init(const float* p) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float rs = (p[3 * i] + p[3 * i + 1]) + p[3 * i + 2];
#pragma unroll
            for (int j = 0; j < 3; ++j) m[3 * i + j] = p[3 * i + j] / rs;
        }
    }
    __device__ __forceinline__ void apply(float& r, float& g, float& b) const {
        const float o0 = (r * m[0] + g * m[1]) + b * m[2];
        const float o1 = (r * m[3] + g * m[4]) + b * m[5];
        const float o2 = (r * m[6] + g * m[7]) + b * m[8];
        r = o0; g = o1; b = o2;
    }
};

// 8-segment piecewise-linear curve (isp/filters.py:293-303,337-347): acc = v * 0; acc += clamp(v - i/8, 0, 1/8) * p_i for
// i = 0..7 in order; result acc * 8 / sum(p). For v in segment j the terms i < j are exactly p_i / 8 (a power-of-two scaling),
// the terms i > j are +0 and leave the running sum unchanged, so the reference's eight-term fp32 sum equals
//     P_j + clamp(v - j/8, 0, 1/8) * p_j,      P_j = ((p_0/8 + p_1/8) + ...) + p_{j-1}/8  in the same order  (P_0 = +0)
// bit for bit; `+ v * 0.0f` keeps the reference's NaN for a non-finite v (inf * 0) and changes nothing else. (P_j, p_j, j/8)
// is an 8-entry table per curve in LDS, one ds_read_b128 per value: ~10 VALU instead of ~34 per curve evaluation — the
// fused-pooling walk has too few waves to hide the difference (T 39 -> see profiles/round4_isp_step_ab.txt).
struct CurveTable {
    const float4* t;        // [8] in LDS: {P_j, p_j, j/8, -}
    float scale;
    __device__ __forceinline__ float eval(float v) const {
        int j = __float2int_rd(v * 8.0f);
        j = min(max(j, 0), 7);
        const float4 e = t[j];
        const float tj = fminf(fmaxf(v - e.z, 0.0f), 0.125f) * e.y;
        return ((e.x + tj) + v * 0.0f) * scale;
    }
};
// entry j of the table of the curve whose eight parameters are c[0..7] (stride `cs` floats)
__device__ __forceinline__ float4 curve_entry(const float* c, int cs, int j) {
    float P = 0.0f;
    for (int i = 0; i < j; ++i) P = (i == 0) ? 0.125f * c[0] : P + 0.125f * c[i * cs];
    return make_float4(P, c[j * cs], 0.125f * (float)j, 0.0f);
}

struct OpTone {  // isp/filters.py:337-347
    static constexpr int kCurves = 1;
    CurveTable tab;
    __device__ void init(const float* p, float4* lds) {
        if (threadIdx.x < 8) lds[threadIdx.x] = curve_entry(p, 1, threadIdx.x);
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += p[i];
        tab.t = lds;
        tab.scale = 8.0f / (s + 1e-30f);
    }
    __device__ __forceinline__ void apply(float& r, float& g, float& b) const {
        r = tab.eval(r); g = tab.eval(g); b = tab.eval(b);
    }
};

struct OpColor {  // isp/filters.py:293-303   per-channel curves, params laid out [step][channel]
    static constexpr int kCurves = 3;
    CurveTable tab[3];
    __device__ void init(const float* p, float4* lds) {
        if (threadIdx.x < 24) lds[threadIdx.x] = curve_entry(p + (threadIdx.x >> 3), 3, threadIdx.x & 7);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) s += p[3 * i + ch];
            tab[ch].t = lds + 8 * ch;
            tab[ch].scale = 8.0f / (s + 1e-30f);
        }
    }
    __device__ __forceinline__ void apply(float& r, float& g, float& b) const {
        r = tab[0].eval(r); g = tab[1].eval(g); b = tab[2].eval(b);
    }
};

// ops with a table in LDS build it cooperatively (the op is block-uniform: every thread of the workgroup is here) and meet
// at one barrier; the others only read their parameters
template <class OP, class = void> struct OpSetup {
    static __device__ __forceinline__ void run(OP& op, const float* p) { op.init(p); }
};
template <class OP> struct OpSetup<OP, decltype((void)OP::kCurves)> {
    static __device__ __forceinline__ void run(OP& op, const float* p) {
        __shared__ float4 curve_lds[8 * OP::kCurves];
        op.init(p, curve_lds);
        __syncthreads();
    }
};

__device__ __forceinline__ float lum_27_67_06(float r, float g, float b) {  // isp/filters.py:12-14
    return (0.27f * r + 0.67f * g) + 0.06f * b;
}

struct OpContrast {  // isp/filters.py:415-419
    float p0, q0;
    __device__ void init(const float* p) { p0 = p[0]; q0 = 1.0f - p[0]; }
    __device__ __forceinline__ void apply(float& r, float& g, float& b) const {
        const float L = clamp01(lum_27_67_06(r, g, b));
        const float cl = -cosf(3.14159274101257324f * L) * 0.5f + 0.5f;
        const float den = L + 1e-6f;
        r = q0 * r + p0 * (r / den * cl);
        g = q0 * g + p0 * (g / den * cl);
        b = q0 * b + p0 * (b / den * cl);
    }
};

struct OpWNB {  // isp/filters.py:435-437
    float p0, q0;
    __device__ void init(const float* p) { p0 = p[0]; q0 = 1.0f - p[0]; }
    __device__ __forceinline__ void apply(float& r, float& g, float& b) const {
        const float pl = p0 * lum_27_67_06(r, g, b);
        r = q0 * r + pl; g = q0 * g + pl; b = q0 * b + pl;
    }
};

// torch.remainder for floats: result takes the sign of the divisor.
__device__ __forceinline__ float py_mod(float a, float m) {
    float r = fmodf(a, m);
    if (r != 0.0f && (r < 0.0f)) r += m;   // m > 0 here
    return r;
}

struct OpSatPlus {  // isp/filters.py:546-560 with rgb2hsv :445-478 and hsv2rgb :481-533
    float p0, q0;
    __device__ void init(const float* p) { p0 = p[0]; q0 = 1.0f - p[0]; }
    __device__ __forceinline__ void apply(float& r_, float& g_, float& b_) const {
        const float r = clamp01(r_), g = clamp01(g_), b = clamp01(b_);
        const float mx = fmaxf(fmaxf(r, g), b), mn = fminf(fminf(r, g), b);
        const float d = (mx - mn) + 1e-8f;
        // sequential masked overwrite: B branch, then G, then R (so ties resolve R > G > B), then grey
        float hue = 0.0f;
        if (b == mx) hue = 4.0f + (r - g) / d;
        if (g == mx) hue = 2.0f + (b - r) / d;
        if (r == mx) hue = py_mod((g - b) / d, 6.0f);
        if (mn == mx) hue = 0.0f;
        hue = hue / 6.0f;
        float s = (mx - mn) / (mx + 1e-8f);
        if (mx == 0.0f) s = 0.0f;
        const float v = mx;
        const float es = s + (1.0f - s) * (0.5f - fabsf(0.5f - v)) * 0.8f;
        // hsv2rgb
        const float h = py_mod(hue, 1.0f);
        const float s2 = clamp01(es), v2 = clamp01(v);
        const float h6 = h * 6.0f;
        const float hi = floorf(h6);
        const float f = h6 - hi;
        const float pp = v2 * (1.0f - s2);
        const float qq = v2 * (1.0f - (f * s2));
        const float tt = v2 * (1.0f - ((1.0f - f) * s2));
        float fr = 0.0f, fg = 0.0f, fb = 0.0f;
        if (hi == 0.0f) { fr = v2; fg = tt; fb = pp; }
        else if (hi == 1.0f) { fr = qq; fg = v2; fb = pp; }
        else if (hi == 2.0f) { fr = pp; fg = v2; fb = tt; }
        else if (hi == 3.0f) { fr = pp; fg = qq; fb = v2; }
        else if (hi == 4.0f) { fr = tt; fg = pp; fb = v2; }
        else if (hi == 5.0f) { fr = v2; fg = pp; fb = qq; }
        r_ = r * q0 + fr * p0;
        g_ = g * q0 + fg * p0;
        b_ = b * q0 + fb * p0;
    }
};

// ---- streaming loop -----------------------------------------------------------------------------

template <class OP>
__device__ __forceinline__ void apply4(const OP& op, float4& r, float4& g, float4& b, const Clip& clip) {
    op.apply(r.x, g.x, b.x); op.apply(r.y, g.y, b.y); op.apply(r.z, g.z, b.z); op.apply(r.w, g.w, b.w);
    r.x = clip(r.x); r.y = clip(r.y); r.z = clip(r.z); r.w = clip(r.w);
    g.x = clip(g.x); g.y = clip(g.y); g.z = clip(g.z); g.w = clip(g.w);
    b.x = clip(b.x); b.y = clip(b.y); b.z = clip(b.z); b.w = clip(b.w);
}

template <class OP>
__device__ __forceinline__ void apply1(const OP& op, float& r, float& g, float& b, const Clip& clip) {
    op.apply(r, g, b);
    r = clip(r); g = clip(g); b = clip(b);
}

// n4 = plane elements / 4.  Two 16-B accesses per plane in flight per lane.
template <class OP>
__device__ void stream_vec(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ p,
                           long plane, const Clip clip) {
    OP op;
    OpSetup<OP>::run(op, p);
    const long n4 = plane >> 2;
    const float4* __restrict__ ir = reinterpret_cast<const float4*>(in);
    const float4* __restrict__ ig = reinterpret_cast<const float4*>(in + plane);
    const float4* __restrict__ ib = reinterpret_cast<const float4*>(in + 2 * plane);
    float4* __restrict__ orr = reinterpret_cast<float4*>(out);
    float4* __restrict__ og = reinterpret_cast<float4*>(out + plane);
    float4* __restrict__ ob = reinterpret_cast<float4*>(out + 2 * plane);
    const long stride = (long)gridDim.x * kThreads;
    long i = (long)blockIdx.x * kThreads + threadIdx.x;
    for (; i + stride < n4; i += 2 * stride) {
        float4 r0 = ld4(ir + i), g0 = ld4(ig + i), b0 = ld4(ib + i);
        float4 r1 = ld4(ir + i + stride), g1 = ld4(ig + i + stride), b1 = ld4(ib + i + stride);
        apply4(op, r0, g0, b0, clip);
        apply4(op, r1, g1, b1, clip);
        st4(orr + i, r0); st4(og + i, g0); st4(ob + i, b0);
        st4(orr + i + stride, r1); st4(og + i + stride, g1); st4(ob + i + stride, b1);
    }
    for (; i < n4; i += stride) {
        float4 r0 = ld4(ir + i), g0 = ld4(ig + i), b0 = ld4(ib + i);
        apply4(op, r0, g0, b0, clip);
        st4(orr + i, r0); st4(og + i, g0); st4(ob + i, b0);
    }
}

template <class OP>
__device__ void stream_scalar(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ p,
                              long plane, const Clip clip) {
    OP op;
    OpSetup<OP>::run(op, p);
    const long stride = (long)gridDim.x * kThreads;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < plane; i += stride) {
        float r = in[i], g = in[i + plane], b = in[i + 2 * plane];
        apply1(op, r, g, b, clip);
        out[i] = r; out[i + plane] = g; out[i + 2 * plane] = b;
    }
}

template <bool VEC, class OP>
__device__ __forceinline__ void stream(const float* in, float* out, const float* p, long plane, const Clip clip) {
    if constexpr (VEC) stream_vec<OP>(in, out, p, plane, clip);
    else stream_scalar<OP>(in, out, p, plane, clip);
}

template <bool VEC>
__global__ __launch_bounds__(kThreads) void k_pointwise(const float* __restrict__ img, float* __restrict__ out,
                                                        const int32_t* __restrict__ ids, int uniform_op,
                                                        const float* __restrict__ params, int pstride,
                                                        long plane, unsigned flags) {
    const int b = blockIdx.y;
    int op = ids ? ids[b] : uniform_op;
    // an id no kernel family owns (the ids live on the device: the host cannot reject it) gives the zero image, like -1
    if (!op_is_pointwise(op) && !op_is_conv(op) && op != ADAISP_OP_NLM) op = ADAISP_OP_ZERO;
    const float* in = img + (long)b * 3 * plane;
    float* o = out + (long)b * 3 * plane;
    const float* p = params + (long)b * pstride;
    const Clip clip((flags & ADAISP_CLIP01) != 0);
    switch (op) {
        case ADAISP_OP_ZERO:     stream<VEC, OpZero>(in, o, p, plane, clip); break;
        case ADAISP_OP_EXPOSURE: stream<VEC, OpExposure>(in, o, p, plane, clip); break;
        case ADAISP_OP_GAMMA:    stream<VEC, OpGamma>(in, o, p, plane, clip); break;
        case ADAISP_OP_WB:       stream<VEC, OpWB>(in, o, p, plane, clip); break;
        case ADAISP_OP_CCM:      stream<VEC, OpCCM>(in, o, p, plane, clip); break;
        case ADAISP_OP_TONE:     stream<VEC, OpTone>(in, o, p, plane, clip); break;
        case ADAISP_OP_COLOR:    stream<VEC, OpColor>(in, o, p, plane, clip); break;
        case ADAISP_OP_CONTRAST: stream<VEC, OpContrast>(in, o, p, plane, clip); break;
        case ADAISP_OP_WNB:      stream<VEC, OpWNB>(in, o, p, plane, clip); break;
        case ADAISP_OP_SATPLUS:  stream<VEC, OpSatPlus>(in, o, p, plane, clip); break;
        default: break;  // owned by the stencil / NLM kernels
    }
}

// ---- the same ops with the next step's 64x64 pooling fused (agent.py:97 applied to this step's output) ---------
// Workgroup = one pool row `oy` (image rows win_lo(oy,H) .. win_hi(oy,H)) over the WHOLE width; wave w owns the 256-pixel
// strip [256 w, 256 w + 256) — 1 KB per plane row, aligned to the 128-byte line whenever the row pitch is, so that no line
// is fetched by two waves (round 3 cut the strips along the pool columns: 220 px = 6.9 lines, the edge lines were read
// twice and the strips' non-temporal loads would have fetched them twice from HBM). A lane owns a pixel quad and walks the
// window's rows, ISP_POOL_ROWS rows (x 3 planes x 16 B) in flight; the three planes' column sums stay in registers, go to
// LDS ([3][W] floats), and after one barrier thread (c, ox) adds its pool cell's columns in ascending x — the order of
// k_pool64 (column sums down the rows from 0.0f, then the x-window from 0.0f, / kh / kw): bit-identical results. A row that
// belongs to two windows is read (and computed) by both owners and written by the first.
#ifndef ISP_POOL_ROWS
#define ISP_POOL_ROWS 3
#endif
// Round 6 (measured, OFF): a row that belongs to TWO pool windows (three of every four window boundaries at H = 720: 12-row windows
// on an 11.25-row pitch) is read by both windows' workgroups — 94.4 MB fetched per launch against 88.6 (profiles/round5_pmc_traffic.json).
// ISP_POOL_SHARED_L2 = 1 reads exactly those rows with PLAIN loads (everything else stays non-temporal) and runs consecutive pool
// rows on ONE XCD (workgroup bx of an image sits on XCD bx % 8: pool row oy = (bx % 8) * 8 + bx / 8), hoping the second reader
// finds the row in that XCD's L2. It does not: the fabric counter stays at 183.4 MB per launch and the time at 38.2 vs 38.0 us
// (profiles/round6_isp_pool_shared_l2_ab.txt, round6_pmc_traffic.json) — the two readers are a whole window walk (~30 us of
// streaming through a 4 MB L2) apart. Bit-identical either way.
#ifndef ISP_POOL_SHARED_L2
#define ISP_POOL_SHARED_L2 0
#endif
#ifndef ISP_POOL_PIPE
#define ISP_POOL_PIPE 0
#endif
// The walk's shape per op CLASS (round 5): "light" ops (a handful of VALU per value: E, G, W, CCM, Ct, BW, zero) have the
// registers for the double-buffered walk — the loads of the next row group are in flight while this one is computed and stored
// — the "heavy" ones (T, C: an LDS table look-up per value; S+: HSV round trip) need the occupancy more (round 4's measurement:
// one form for all ops cost E / CCM 3-4 us or T / S+ 10-30). ISP_POOL_{ROWS,PIPE}_{LIGHT,HEAVY} override per class.
#ifndef ISP_POOL_ROWS_LIGHT
#define ISP_POOL_ROWS_LIGHT ISP_POOL_ROWS
#endif
#ifndef ISP_POOL_PIPE_LIGHT
#define ISP_POOL_PIPE_LIGHT ISP_POOL_PIPE
#endif
#ifndef ISP_POOL_ROWS_HEAVY
#define ISP_POOL_ROWS_HEAVY ISP_POOL_ROWS
#endif
#ifndef ISP_POOL_PIPE_HEAVY
#define ISP_POOL_PIPE_HEAVY ISP_POOL_PIPE
#endif
#ifndef ISP_POOL_ROWS_MEDIUM
#define ISP_POOL_ROWS_MEDIUM ISP_POOL_ROWS
#endif
#ifndef ISP_POOL_PIPE_MEDIUM
#define ISP_POOL_PIPE_MEDIUM ISP_POOL_PIPE
#endif
template <class OP> struct PoolWalk { static constexpr int rows = ISP_POOL_ROWS_LIGHT; static constexpr bool pipe = ISP_POOL_PIPE_LIGHT != 0; };
// (CCM holds 15 more registers than exposure — the three-row double buffer spills at the 128-register cap, two rows fit; contrast
// spills in every double-buffered form and keeps the plain walk)
template <> struct PoolWalk<OpCCM> { static constexpr int rows = ISP_POOL_ROWS_MEDIUM; static constexpr bool pipe = ISP_POOL_PIPE_MEDIUM != 0; };
template <> struct PoolWalk<OpContrast> { static constexpr int rows = ISP_POOL_ROWS; static constexpr bool pipe = ISP_POOL_PIPE != 0; };
template <> struct PoolWalk<OpTone> { static constexpr int rows = ISP_POOL_ROWS_HEAVY; static constexpr bool pipe = ISP_POOL_PIPE_HEAVY != 0; };
template <> struct PoolWalk<OpColor> { static constexpr int rows = ISP_POOL_ROWS_HEAVY; static constexpr bool pipe = ISP_POOL_PIPE_HEAVY != 0; };
template <> struct PoolWalk<OpSatPlus> { static constexpr int rows = ISP_POOL_ROWS_HEAVY; static constexpr bool pipe = ISP_POOL_PIPE_HEAVY != 0; };
template <class OP>
__device__ __forceinline__ void stream_pool(const OP& op, const float* __restrict__ in, float* __restrict__ out,
                                            int H, int W, int ys, int ye, int y_own_end,
                                            int x, bool active, const Clip clip, float4 (&acc)[3], int y_shared_top, int y_shared_bot) {
    constexpr int R = PoolWalk<OP>::rows;
    const long plane = (long)H * W;
    const int xs = active ? x : 0;                      // inactive lanes (beyond a ragged right edge) read a valid quad and drop it
    // the loads of the NEXT group of R rows are issued before the current group is computed and stored (two register sets,
    // the loop body written out for both): a wave always has R x 3 x 16 B in flight behind its arithmetic — the workgroup is
    // only `W / 256` waves, 2-3 per SIMD at 720p, too few to hide a whole load -> compute -> store round trip in each other
    auto load = [&](float4 (&r)[R], float4 (&g)[R], float4 (&b)[R], int y0) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int yy = min(y0 + u, ye - 1);
            const long o = (long)yy * W + xs;
            if (ISP_POOL_SHARED_L2 && (yy == y_shared_top || yy == y_shared_bot)) {       // (wave-uniform) the rows two windows share
                r[u] = *reinterpret_cast<const float4*>(in + o);
                g[u] = *reinterpret_cast<const float4*>(in + plane + o);
                b[u] = *reinterpret_cast<const float4*>(in + 2 * plane + o);
            } else {
                r[u] = ld4(reinterpret_cast<const float4*>(in + o));
                g[u] = ld4(reinterpret_cast<const float4*>(in + plane + o));
                b[u] = ld4(reinterpret_cast<const float4*>(in + 2 * plane + o));
            }
        }
    };
    auto work = [&](float4 (&r)[R], float4 (&g)[R], float4 (&b)[R], int y0) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int y = y0 + u;
            if (y < ye) {
                apply4(op, r[u], g[u], b[u], clip);
                if (active) {
                    acc[0].x += r[u].x; acc[0].y += r[u].y; acc[0].z += r[u].z; acc[0].w += r[u].w;
                    acc[1].x += g[u].x; acc[1].y += g[u].y; acc[1].z += g[u].z; acc[1].w += g[u].w;
                    acc[2].x += b[u].x; acc[2].y += b[u].y; acc[2].z += b[u].z; acc[2].w += b[u].w;
                    if (y < y_own_end) {
                        const long o = (long)y * W + x;
                        st4(reinterpret_cast<float4*>(out + o), r[u]);
                        st4(reinterpret_cast<float4*>(out + plane + o), g[u]);
                        st4(reinterpret_cast<float4*>(out + 2 * plane + o), b[u]);
                    }
                }
            }
        }
    };
    if constexpr (PoolWalk<OP>::pipe) {
        float4 ra[R], ga[R], ba[R], rb[R], gb[R], bb[R];
        load(ra, ga, ba, ys);
        for (int y0 = ys; y0 < ye; y0 += 2 * R) {
            if (y0 + R < ye) load(rb, gb, bb, y0 + R);
            work(ra, ga, ba, y0);
            if (y0 + R < ye) {
                if (y0 + 2 * R < ye) load(ra, ga, ba, y0 + 2 * R);
                work(rb, gb, bb, y0 + R);
            }
        }
    } else {
        float4 r[R], g[R], b[R];
        for (int y0 = ys; y0 < ye; y0 += R) {
            load(r, g, b, y0);
            work(r, g, b, y0);
        }
    }
}

// At most 8 waves per workgroup; rows wider than 2048 px give a wave a second 256-px strip (wave, wave + 8, ...).
// OPSEL >= 0: the op is a compile-time constant (the host knows it: adaisp_forward_uniform) — no switch, and the register
// allocation is that op's, not the largest branch's; OPSEL = kOpRuntime: the op is read from the device-side ids.
constexpr int kPoolWaves = 8;
constexpr int kOpRuntime = -2;
#ifndef ISP_POOL_WPE
#define ISP_POOL_WPE 4          // waves per SIMD the register allocation must allow (4 -> 128 registers per lane)
#endif

template <class OP>
__device__ __forceinline__ void pool_strips(const float* __restrict__ in, float* __restrict__ o, const float* __restrict__ p,
                                            float* __restrict__ colsum, int H, int W, int ys, int ye, int y_own_end,
                                            const Clip clip, int y_shared_top, int y_shared_bot) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = (int)blockDim.x >> 6;
    const int strips = (W + 255) >> 8, Wp = strips << 8;
    OP op;
    OpSetup<OP>::run(op, p);
    for (int sw = wave; sw < strips; sw += nwaves) {
        const int x = 256 * sw + 4 * lane;
        float4 acc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        stream_pool<OP>(op, in, o, H, W, ys, ye, y_own_end, x, x < W, clip, acc, y_shared_top, y_shared_bot);
#pragma unroll
        for (int c = 0; c < 3; ++c) *reinterpret_cast<float4*>(colsum + c * Wp + x) = acc[c];
    }
}

template <int OPSEL>
__global__ __launch_bounds__(64 * kPoolWaves, ISP_POOL_WPE)
void k_pointwise_pool(const float* __restrict__ img, float* __restrict__ out, float* __restrict__ pooled,
                      const int32_t* __restrict__ ids, int uniform_op, const float* __restrict__ params, int pstride,
                      int H, int W, unsigned flags) {
    extern __shared__ __attribute__((aligned(16))) float colsum[];      // [3][Wp], Wp = 256 x strips
#if ISP_POOL_SHARED_L2
    const int oy = ((int)blockIdx.x & 7) * 8 + ((int)blockIdx.x >> 3), b = blockIdx.y;      // consecutive pool rows on one XCD
#else
    const int oy = blockIdx.x, b = blockIdx.y;
#endif
    int op = OPSEL != kOpRuntime ? OPSEL : (ids ? ids[b] : uniform_op);
    if (op_is_conv(op) || op == ADAISP_OP_NLM) return;                   // block-uniform: the whole workgroup leaves
    if (!op_is_pointwise(op)) op = ADAISP_OP_ZERO;                       // unknown id (device data): the zero image, like -1
    const int Wp = ((W + 255) >> 8) << 8;
    const long plane = (long)H * W;
    const float* in = img + (long)b * 3 * plane;
    float* o = out + (long)b * 3 * plane;
    const float* p = params + (long)b * pstride;
    const Clip clip((flags & ADAISP_CLIP01) != 0);
    const int ys = win_lo(oy, H), ye = win_hi(oy, H);
    const int y_own_end = oy == 63 ? H : win_lo(oy + 1, H);
    // the rows this window shares with its neighbours (-1: none): its first row if the window above ends behind it, its last row
    // if the window below starts on it
    const int y_shared_top = (oy > 0 && win_hi(oy - 1, H) > ys) ? ys : -1;
    const int y_shared_bot = (oy < 63 && win_lo(oy + 1, H) < ye) ? ye - 1 : -1;
#define POOL_CASE(ID, OP) case ID: if (OPSEL == kOpRuntime || OPSEL == ID) pool_strips<OP>(in, o, p, colsum, H, W, ys, ye, y_own_end, clip, y_shared_top, y_shared_bot); break;
    switch (op) {
        POOL_CASE(ADAISP_OP_ZERO, OpZero) POOL_CASE(ADAISP_OP_EXPOSURE, OpExposure) POOL_CASE(ADAISP_OP_GAMMA, OpGamma)
        POOL_CASE(ADAISP_OP_WB, OpWB) POOL_CASE(ADAISP_OP_CCM, OpCCM) POOL_CASE(ADAISP_OP_TONE, OpTone)
        POOL_CASE(ADAISP_OP_COLOR, OpColor) POOL_CASE(ADAISP_OP_CONTRAST, OpContrast) POOL_CASE(ADAISP_OP_WNB, OpWNB)
        POOL_CASE(ADAISP_OP_SATPLUS, OpSatPlus)
        default: break;
    }
#undef POOL_CASE
    __syncthreads();
    const float kh = (float)(ye - ys);
    for (int i = threadIdx.x; i < 3 * 64; i += blockDim.x) {
        const int c = i >> 6, ox = i & 63;
        const int xs = win_lo(ox, W), xe = win_hi(ox, W);
        const float a = window_sum(colsum + c * Wp, xs, xe);
        pooled[(((long)b * 3 + c) * 64 + oy) * 64 + ox] = a / kh / (float)(xe - xs);
    }
}

}  // namespace

hipError_t launch_pointwise_pool(const Batch& a, float* pooled, const PoolGeom& g, hipStream_t s) {
    (void)g;                                                            // (the stencil family's cut; this one's is W / 256)
    const int strips = (a.W + 255) / 256;
    const int waves = strips < kPoolWaves ? strips : kPoolWaves;
    const dim3 grid(64, (unsigned)a.B), block(64 * waves);
    const size_t smem = (size_t)strips * 3 * 256 * sizeof(float);
#define POOL_LAUNCH(SEL) hipLaunchKernelGGL(k_pointwise_pool<SEL>, grid, block, smem, s, a.img, a.out, pooled, a.ids, a.uniform_op, \
                                            a.params, a.pstride, a.H, a.W, a.flags)
    if (a.ids) POOL_LAUNCH(kOpRuntime);
    else switch (a.uniform_op) {
        case ADAISP_OP_EXPOSURE: POOL_LAUNCH(ADAISP_OP_EXPOSURE); break;
        case ADAISP_OP_GAMMA:    POOL_LAUNCH(ADAISP_OP_GAMMA); break;
        case ADAISP_OP_WB:       POOL_LAUNCH(ADAISP_OP_WB); break;
        case ADAISP_OP_CCM:      POOL_LAUNCH(ADAISP_OP_CCM); break;
        case ADAISP_OP_TONE:     POOL_LAUNCH(ADAISP_OP_TONE); break;
        case ADAISP_OP_COLOR:    POOL_LAUNCH(ADAISP_OP_COLOR); break;
        case ADAISP_OP_CONTRAST: POOL_LAUNCH(ADAISP_OP_CONTRAST); break;
        case ADAISP_OP_WNB:      POOL_LAUNCH(ADAISP_OP_WNB); break;
        case ADAISP_OP_SATPLUS:  POOL_LAUNCH(ADAISP_OP_SATPLUS); break;
        default:                 POOL_LAUNCH(kOpRuntime); break;         // ZERO and everything the kernel maps to it
    }
#undef POOL_LAUNCH
    return hipGetLastError();
}

hipError_t launch_pointwise(const Batch& a, hipStream_t s) {
    const long plane = (long)a.H * a.W;
    const bool vec = (plane % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.img) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(a.out) & 15) == 0);
    const long items = vec ? plane / 4 : plane;
    long bx = (items + kThreads - 1) / kThreads;
    // two items per lane per trip; cap the grid at ~8k workgroups (>> 256 CUs) and grid-stride the rest
    bx = (bx + 1) / 2;
    const long cap = 8192 / (a.B > 0 ? a.B : 1);
    if (bx > cap) bx = cap > 0 ? cap : 1;
    if (bx < 1) bx = 1;
    dim3 grid((unsigned)bx, (unsigned)a.B);
    if (vec)
        hipLaunchKernelGGL(k_pointwise<true>, grid, dim3(kThreads), 0, s, a.img, a.out, a.ids, a.uniform_op,
                           a.params, a.pstride, plane, a.flags);
    else
        hipLaunchKernelGGL(k_pointwise<false>, grid, dim3(kThreads), 0, s, a.img, a.out, a.ids, a.uniform_op,
                           a.params, a.pstride, plane, a.flags);
    return hipGetLastError();
}

}  // namespace adaisp
