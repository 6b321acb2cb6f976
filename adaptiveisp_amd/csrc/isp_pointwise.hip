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

// 8-segment piecewise-linear curve: acc += clamp(v - i/8, 0, 1/8) * p_i in segment order, then * 8/sum.
__device__ __forceinline__ float curve8(float v, const float* c, float scale) {
    float acc = v * 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += fminf(fmaxf(v - 0.125f * (float)i, 0.0f), 0.125f) * c[i];
    return acc * scale;
}

struct OpTone {  // isp/filters.py:337-347
    float c[8], scale;
    __device__ void init(const float* p) {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { c[i] = p[i]; s += p[i]; }
        scale = 8.0f / (s + 1e-30f);
    }
    __device__ __forceinline__ void apply(float& r, float& g, float& b) const {
        r = curve8(r, c, scale); g = curve8(g, c, scale); b = curve8(b, c, scale);
    }
};

struct OpColor {  // isp/filters.py:293-303   per-channel curves, params laid out [step][channel]
    float c[3][8], scale[3];
    __device__ void init(const float* p) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { c[ch][i] = p[3 * i + ch]; s += p[3 * i + ch]; }
            scale[ch] = 8.0f / (s + 1e-30f);
        }
    }
    __device__ __forceinline__ void apply(float& r, float& g, float& b) const {
        r = curve8(r, c[0], scale[0]); g = curve8(g, c[1], scale[1]); b = curve8(b, c[2], scale[2]);
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
    op.init(p);
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
        float4 r0 = ir[i], g0 = ig[i], b0 = ib[i];
        float4 r1 = ir[i + stride], g1 = ig[i + stride], b1 = ib[i + stride];
        apply4(op, r0, g0, b0, clip);
        apply4(op, r1, g1, b1, clip);
        orr[i] = r0; og[i] = g0; ob[i] = b0;
        orr[i + stride] = r1; og[i + stride] = g1; ob[i + stride] = b1;
    }
    for (; i < n4; i += stride) {
        float4 r0 = ir[i], g0 = ig[i], b0 = ib[i];
        apply4(op, r0, g0, b0, clip);
        orr[i] = r0; og[i] = g0; ob[i] = b0;
    }
}

template <class OP>
__device__ void stream_scalar(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ p,
                              long plane, const Clip clip) {
    OP op;
    op.init(p);
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
// One wave = one strip of pool columns x one pool row (isp_internal.h: PoolGeom); the lane owns a pixel quad and walks
// the window's rows, four rows (12 x 16-B loads) in flight; the three planes' column sums stay in registers.
template <class OP>
__device__ __forceinline__ void stream_pool(const float* __restrict__ in, float* __restrict__ out,
                                            const float* __restrict__ p, int H, int W, int ys, int ye, int y_own_end,
                                            int x, bool active, bool own_x, const Clip clip, float4 (&acc)[3]) {
    OP op;
    op.init(p);
    const long plane = (long)H * W;
    const int xs = active ? x : 0;                      // inactive lanes read a valid quad and drop it
    for (int y0 = ys; y0 < ye; y0 += 4) {
        float4 r[4], g[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long o = (long)min(y0 + u, ye - 1) * W + xs;
            r[u] = *reinterpret_cast<const float4*>(in + o);
            g[u] = *reinterpret_cast<const float4*>(in + plane + o);
            b[u] = *reinterpret_cast<const float4*>(in + 2 * plane + o);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int y = y0 + u;
            if (y < ye) {
                apply4(op, r[u], g[u], b[u], clip);
                if (active) {
                    acc[0].x += r[u].x; acc[0].y += r[u].y; acc[0].z += r[u].z; acc[0].w += r[u].w;
                    acc[1].x += g[u].x; acc[1].y += g[u].y; acc[1].z += g[u].z; acc[1].w += g[u].w;
                    acc[2].x += b[u].x; acc[2].y += b[u].y; acc[2].z += b[u].z; acc[2].w += b[u].w;
                    if (own_x && y < y_own_end) {
                        const long o = (long)y * W + x;
                        *reinterpret_cast<float4*>(out + o) = r[u];
                        *reinterpret_cast<float4*>(out + plane + o) = g[u];
                        *reinterpret_cast<float4*>(out + 2 * plane + o) = b[u];
                    }
                }
            }
        }
    }
}

__global__ __launch_bounds__(1024) void k_pointwise_pool(const float* __restrict__ img, float* __restrict__ out,
                                                         float* __restrict__ pooled, const int32_t* __restrict__ ids,
                                                         int uniform_op, const float* __restrict__ params, int pstride,
                                                         int H, int W, int cps, unsigned flags) {
    extern __shared__ __attribute__((aligned(16))) float colsum[];      // [waves][3][256]
    const int oy = blockIdx.x, b = blockIdx.y;
    int op = ids ? ids[b] : uniform_op;
    if (op_is_conv(op) || op == ADAISP_OP_NLM) return;                   // block-uniform: the whole workgroup leaves
    if (!op_is_pointwise(op)) op = ADAISP_OP_ZERO;                       // unknown id (device data): the zero image, like -1
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long plane = (long)H * W;
    const float* in = img + (long)b * 3 * plane;
    float* o = out + (long)b * 3 * plane;
    const float* p = params + (long)b * pstride;
    const Clip clip((flags & ADAISP_CLIP01) != 0);
    const int ys = win_lo(oy, H), ye = win_hi(oy, H);
    const int y_own_end = oy == 63 ? H : win_lo(oy + 1, H);
    const int x_lo = strip_x_lo(wave, cps, W), x_end = strip_x_end(wave, cps, W);
    const int x = x_lo + 4 * lane;
    const bool active = x < x_end, own_x = x < strip_x_lo(wave + 1, cps, W);
    float4 acc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    switch (op) {
        case ADAISP_OP_ZERO:     stream_pool<OpZero>(in, o, p, H, W, ys, ye, y_own_end, x, active, own_x, clip, acc); break;
        case ADAISP_OP_EXPOSURE: stream_pool<OpExposure>(in, o, p, H, W, ys, ye, y_own_end, x, active, own_x, clip, acc); break;
        case ADAISP_OP_GAMMA:    stream_pool<OpGamma>(in, o, p, H, W, ys, ye, y_own_end, x, active, own_x, clip, acc); break;
        case ADAISP_OP_WB:       stream_pool<OpWB>(in, o, p, H, W, ys, ye, y_own_end, x, active, own_x, clip, acc); break;
        case ADAISP_OP_CCM:      stream_pool<OpCCM>(in, o, p, H, W, ys, ye, y_own_end, x, active, own_x, clip, acc); break;
        case ADAISP_OP_TONE:     stream_pool<OpTone>(in, o, p, H, W, ys, ye, y_own_end, x, active, own_x, clip, acc); break;
        case ADAISP_OP_COLOR:    stream_pool<OpColor>(in, o, p, H, W, ys, ye, y_own_end, x, active, own_x, clip, acc); break;
        case ADAISP_OP_CONTRAST: stream_pool<OpContrast>(in, o, p, H, W, ys, ye, y_own_end, x, active, own_x, clip, acc); break;
        case ADAISP_OP_WNB:      stream_pool<OpWNB>(in, o, p, H, W, ys, ye, y_own_end, x, active, own_x, clip, acc); break;
        case ADAISP_OP_SATPLUS:  stream_pool<OpSatPlus>(in, o, p, H, W, ys, ye, y_own_end, x, active, own_x, clip, acc); break;
        default: break;
    }
    float* cs = colsum + wave * 3 * 256;
#pragma unroll
    for (int c = 0; c < 3; ++c) *reinterpret_cast<float4*>(cs + c * 256 + 4 * lane) = acc[c];
    __syncthreads();
    const int c0 = strip_cell0(wave, cps), ncell = strip_cell0(wave + 1, cps) - c0;
    const float kh = (float)(ye - ys);
    for (int i = lane; i < 3 * ncell; i += 64) {
        const int c = i / ncell, ox = c0 + (i - c * ncell);
        const int xs = win_lo(ox, W), xe = win_hi(ox, W);
        const float* col = cs + c * 256 - x_lo;
        float a = 0.f;
        for (int xx = xs; xx < xe; ++xx) a += col[xx];
        pooled[(((long)b * 3 + c) * 64 + oy) * 64 + ox] = a / kh / (float)(xe - xs);
    }
}

}  // namespace

hipError_t launch_pointwise_pool(const Batch& a, float* pooled, const PoolGeom& g, hipStream_t s) {
    hipLaunchKernelGGL(k_pointwise_pool, dim3(64, (unsigned)a.B), dim3(64 * g.strips), (size_t)g.strips * 3 * 256 * sizeof(float),
                       s, a.img, a.out, pooled, a.ids, a.uniform_op, a.params, a.pstride, a.H, a.W, g.cps, a.flags);
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
