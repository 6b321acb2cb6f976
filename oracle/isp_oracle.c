/*
 * isp_oracle.c — CPU restatement of the reference ISP filter stack.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker for the HIP path: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it. The product (adaptiveisp_amd/) never does.
 *
 * It restates, per pixel and in plain scalar C, the algorithm of the reference's Python/ATen op
 * chains (paths relative to the reference checkout, OpenImagingLab/AdaptiveISP):
 *     isp/filters.py   (Filter.forward, the 10 default filters, Color/USM/SharpenV2 variants)
 *     isp/denoise.py   (NonLocalMeansGray 11/5, BoxFilter, rgb_to_luminance)
 *     isp/sharpen.py   (adjust_sharpness, sharpness, unsharp_mask, gaussian kernels)
 *     agent.py         (pdf_sample, one_hot select, state update), AdaptiveAvgPool2d((64,64))
 * Parity pin: tests/golden/ *.npz were produced by importing the reference itself in the build
 * container (tests/golden/gen_golden.py); tests/test_oracle_golden.py checks this file against them.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp). All images planar [B,3,H,W] fp32.
 * Op codes are those of include/adaisp.h.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define OP_ZERO (-1)
#define OP_EXPOSURE 0
#define OP_GAMMA 1
#define OP_CCM 2
#define OP_SHARPEN 3
#define OP_NLM 4
#define OP_TONE 5
#define OP_CONTRAST 6
#define OP_SATPLUS 7
#define OP_WNB 8
#define OP_WB 9
#define OP_USM 10
#define OP_SHARPEN_V2 11
#define OP_COLOR 12

static inline float clamp01f(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }
static inline float clipf(float v, int clip) { return clip ? clamp01f(v) : v; }

/* torch.remainder on floats (sign follows the divisor; divisor > 0 here). */
static inline float py_modf(float a, float m) {
    float r = fmodf(a, m);
    if (r != 0.0f && r < 0.0f) r += m;
    return r;
}

/* rgb2lum, isp/filters.py:12-14 */
static inline float lum_filters(float r, float g, float b) { return (0.27f * r + 0.67f * g) + 0.06f * b; }

/* 8-step piecewise-linear curve, isp/filters.py:341-345 (tone) and :298-302 (colour) */
static inline float curve8(float v, const float* c, int cstride, float scale) {
    float acc = v * 0.0f;
    for (int i = 0; i < 8; ++i) {
        float t = v - 0.125f * (float)i;
        t = t < 0.0f ? 0.0f : (t > 0.125f ? 0.125f : t);
        acc += t * c[i * cstride];
    }
    return acc * scale;
}

/* SaturationPlusFilter.process with rgb2hsv / hsv2rgb, isp/filters.py:546-560, :445-478, :481-533 */
static void satplus_px(float* r_, float* g_, float* b_, float p) {
    const float r = clamp01f(*r_), g = clamp01f(*g_), b = clamp01f(*b_);
    float mx = r > g ? r : g; mx = mx > b ? mx : b;
    float mn = r < g ? r : g; mn = mn < b ? mn : b;
    const float d = (mx - mn) + 1e-8f;
    float hue = 0.0f;
    if (b == mx) hue = 4.0f + (r - g) / d;          /* :456 */
    if (g == mx) hue = 2.0f + (b - r) / d;          /* :459 */
    if (r == mx) hue = py_modf((g - b) / d, 6.0f);  /* :462-464 */
    if (mn == mx) hue = 0.0f;                       /* :466 */
    hue = hue / 6.0f;
    float s = (mx - mn) / (mx + 1e-8f);
    if (mx == 0.0f) s = 0.0f;
    const float v = mx;
    const float es = s + (1.0f - s) * (0.5f - fabsf(0.5f - v)) * 0.8f; /* :552 */
    const float h = py_modf(hue, 1.0f);
    const float s2 = clamp01f(es), v2 = clamp01f(v);
    const float h6 = h * 6.0f, hi = floorf(h6), f = h6 - hi;
    const float pp = v2 * (1.0f - s2), qq = v2 * (1.0f - (f * s2)), tt = v2 * (1.0f - ((1.0f - f) * s2));
    float fr = 0.0f, fg = 0.0f, fb = 0.0f;
    if (hi == 0.0f) { fr = v2; fg = tt; fb = pp; }
    else if (hi == 1.0f) { fr = qq; fg = v2; fb = pp; }
    else if (hi == 2.0f) { fr = pp; fg = v2; fb = tt; }
    else if (hi == 3.0f) { fr = pp; fg = qq; fb = v2; }
    else if (hi == 4.0f) { fr = tt; fg = pp; fb = v2; }
    else if (hi == 5.0f) { fr = v2; fg = pp; fb = qq; }
    const float q = 1.0f - p;
    *r_ = r * q + fr * p; *g_ = g * q + fg * p; *b_ = b * q + fb * p;   /* :560 */
}

static void pointwise_image(int op, const float* in, float* out, const float* p, long plane, int clip) {
    float c[27];
    memset(c, 0, sizeof c);
    switch (op) {
        case OP_EXPOSURE: c[0] = expf(p[0] * 0.6931471805599453f); break;                 /* :224 */
        case OP_CCM:                                                                       /* :706-707 */
            for (int i = 0; i < 3; ++i) {
                const float rs = (p[3 * i] + p[3 * i + 1]) + p[3 * i + 2];
                for (int j = 0; j < 3; ++j) c[3 * i + j] = p[3 * i + j] / rs;
            }
            break;
        case OP_TONE: {                                                                    /* :340,345 */
            float s = 0.0f;
            for (int i = 0; i < 8; ++i) s += p[i];
            c[0] = 8.0f / (s + 1e-30f);
        } break;
        case OP_COLOR:                                                                     /* :297,302 */
            for (int ch = 0; ch < 3; ++ch) {
                float s = 0.0f;
                for (int i = 0; i < 8; ++i) s += p[3 * i + ch];
                c[ch] = 8.0f / (s + 1e-30f);
            }
            break;
        default: break;
    }
#pragma omp parallel for schedule(static)
    for (long i = 0; i < plane; ++i) {
        float r = in[i], g = in[i + plane], b = in[i + 2 * plane];
        switch (op) {
            case OP_ZERO: r = g = b = 0.0f; break;
            case OP_EXPOSURE: r *= c[0]; g *= c[0]; b *= c[0]; break;
            case OP_GAMMA:                                                                 /* :245 */
                r = powf(r > 0.001f ? r : 0.001f, p[0]);
                g = powf(g > 0.001f ? g : 0.001f, p[0]);
                b = powf(b > 0.001f ? b : 0.001f, p[0]);
                break;
            case OP_WB: r *= p[0]; g *= p[1]; b *= p[2]; break;                            /* :272 */
            case OP_CCM: {                                                                 /* :671 */
                const float o0 = (r * c[0] + g * c[1]) + b * c[2];
                const float o1 = (r * c[3] + g * c[4]) + b * c[5];
                const float o2 = (r * c[6] + g * c[7]) + b * c[8];
                r = o0; g = o1; b = o2;
            } break;
            case OP_TONE:
                r = curve8(r, p, 1, c[0]); g = curve8(g, p, 1, c[0]); b = curve8(b, p, 1, c[0]);
                break;
            case OP_COLOR:
                r = curve8(r, p + 0, 3, c[0]); g = curve8(g, p + 1, 3, c[1]); b = curve8(b, p + 2, 3, c[2]);
                break;
            case OP_CONTRAST: {                                                            /* :416-419 */
                const float L = clamp01f(lum_filters(r, g, b));
                const float cl = -cosf(3.14159274101257324f * L) * 0.5f + 0.5f;
                const float den = L + 1e-6f, q = 1.0f - p[0];
                r = q * r + p[0] * (r / den * cl);
                g = q * g + p[0] * (g / den * cl);
                b = q * b + p[0] * (b / den * cl);
            } break;
            case OP_WNB: {                                                                 /* :436-437 */
                const float pl = p[0] * lum_filters(r, g, b), q = 1.0f - p[0];
                r = q * r + pl; g = q * g + pl; b = q * b + pl;
            } break;
            case OP_SATPLUS: satplus_px(&r, &g, &b, p[0]); break;
            default: break;
        }
        out[i] = clipf(r, clip); out[i + plane] = clipf(g, clip); out[i + 2 * plane] = clipf(b, clip);
    }
}

/* torch 'reflect' pad index */
static inline int reflect_idx(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

/* adjust_sharpness (mode 0, isp/sharpen.py:105-142), sharpness (mode 1, :145-182), unsharp_mask (mode 2, :84-102) */
static void conv_image(int op, const float* in, float* out, const float* p, int H, int W) {
    const long plane = (long)H * W;
    float w5[5][5];
    if (op == OP_USM) {
        float g1[5], s = 0.0f;
        for (int i = 0; i < 5; ++i) {            /* _get_gaussian_kernel1d, sharpen.py:15-23 */
            const float t = (float)(i - 2) / p[0];
            g1[i] = expf(-0.5f * (t * t));
            s += g1[i];
        }
        for (int i = 0; i < 5; ++i) g1[i] = g1[i] / s;
        for (int i = 0; i < 5; ++i)
            for (int j = 0; j < 5; ++j) w5[i][j] = g1[i] * g1[j];   /* torch.mm of the 1-D kernels, :30 */
    }
    const float a = 1.0f / 13.0f, c5 = 5.0f / 13.0f;                /* sharpen.py:118-120 */
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y)
        for (int c = 0; c < 3; ++c)
            for (int x = 0; x < W; ++x) {
                const float* src = in + c * plane;
                const float ctr = src[(long)y * W + x];
                float blur = 0.0f, r;
                if (op == OP_USM) {
                    for (int i = 0; i < 5; ++i)
                        for (int j = 0; j < 5; ++j)
                            blur = fmaf(w5[i][j], src[(long)reflect_idx(y + i - 2, H) * W + reflect_idx(x + j - 2, W)], blur);
                    r = ctr + (ctr - blur) * p[1];
                } else {
                    if (y == 0 || y == H - 1 || x == 0 || x == W - 1) {
                        blur = ctr;                                  /* sharpen.py:133-138 */
                    } else {
                        for (int i = 0; i < 3; ++i)
                            for (int j = 0; j < 3; ++j)
                                blur = fmaf((i == 1 && j == 1) ? c5 : a, src[(long)(y + i - 1) * W + (x + j - 1)], blur);
                    }
                    r = (op == OP_SHARPEN) ? ctr * p[0] + blur * (1.0f - p[0]) : ctr + (ctr - blur) * p[0];
                }
                out[c * plane + (long)y * W + x] = clamp01f(r);
            }
}

static inline int wrapi(int v, int n) { v %= n; return v < 0 ? v + n : v; }

/* NonLocalMeansGray(2 sr + 1, 2 pr + 1), isp/denoise.py:93-119. clip_rgb: the colours are clipped first (DenoiseFilter.process,
 * isp/filters.py:584, hands the class a clipped image); without it they are the input as given and only the luminance is
 * computed from the clipped image (rgb_to_luminance, denoise.py:11-17) — the class on its own. */
static void nlm_image_sp(const float* in, float* out, float h, int H, int W, int sr, int pr, int clip_rgb) {
    const long plane = (long)H * W;
    float* rgb = (float*)malloc(sizeof(float) * 3 * plane);
    float* y = (float*)malloc(sizeof(float) * plane);
    for (long i = 0; i < plane; ++i) {
        const float r = clamp01f(in[i]), g = clamp01f(in[i + plane]), b = clamp01f(in[i + 2 * plane]);
        rgb[i] = clip_rgb ? r : in[i]; rgb[i + plane] = clip_rgb ? g : in[i + plane]; rgb[i + 2 * plane] = clip_rgb ? b : in[i + 2 * plane];
        y[i] = (0.299f * r + 0.587f * g) + 0.114f * b;               /* denoise.py:17 */
    }
    const float hh = (h > 0.0f ? h : 0.0f) + 1e-8f;                  /* denoise.py:113 */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j) {
            float num0 = 0.0f, num1 = 0.0f, num2 = 0.0f, den = 0.0f;
            for (int dx = -sr; dx <= sr; ++dx)                       /* x_shift outer, :104 */
                for (int dy = -sr; dy <= sr; ++dy) {                 /* y_shift inner, :105 */
                    float D = 0.0f;
                    for (int bx = -pr; bx <= pr; ++bx)               /* BoxFilter: x outer, y inner, :60-63 */
                        for (int by = -pr; by <= pr; ++by) {
                            const int ii = wrapi(i - by, H), jj = wrapi(j - bx, W);
                            const float d = y[(long)ii * W + jj] - y[(long)wrapi(ii - dy, H) * W + wrapi(jj - dx, W)];
                            D += d * d;
                        }
                    const float dist = sqrtf(D > 0.0f ? D : 0.0f);
                    const float wgt = expf(-dist / hh);
                    const long s = (long)wrapi(i - dy, H) * W + wrapi(j - dx, W);
                    num0 += rgb[s] * wgt; num1 += rgb[s + plane] * wgt; num2 += rgb[s + 2 * plane] * wgt;
                    den += wgt;
                }
            const long o = (long)i * W + j;
            out[o] = clamp01f(num0 / den); out[o + plane] = clamp01f(num1 / den); out[o + 2 * plane] = clamp01f(num2 / den);
        }
    free(rgb); free(y);
}
static void nlm_image(const float* in, float* out, float h, int H, int W) { nlm_image_sp(in, out, h, H, W, 5, 2, 1); }

/* NonLocalMeansGray(search, patch).forward(img, h): the counterpart of adaisp_nlm_general (include/adaisp.h) */
int oracle_nlm_general(const float* img, float* out, const float* h, int B, int H, int W, int search, int patch) {
    if (!img || !out || !h || B <= 0 || H <= 0 || W <= 0 || search < 1 || patch < 1 || !(search & 1) || !(patch & 1)) return -1;
    for (int b = 0; b < B; ++b)
        nlm_image_sp(img + (long)b * 3 * H * W, out + (long)b * 3 * H * W, h[b], H, W, search / 2, patch / 2, 0);
    return 0;
}

/* ---- exported entry points (host pointers; same argument meaning as include/adaisp.h) ------------- */

int oracle_num_params(int op) {
    switch (op) {
        case OP_ZERO: return 0;
        case OP_EXPOSURE: case OP_GAMMA: case OP_SHARPEN: case OP_NLM: case OP_CONTRAST: case OP_SATPLUS:
        case OP_WNB: case OP_SHARPEN_V2: return 1;
        case OP_USM: return 2;
        case OP_WB: return 3;
        case OP_TONE: return 8;
        case OP_CCM: return 9;
        case OP_COLOR: return 24;
        default: return -1;
    }
}

/* Filter.forward's image path: clip(process(img, p), 0, 1) when flags&1, plain process otherwise. */
int oracle_forward(const float* img, float* out, const int32_t* filter_id, const float* params, int pstride,
                   int B, int H, int W, unsigned flags) {
    const long plane = (long)H * W;
    for (int b = 0; b < B; ++b) {
        const int op = filter_id[b];
        const float* in = img + (long)b * 3 * plane;
        float* o = out + (long)b * 3 * plane;
        const float* p = params + (long)b * pstride;
        if (oracle_num_params(op) < 0) return -2;
        if (op == OP_NLM) {
            nlm_image(in, o, p[0], H, W);           /* already in [0,1]: the clip is the identity */
        } else if (op == OP_SHARPEN || op == OP_SHARPEN_V2 || op == OP_USM) {
            if (H < 3 || W < 3) return -4;
            conv_image(op, in, o, p, H, W);
        } else {
            pointwise_image(op, in, o, p, plane, (flags & 1u) != 0);
        }
    }
    return 0;
}

/* AdaptiveAvgPool2d((64,64)): window [floor(o*n/64), ceil((o+1)*n/64)), row-major sum, / kh / kw. */
int oracle_pool64(const float* img, float* pooled, int B, int H, int W) {
    for (int bc = 0; bc < B * 3; ++bc) {
        const float* src = img + (long)bc * H * W;
        for (int oy = 0; oy < 64; ++oy) {
            const int ys = (int)(((long)oy * H) / 64), ye = (int)((((long)oy + 1) * H + 63) / 64);
            for (int ox = 0; ox < 64; ++ox) {
                const int xs = (int)(((long)ox * W) / 64), xe = (int)((((long)ox + 1) * W + 63) / 64);
                float s = 0.0f;
                for (int y = ys; y < ye; ++y)
                    for (int x = xs; x < xe; ++x) s += src[(long)y * W + x];
                pooled[((long)bc * 64 + oy) * 64 + ox] = s / (float)(ye - ys) / (float)(xe - xs);
            }
        }
    }
    return 0;
}

/*
 * Integer stages of Agent.forward (bit-exact contract), agent.py:12-16 (pdf_sample), :138-149
 * (selection / one_hot), :234-259 (state update). pdf [B,F] fp32 (already exploration-mixed and
 * renormalised), u [B] selection noise, states [B,3+F] -> selected [B] int32, new_states [B,3+F].
 * mode: 1 = train (sample), 0 = eval (argmax), forced >= 0 overrides both.
 */
int oracle_select_and_update(const float* pdf, const float* u, const float* states, int B, int F, int mode,
                             int forced, float test_steps, int32_t* selected, float* new_states) {
    for (int b = 0; b < B; ++b) {
        const float* p = pdf + (long)b * F;
        float tot = 0.0f;
        for (int k = 0; k < F; ++k) tot += p[k];
        tot += 1e-36f;
        /* cdf_exclusive_k < u counted over k, minus 1 */
        int cnt = 0;
        float run = 0.0f;
        for (int k = 0; k < F; ++k) {
            const float pk = p[k] / tot;
            run += pk;                     /* cumsum */
            if (run - pk < u[b]) ++cnt;    /* cdf - pdf < noise */
        }
        const int rnd = cnt - 1;
        int amax = 0;
        for (int k = 1; k < F; ++k)
            if (p[k] > p[amax]) amax = k;
        const int sel = forced >= 0 ? forced : (mode ? rnd : amax);
        selected[b] = sel;
        const float* s = states + (long)b * (3 + F);
        float* ns = new_states + (long)b * (3 + F);
        const float last = fabsf(s[2] + 1.0f - test_steps) < 1e-4f ? 1.0f : 0.0f;
        ns[0] = last; ns[1] = last; ns[2] = s[2] + 1.0f;
        for (int k = 0; k < F; ++k) {
            const float oh = (k == sel) ? 1.0f : 0.0f;
            ns[3 + k] = s[3 + k] > oh ? s[3 + k] : oh;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------
 * Greedy IoU NMS — restates `torchvision.ops.nms(boxes, scores, iou_thres)` as the reference calls it at
 * yolov3/utils/general.py:949 on boxes already sorted by descending score (general.py:943) and offset by class
 * (general.py:946-947). torchvision (pinned 0.15.2, requirements.txt:135) is NOT vendored by the reference and
 * not installed here: this function is written from its published algorithm (keep a box unless an earlier kept
 * box has IoU > thr; IoU = inter / (area_a + area_b - inter), intersection extents clamped at 0) and is
 * PARITY UNPINNED against torchvision itself; everything around it (candidate selection, multi-label expansion,
 * class offsets, max_det) is pinned by tests/golden/evalharness.npz generated from the reference's own
 * non_max_suppression.
 * Returns the number of kept boxes (<= max_det); keep[] receives their indices in score order.
 * ------------------------------------------------------------------------------------------------------------ */
int oracle_nms(const float* boxes, int n, float thr, int max_det, int32_t* keep) {
    int count = 0;
    unsigned char* removed = (unsigned char*)calloc((size_t)(n > 0 ? n : 1), 1);
    if (!removed) return -1;
    for (int i = 0; i < n && count < max_det; ++i) {
        if (removed[i]) continue;
        keep[count++] = i;
        const float ax1 = boxes[4 * i], ay1 = boxes[4 * i + 1], ax2 = boxes[4 * i + 2], ay2 = boxes[4 * i + 3];
        const float area_a = (ax2 - ax1) * (ay2 - ay1);
        for (int j = i + 1; j < n; ++j) {
            if (removed[j]) continue;
            const float bx1 = boxes[4 * j], by1 = boxes[4 * j + 1], bx2 = boxes[4 * j + 2], by2 = boxes[4 * j + 3];
            const float area_b = (bx2 - bx1) * (by2 - by1);
            float w = fminf(ax2, bx2) - fmaxf(ax1, bx1), h = fminf(ay2, by2) - fmaxf(ay1, by1);
            w = fmaxf(w, 0.0f); h = fmaxf(h, 0.0f);
            const float inter = w * h;
            if (inter / (area_a + area_b - inter) > thr) removed[j] = 1;
        }
    }
    free(removed);
    return count;
}

/* ------------------------------------------------------------------------------------------------------------
 * Bayer demosaic (EXTENSION; no reference implementation — the reference only has the inverse packing,
 * isp/unprocess_np.py:82-98 `mosaic`, :111-128 `reconstruct_bayer`). This function DEFINES the operation for the HIP
 * kernel: bilinear interpolation of the normalised samples with mirrored (edge-not-repeated) borders; fixed
 * summation order ((N+S)+(W+E)) and ((NW+NE)+(SW+SE)). pattern = 2*ry + rx: position of the red sample in the cell.
 * ------------------------------------------------------------------------------------------------------------ */
static int mirror_idx(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

int oracle_demosaic(const uint16_t* raw, float* out, int B, int H, int W, int pattern, float black, float white) {
    if (!raw || !out || (H & 1) || (W & 1) || pattern < 0 || pattern > 3) return -1;
    const int ry = pattern >> 1, rx = pattern & 1;
    const float inv = 1.0f / (white - black);
    const long plane = (long)H * W;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y) {
            const uint16_t* src = raw + (long)b * plane;
            float* o = out + (long)b * 3 * plane;
            for (int x = 0; x < W; ++x) {
#define S_(yy, xx) ((((float)src[(long)mirror_idx((yy), H) * W + mirror_idx((xx), W)]) - black) * inv)
                const float c = S_(y, x), n = S_(y - 1, x), s = S_(y + 1, x), w = S_(y, x - 1), e = S_(y, x + 1);
                const float cross = ((n + s) + (w + e)) * 0.25f;
                const float diag = ((S_(y - 1, x - 1) + S_(y - 1, x + 1)) + (S_(y + 1, x - 1) + S_(y + 1, x + 1))) * 0.25f;
                const float horiz = (w + e) * 0.5f, vert = (n + s) * 0.5f;
#undef S_
                const int py = (y - ry) & 1, px = (x - rx) & 1;
                float r, g, bl;
                if (py == 0 && px == 0) { r = c; g = cross; bl = diag; }
                else if (py == 0) { r = horiz; g = c; bl = vert; }
                else if (px == 0) { r = vert; g = c; bl = horiz; }
                else { r = diag; g = cross; bl = c; }
                o[(long)y * W + x] = r; o[plane + (long)y * W + x] = g; o[2 * plane + (long)y * W + x] = bl;
            }
        }
    return 0;
}
