// Per-image detection loss of the RL reward (CIoU box + BCE objectness + BCE class) on the raw head maps, forward and
// backward, for gfx950.
//
// Reference: ComputeLossBatch / ComputeLoss.__call__, yolov3/utils/loss.py:237-318,91-170 (per layer: gather the matched
// cells, pxy = 2 sigmoid - 0.5, pwh = (2 sigmoid)^2 anchor, CIoU against the target box, tobj[cell] = clamp(CIoU, 0),
// class BCE on the matched cells, objectness BCE on EVERY cell, layer balance 4 / 1 / 0.4), bbox_iou(CIoU=True),
// yolov3/utils/metrics.py:222-260, called once per SAMPLE with the image index of its targets set to 0
// (train.py:175-197): 2 B evaluations of ~150 ATen launches per training iteration. The target assignment
// (build_targets, loss.py:320-380) depends on the labels and the map SHAPES only and stays with the caller; what it
// yields per layer — (image, anchor, gj, gi, class) and (target box, anchor size) per match — is this file's input.
//
// One workgroup per image and layer: the image's matches, one per wave (lanes split the 80 classes and share the scans
// of the match list: CIoU, class BCE), the objectness targets (a cell matched more than once keeps the LAST match's
// value, as a sequential index assignment does), then the dense objectness BCE; the last workgroup of an image adds the
// layers. Every sum is a fixed-order tree: the loss is bit-reproducible. The backward
// pass is two launches: a dense one that writes the WHOLE gradient map (zeros + the objectness channel) and a
// per-match one that recomputes CIoU with its analytic gradient and adds the box / class terms (duplicates of a cell
// are summed in match order by the first of them). The maps are the detector's own bf16 NHWC buffers — no fp32
// [B, na, ny, nx, no] copies in either direction.
#include "yolo_internal.h"

namespace adayolo {
namespace dl {

constexpr int kThreads = 256;
constexpr float kEps = 1e-7f;

__device__ __forceinline__ float bf(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short to_bf(float f) {            // round to nearest even (as torch's .to(bfloat16))
    unsigned u = __float_as_uint(f);
    if ((u & 0x7F800000u) == 0x7F800000u && (u & 0x007FFFFFu)) return (unsigned short)((u >> 16) | 0x40u);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }
// F.binary_cross_entropy_with_logits(x, t, pos_weight = pw): (1 - t) x + (1 + (pw - 1) t) (log1p(exp(-|x|)) + max(-x, 0))
__device__ __forceinline__ float bce(float x, float t, float pw) {
    const float lw = (pw - 1.0f) * t + 1.0f;
    return (1.0f - t) * x + lw * (log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.0f));
}
__device__ __forceinline__ float bce_grad(float x, float t, float pw) {   // d bce / d x
    const float lw = (pw - 1.0f) * t + 1.0f;
    return (1.0f - t) - lw * (1.0f - sigmoidf(x));
}
// subgradient weights of torch.minimum / torch.maximum towards their FIRST argument (ties are split)
__device__ __forceinline__ float wmin(float a, float b) { return a < b ? 1.0f : (a == b ? 0.5f : 0.0f); }
__device__ __forceinline__ float wmax(float a, float b) { return a > b ? 1.0f : (a == b ? 0.5f : 0.0f); }

struct Match {
    float ciou;            // CIoU of the predicted box against the target
    float g[4];            // d ciou / d (logit x, y, w, h) — filled when GRAD
};

// box logits (lx, ly, lw, lh), anchor (aw, ah), target (x2, y2, w2, h2) in cell units
template <bool GRAD>
__device__ __forceinline__ Match ciou_match(const float lg[4], float aw, float ah, float x2, float y2, float w2, float h2) {
    const float sx = sigmoidf(lg[0]), sy = sigmoidf(lg[1]), sw = sigmoidf(lg[2]), sh = sigmoidf(lg[3]);
    const float x1 = sx * 2.0f - 0.5f, y1 = sy * 2.0f - 0.5f;
    const float w1 = (sw * 2.0f) * (sw * 2.0f) * aw, h1 = (sh * 2.0f) * (sh * 2.0f) * ah;
    const float l1 = x1 - w1 / 2, r1 = x1 + w1 / 2, t1 = y1 - h1 / 2, b1 = y1 + h1 / 2;
    const float l2 = x2 - w2 / 2, r2 = x2 + w2 / 2, t2 = y2 - h2 / 2, b2 = y2 + h2 / 2;
    const float iwr = fminf(r1, r2) - fmaxf(l1, l2), ihr = fminf(b1, b2) - fmaxf(t1, t2);
    const float iw = fmaxf(iwr, 0.0f), ih = fmaxf(ihr, 0.0f);
    const float inter = iw * ih;
    const float uni = w1 * h1 + w2 * h2 - inter + kEps;
    const float iou = inter / uni;
    const float cw = fmaxf(r1, r2) - fminf(l1, l2), ch = fmaxf(b1, b2) - fminf(t1, t2);
    const float c2 = cw * cw + ch * ch + kEps;
    const float dx = l2 + r2 - l1 - r1, dy = t2 + b2 - t1 - b1;
    const float rho2 = (dx * dx + dy * dy) / 4;
    const float kV = 0.40528473456935109f;                               // 4 / pi^2
    const float da = atanf(w2 / h2) - atanf(w1 / h1);
    const float v = kV * da * da;
    const float alpha = v / (v - iou + (1.0f + kEps));                    // no gradient through alpha (torch.no_grad)
    Match m;
    m.ciou = iou - (rho2 / c2 + v * alpha);
    if (GRAD) {
        // base variables q = (x1, y1, w1, h1); l1 = x1 - w1/2, r1 = x1 + w1/2, t1 = y1 - h1/2, b1 = y1 + h1/2
        const float kiw = iwr >= 0.0f ? 1.0f : 0.0f, kih = ihr >= 0.0f ? 1.0f : 0.0f;   // clamp(0) passes the gradient at 0
        const float diw_dr = wmin(r1, r2) * kiw, diw_dl = -wmax(l1, l2) * kiw;
        const float dih_db = wmin(b1, b2) * kih, dih_dt = -wmax(t1, t2) * kih;
        const float dcw_dr = wmax(r1, r2), dcw_dl = -wmin(l1, l2);
        const float dch_db = wmax(b1, b2), dch_dt = -wmin(t1, t2);
        const float u = w1 / h1, dat = 1.0f / (1.0f + u * u);
        // per base variable: (d l1|t1, d r1|b1) selectors
        const float dinter[4] = {ih * (diw_dr + diw_dl), iw * (dih_db + dih_dt), ih * 0.5f * (diw_dr - diw_dl), iw * 0.5f * (dih_db - dih_dt)};
        const float duni[4] = {-dinter[0], -dinter[1], h1 - dinter[2], w1 - dinter[3]};
        const float dc2[4] = {2 * cw * (dcw_dr + dcw_dl), 2 * ch * (dch_db + dch_dt), 2 * cw * 0.5f * (dcw_dr - dcw_dl),
                              2 * ch * 0.5f * (dch_db - dch_dt)};
        const float drho[4] = {-dx, -dy, 0.0f, 0.0f};
        const float dv[4] = {0.0f, 0.0f, 2 * kV * da * (-dat / h1), 2 * kV * da * (dat * w1 / (h1 * h1))};
        const float dq[4] = {2.0f * sx * (1.0f - sx), 2.0f * sy * (1.0f - sy), 8.0f * sw * sw * (1.0f - sw) * aw,
                             8.0f * sh * sh * (1.0f - sh) * ah};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float diou = (dinter[k] * uni - inter * duni[k]) / (uni * uni);
            const float dpen = (drho[k] * c2 - rho2 * dc2[k]) / (c2 * c2) + alpha * dv[k];
            m.g[k] = (diou - dpen) * dq[k];
        }
    }
    return m;
}

// fixed-order sum of one value per thread over the workgroup (every thread gets the result)
__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = red[0];
    for (int w = 1; w < kThreads / 64; ++w) s += red[w];
    return s;
}

__device__ __forceinline__ void load_box_logits(const unsigned short* cell, float lg[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) lg[k] = bf(cell[k]);
}
__device__ __forceinline__ float wave_sum(float v) {                    // fixed xor tree: every lane gets the sum
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// The matches of a layer are walked one per WAVE (wave w of the workgroup takes matches w, w + 4, ...): the lanes share
// the scans over the match list (is a later / an earlier match on the same cell?) and split the classes. The list of
// (image, anchor, gj, gi, class) rows is staged in LDS when it fits (kMaxStage rows), otherwise read from memory.
constexpr int kMaxStage = 2048;
struct IdxList {
    const int* g;                 // [n][5] in memory
    const int* s;                 // LDS copy or nullptr
    __device__ __forceinline__ const int* row(int j) const { return (s ? s : g) + 5 * j; }
};
__device__ __forceinline__ bool same_cell(const int* p, const int* q) {
    return p[0] == q[0] && p[1] == q[1] && p[2] == q[2] && p[3] == q[3];
}
// any j2 in [lo, hi) on the same cell as `id`? (wave-cooperative, wave-uniform result)
__device__ __forceinline__ bool any_same(const IdxList& L, const int* id, int lo, int hi, int lane) {
    bool hit = false;
    for (int j2 = lo + lane; j2 < hi; j2 += 64) hit |= same_cell(L.row(j2), id);
    return __ballot(hit) != 0ull;
}

// grid (B, nl): one workgroup per image and layer; the last of an image's workgroups adds the layers up (fixed order)
__global__ __launch_bounds__(kThreads) void k_detloss_fwd(const adayolo_loss_args a) {
    __shared__ float red[kThreads / 64];
    __shared__ int sidx[5 * kMaxStage];
    __shared__ int ticket;
    const int b = blockIdx.x, i = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const adayolo_loss_layer L = a.layer[i];
    const int plane = L.ny * L.nx, cells = a.na * plane;
    float* tobj = L.tobj + (long)b * cells;
    const unsigned short* raw = static_cast<const unsigned short*>(L.raw) + (long)b * plane * L.cs;
    IdxList idx{L.idx, nullptr};
    if (L.n <= kMaxStage) {
        for (int t = tid; t < 5 * L.n; t += kThreads) sidx[t] = L.idx[t];
        idx.s = sidx;
    }
    for (int c = tid; c < cells; c += kThreads) tobj[c] = 0.0f;
    __syncthreads();                                        // list staged, objectness targets zeroed (workgroup scope)
    float sbox = 0.0f, scls = 0.0f, cnt = 0.0f;             // lane 0 of each wave accumulates its wave's matches in order
    for (int j = wave; j < L.n; j += kThreads / 64) {
        const int* id = idx.row(j);
        if (id[0] != b) continue;                           // wave-uniform
        const float* bx = L.box + 6 * j;
        const unsigned short* cell = raw + ((long)id[2] * L.nx + id[3]) * L.cs + id[1] * a.no;
        float lg[4];
        load_box_logits(cell, lg);
        const Match m = ciou_match<false>(lg, bx[4], bx[5], bx[0], bx[1], bx[2], bx[3]);
        float s = 0.0f;
        if (a.nc > 1)
            for (int c = lane; c < a.nc; c += 64) s += bce(bf(cell[5 + c]), c == id[4] ? a.cp : a.cn, a.cls_pw);
        s = wave_sum(s);
        // a later match of the same cell overwrites this one's objectness target (sequential index assignment)
        const bool last = !any_same(idx, id, j + 1, L.n, lane);
        if (lane == 0) {
            sbox += 1.0f - m.ciou;
            scls += s;
            cnt += 1.0f;
            if (last) tobj[(id[1] * L.ny + id[2]) * L.nx + id[3]] = fmaxf(m.ciou, 0.0f);
        }
    }
    __syncthreads();
    float sobj = 0.0f;
    for (int c = tid; c < cells; c += kThreads) {
        const int an = c / plane, p = c - an * plane;
        sobj += bce(bf(raw[(long)p * L.cs + an * a.no + 4]), tobj[c], a.obj_pw);
    }
    sbox = block_sum(sbox, red);
    scls = block_sum(scls, red);
    cnt = block_sum(cnt, red);
    sobj = block_sum(sobj, red);
    if (tid == 0) {
        L.cnt[b] = cnt;
        float* part = L.part + 3 * b;                       // (lbox, lobj, lcls) of this image in this layer
        part[0] = cnt > 0.0f ? sbox / cnt : 0.0f;
        part[1] = sobj / (float)cells * L.balance;
        part[2] = cnt > 0.0f ? scls / cnt / (float)a.nc : 0.0f;
        __threadfence();
        ticket = atomicAdd(a.ticket + b, 1);
    }
    __syncthreads();
    if (ticket == a.nl - 1 && tid == 0) {                   // every layer of this image is in: add them in layer order
        __threadfence();
        float lbox = 0.0f, lobj = 0.0f, lcls = 0.0f;
        for (int l = 0; l < a.nl; ++l) {
            const volatile float* part = a.layer[l].part + 3 * b;
            lbox += part[0]; lobj += part[1]; lcls += part[2];
        }
        a.loss[b] = lbox * a.hyp_box + lobj * a.hyp_obj + lcls * a.hyp_cls;
        a.ticket[b] = 0;                                    // ready for the next call on this stream
    }
}

// dense part of the backward pass: every 16-byte chunk of the gradient maps (zeros, and the objectness channel of each anchor)
__global__ __launch_bounds__(kThreads) void k_detloss_bwd_dense(const adayolo_loss_args a) {
    const adayolo_loss_layer L = a.layer[blockIdx.z];
    const int b = blockIdx.y;
    const int plane = L.ny * L.nx, cells = a.na * plane, chunks = L.grad_cs >> 3;
    const long t = (long)blockIdx.x * kThreads + threadIdx.x;
    if (t >= (long)plane * chunks) return;
    const int p = (int)(t / chunks), c0 = (int)(t - (long)p * chunks) * 8;
    const unsigned short* raw = static_cast<const unsigned short*>(L.raw) + ((long)b * plane + p) * L.cs;
    const float* tobj = L.tobj + (long)b * cells;
    const float scale = a.grad_loss[b] * a.hyp_obj * L.balance / (float)cells;
    unsigned short o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int ch = c0 + k;
        o[k] = 0;
        if (ch < a.na * a.no) {
            const int an = ch / a.no;
            if (ch - an * a.no == 4) o[k] = to_bf(scale * bce_grad(bf(raw[ch]), tobj[an * plane + p], a.obj_pw));
        }
    }
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    const u32x4 v = {o[0] | ((unsigned)o[1] << 16), o[2] | ((unsigned)o[3] << 16), o[4] | ((unsigned)o[5] << 16),
                     o[6] | ((unsigned)o[7] << 16)};
    *reinterpret_cast<u32x4*>(static_cast<unsigned short*>(L.grad) + ((long)b * plane + p) * L.grad_cs + c0) = v;
}

// per-match part: box and class terms of the matched cells; the first match of a cell sums all matches of that cell
__global__ __launch_bounds__(kThreads) void k_detloss_bwd_match(const adayolo_loss_args a) {
    __shared__ int sidx[5 * kMaxStage];
    const adayolo_loss_layer L = a.layer[blockIdx.y];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float cnt = L.cnt[b];
    if (cnt <= 0.0f) return;
    IdxList idx{L.idx, nullptr};
    if (L.n <= kMaxStage) {
        for (int t = tid; t < 5 * L.n; t += kThreads) sidx[t] = L.idx[t];
        idx.s = sidx;
    }
    __syncthreads();
    const int plane = L.ny * L.nx;
    const unsigned short* raw = static_cast<const unsigned short*>(L.raw) + (long)b * plane * L.cs;
    unsigned short* grad = static_cast<unsigned short*>(L.grad) + (long)b * plane * L.grad_cs;
    const float gbox = -a.grad_loss[b] * a.hyp_box / cnt;                // d (1 - ciou)
    const float gcls = a.grad_loss[b] * a.hyp_cls / cnt / (float)a.nc;
    for (int j = wave; j < L.n; j += kThreads / 64) {
        const int* id = idx.row(j);
        if (id[0] != b) continue;                                       // wave-uniform
        if (any_same(idx, id, 0, j, lane)) continue;                    // an earlier match of this cell does the work
        const long off = ((long)id[2] * L.nx + id[3]);
        const unsigned short* cell = raw + off * L.cs + id[1] * a.no;
        unsigned short* gcell = grad + off * L.grad_cs + id[1] * a.no;
        float lg[4], gb[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        load_box_logits(cell, lg);
        const float x0 = lane < a.nc ? bf(cell[5 + lane]) : 0.0f, x1 = lane + 64 < a.nc ? bf(cell[5 + lane + 64]) : 0.0f;
        float g0 = 0.0f, g1 = 0.0f;
        for (int base = j; base < L.n; base += 64) {                    // the cell's matches in match order
            const int j2 = base + lane;
            unsigned long long mask = __ballot(j2 < L.n && same_cell(idx.row(j2 < L.n ? j2 : j), id));
            while (mask) {
                const int k = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const int jm = base + k;
                const float* bx = L.box + 6 * jm;
                const Match m = ciou_match<true>(lg, bx[4], bx[5], bx[0], bx[1], bx[2], bx[3]);
#pragma unroll
                for (int q = 0; q < 4; ++q) gb[q] += gbox * m.g[q];
                const int cls = idx.row(jm)[4];
                g0 += gcls * bce_grad(x0, lane == cls ? a.cp : a.cn, a.cls_pw);
                g1 += gcls * bce_grad(x1, lane + 64 == cls ? a.cp : a.cn, a.cls_pw);
            }
        }
        if (lane < 4) gcell[lane] = to_bf(lane == 0 ? gb[0] : lane == 1 ? gb[1] : lane == 2 ? gb[2] : gb[3]);
        if (a.nc > 1) {
            if (lane < a.nc) gcell[5 + lane] = to_bf(g0);
            if (lane + 64 < a.nc) gcell[5 + lane + 64] = to_bf(g1);
        }
    }
}

}  // namespace dl

hipError_t launch_detloss_fwd(const adayolo_loss_args& a, hipStream_t s) {
    hipLaunchKernelGGL(dl::k_detloss_fwd, dim3(a.B, a.nl), dim3(dl::kThreads), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_detloss_bwd(const adayolo_loss_args& a, hipStream_t s) {
    long most = 0;
    for (int i = 0; i < a.nl; ++i) {
        const long t = (long)a.layer[i].ny * a.layer[i].nx * (a.layer[i].grad_cs >> 3);
        most = t > most ? t : most;
    }
    hipLaunchKernelGGL(dl::k_detloss_bwd_dense, dim3((unsigned)((most + dl::kThreads - 1) / dl::kThreads), a.B, a.nl),
                       dim3(dl::kThreads), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(dl::k_detloss_bwd_match, dim3(a.B, a.nl), dim3(dl::kThreads), 0, s, a);
    return hipGetLastError();
}

}  // namespace adayolo
