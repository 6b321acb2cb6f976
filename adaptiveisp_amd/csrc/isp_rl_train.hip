// The small arithmetic of one RL training iteration around the networks, as kernels with hand-written backward passes
// (gfx950). Through ATen these are ~25 + ~40 launches per critic call (hand statistics of the 64x64 planes, value.py:65-80)
// and ~40 + ~60 for the TD target / loss arithmetic (train.py:262-305) on tensors of B or B x 4096 floats — host enqueue
// time that bounds the iteration. Here: one launch each way.
#include "isp_internal.h"

namespace adaisp {
namespace {

constexpr int kMaxG = ADAISP_TRUNK_MAX_G;
constexpr int kHW = 64 * 64;
constexpr int kPix = kHW / 256;            // pixels per thread of a 256-thread workgroup

__device__ __forceinline__ float block_sum256(float v, float* red) {
#pragma unroll
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float t = ((red[0] + red[1]) + red[2]) + red[3];
    __syncthreads();
    return t;
}

struct PlanesIO {
    const float* small[kMaxG];
    const float* states[kMaxG];
    float* svec[kMaxG];
    const float* dsvec[kMaxG];
    const float* dsmall_in[kMaxG];
    float* dsmall[kMaxG];
};

// saturation of a clipped pixel (value.py:72-75) and which channels carry its max / min (first index on ties, as
// torch.max / torch.min over a dimension report them)
__device__ __forceinline__ void sat_terms(const float c[3], float& mx, float& mn, int& imx, int& imn) {
    mx = c[0]; mn = c[0]; imx = 0; imn = 0;
#pragma unroll
    for (int k = 1; k < 3; ++k) {
        if (c[k] > mx) { mx = c[k]; imx = k; }
        if (c[k] < mn) { mn = c[k]; imn = k; }
    }
}

// svec[b] = [states[b], mean luminance, luminance variance (unbiased), mean saturation]
__global__ __launch_bounds__(256) void k_critic_planes_fwd(PlanesIO io, int n_state) {
    __shared__ float red[4];
    const int b = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
    const float* s = io.small[g] + (long)b * 3 * kHW;
    float lum[kPix];
    float sl = 0.0f, ss = 0.0f;
#pragma unroll
    for (int k = 0; k < kPix; ++k) {
        const int p = tid + 256 * k;
        const float r = s[p], gg = s[kHW + p], bb = s[2 * kHW + p];
        lum[k] = r * 0.27f + gg * 0.67f + bb * 0.06f + 1e-5f;
        sl += lum[k];
        const float c[3] = {fminf(fmaxf(r, 0.0f), 1.0f), fminf(fmaxf(gg, 0.0f), 1.0f), fminf(fmaxf(bb, 0.0f), 1.0f)};
        float mx, mn; int i0, i1;
        sat_terms(c, mx, mn, i0, i1);
        ss += (mx - mn) / (fminf(mx + mn, 2.0f - mx - mn) + 1e-2f);
    }
    const float mean = block_sum256(sl, red) / (float)kHW;
    float sd = 0.0f;
#pragma unroll
    for (int k = 0; k < kPix; ++k) { const float d = lum[k] - mean; sd += d * d; }
    const float var = block_sum256(sd, red) / (float)(kHW - 1);
    const float sat = block_sum256(ss, red) / (float)kHW;
    float* o = io.svec[g] + (long)b * (n_state + 3);
    for (int i = tid; i < n_state; i += 256) o[i] = io.states[g][(long)b * n_state + i];
    if (tid == 0) { o[n_state] = mean; o[n_state + 1] = var; o[n_state + 2] = sat; }
}

// dsmall = dsmall_in + the three statistics' gradients (mean luminance is read back from svec)
__global__ __launch_bounds__(256) void k_critic_planes_bwd(PlanesIO io, int n_state) {
    const int b = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
    if (!io.dsmall[g]) return;
    const float* s = io.small[g] + (long)b * 3 * kHW;
    const float* din = io.dsmall_in[g] ? io.dsmall_in[g] + (long)b * 3 * kHW : nullptr;
    float* d = io.dsmall[g] + (long)b * 3 * kHW;
    const float* dv = io.dsvec[g] + (long)b * (n_state + 3) + n_state;
    const float mean = io.svec[g][(long)b * (n_state + 3) + n_state];
    const float gm = dv[0] / (float)kHW, gv = dv[1] * 2.0f / (float)(kHW - 1), gs = dv[2] / (float)kHW;
#pragma unroll
    for (int k = 0; k < kPix; ++k) {
        const int p = tid + 256 * k;
        const float x[3] = {s[p], s[kHW + p], s[2 * kHW + p]};
        const float lum = x[0] * 0.27f + x[1] * 0.67f + x[2] * 0.06f + 1e-5f;
        const float dl = gm + gv * (lum - mean);
        float o[3] = {dl * 0.27f, dl * 0.67f, dl * 0.06f};
        const float c[3] = {fminf(fmaxf(x[0], 0.0f), 1.0f), fminf(fmaxf(x[1], 0.0f), 1.0f), fminf(fmaxf(x[2], 0.0f), 1.0f)};
        float mx, mn; int imx, imn;
        sat_terms(c, mx, mn, imx, imn);
        const float sum = mx + mn, alt = 2.0f - mx - mn;
        const float den = fminf(sum, alt) + 1e-2f;
        // d min(sum, alt) / d sum: 1 where sum is the smaller, -1 where alt is, 0 on a tie (torch.minimum halves the
        // gradient between its arguments there: 1/2 - 1/2)
        const float dd = sum < alt ? 1.0f : (sum > alt ? -1.0f : 0.0f);
        const float q = gs / den, t = q * (mx - mn) / den * dd;
        const float dmx = q - t, dmn = -q - t;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float pass = (x[ch] >= 0.0f && x[ch] <= 1.0f) ? 1.0f : 0.0f;        // torch.clip's backward
            float a = 0.0f;
            if (ch == imx) a += dmx;
            if (ch == imn) a += dmn;
            o[ch] += a * pass;
        }
        d[p] = o[0] + (din ? din[p] : 0.0f);
        d[kHW + p] = o[1] + (din ? din[kHW + p] : 0.0f);
        d[2 * kHW + p] = o[2] + (din ? din[2 * kHW + p] : 0.0f);
    }
}

// ---- TD target / losses (train.py:262-305 as rl.td_losses states them) ---------------------------------------------------
__device__ __forceinline__ float clip01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

struct TdRow {
    float k, reward, m, cf, q, adv, lre_pass;
};
__device__ __forceinline__ TdRow td_row(const adaisp_td_args& a, int b) {
    TdRow r;
    const float* st = a.new_states + (long)b * a.state_dim;
    const float stopped = st[1], step = st[2];
    const float lin = clip01(a.l_in[b] * a.detect_loss_weight);
    const float lre_raw = a.l_re[b] * a.detect_loss_weight;
    const float lre = clip01(lre_raw);
    r.lre_pass = (lre_raw >= 0.0f && lre_raw <= 1.0f) ? 1.0f : 0.0f;
    r.k = (a.all_reward + (1.0f - a.all_reward) * stopped) * a.critic_logit_multiplier;
    r.reward = (a.all_reward + (1.0f - a.all_reward) * stopped) * (lin - lre) * a.critic_logit_multiplier;
    if (a.use_penalty) r.reward = r.reward - a.penalty[b];
    r.cf = step > a.maximum_trajectory_length ? 1.0f : 0.0f;
    const float nv = a.new_value[b] * (1.0f - r.cf);
    if (a.use_truncated) {
        const float rm = a.retouch_mean[b];
        const float trunc = (0.01f < rm && rm < a.max_bri) ? 1.0f : 0.0f;
        r.m = (1.0f - stopped) * a.discount_factor * (1.0f - trunc);
        r.q = r.reward + (1.0f - stopped) * a.discount_factor * nv * (1.0f - trunc);
    } else {
        r.m = (1.0f - stopped) * a.discount_factor;
        r.q = r.reward + (1.0f - stopped) * a.discount_factor * nv;
    }
    r.adv = r.q - a.old_value[b];
    return r;
}

__global__ __launch_bounds__(64) void k_td_fwd(adaisp_td_args a) {
    float sv = 0.0f, sa = 0.0f;
    for (int b = threadIdx.x; b < a.B; b += 64) {
        const TdRow r = td_row(a, b);
        a.reward[b] = r.reward;
        a.q_value[b] = r.q;
        a.advantage[b] = r.adv;
        sv += r.adv * r.adv;
        const float routine = a.use_td ? -r.q * a.parameter_lr_mul : -r.reward;
        const float advp = a.use_td ? -r.adv : -r.reward;
        sa += routine + a.surrogate[b] * advp;
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) { sv += __shfl_xor(sv, off); sa += __shfl_xor(sa, off); }
    if (threadIdx.x == 0) { a.losses[0] = sv / (float)a.B; a.losses[1] = sa / (float)a.B; }
}

__global__ __launch_bounds__(64) void k_td_bwd(adaisp_td_args a) {
    const float gv = a.dlosses[0] / (float)a.B, ga = a.dlosses[1] / (float)a.B;
    for (int b = threadIdx.x; b < a.B; b += 64) {
        const TdRow r = td_row(a, b);
        a.d_old_value[b] = -2.0f * r.adv * gv;
        float dreward, dnv, dsur;
        if (a.use_td) {
            const float dq = -a.parameter_lr_mul * ga;
            dreward = dq;
            dnv = dq * r.m;
            dsur = -r.adv * ga;
        } else {
            dreward = -ga;
            dnv = 0.0f;
            dsur = -r.reward * ga;
        }
        a.d_new_value[b] = dnv * (1.0f - r.cf);
        a.d_surrogate[b] = dsur;
        a.d_l_re[b] = -dreward * r.k * r.lre_pass * a.detect_loss_weight;
        a.d_penalty[b] = a.use_penalty ? -dreward : 0.0f;
    }
}

// ---- the policy's tail in training mode (agent.py:103-149, 234-280) ---------------------------------------------------------
// Forward: every filter's regressor on the heads' pre-activations x [B][F][pw] (isp/filters.py: filter_param_regressor of each
// class, the table include/adaisp.h names), the selector's pdf from its logits (softmax + 1e-37, exploration mix,
// renormalisation), entropy, pdf_sample / forced id, surrogate, packed parameter row + op code of the selected filter, state
// update, penalty — the arithmetic and summation orders of the eval kernel (isp_policy.hip k_finish). Backward: only the selected
// filter's parameters reach the pixels, so d x is that row's regressor derivative; d logits collects the surrogate's and the
// entropy penalty's gradients through the renormalisation and the softmax. Workgroup = one image.
__device__ __forceinline__ float tanh01t(float x) { return tanhf(x) * 0.5f + 0.5f; }

__device__ __forceinline__ float regress(const adaisp_regressor& rg, const float* xr, int s) {
    const float x = xr[s];
    switch (rg.kind) {
        case ADAISP_REG_TANH_RANGE: return tanh01t(x + rg.bias) * rg.scale + rg.lo;
        case ADAISP_REG_EXP_TANH_RANGE: return expf(tanh01t(x + rg.bias) * rg.scale + rg.lo);
        case ADAISP_REG_SIGMOID: return 1.0f / (1.0f + expf(-x));
        case ADAISP_REG_TANH: return tanhf(x);
        default: {
            float gsc[3];
            for (int c = 0; c < 3; ++c) gsc[c] = expf(tanh01t(xr[c] * (c == 0 ? 0.0f : 1.0f) + rg.bias) * rg.scale + rg.lo);
            const float lum = ((1e-5f + 0.27f * gsc[0]) + 0.67f * gsc[1]) + 0.06f * gsc[2];
            return gsc[s] * (1.0f / lum);
        }
    }
}

__global__ __launch_bounds__(256) void k_policy_tail_fwd(adaisp_policy_tail_args a) {
    const int b = blockIdx.x, t = threadIdx.x;
    const int F = a.num_filters, PW = a.param_width;
    __shared__ float pdf[ADAISP_POLICY_MAX_FILTERS];
    __shared__ float entl[ADAISP_POLICY_MAX_FILTERS];
    __shared__ float sc[2];
    __shared__ int sel_sh;
    const float* xb = a.x + (long)b * F * PW;
    for (int i = t; i < F * PW; i += 256) {
        const int f = i / PW, s = i - f * PW;
        a.table[(long)b * F * PW + i] = s < a.reg[f].n ? regress(a.reg[f], xb + f * PW, s) : 0.0f;
    }
    const float* lg = a.logits + (long)b * F;
    if (t < F) {
        float mx = lg[0];
        for (int k = 1; k < F; ++k) mx = fmaxf(mx, lg[k]);
        pdf[t] = expf(lg[t] - mx);
    }
    __syncthreads();
    if (t == 0) {
        float sum = 0.0f;
        for (int k = 0; k < F; ++k) sum += pdf[k];
        sc[0] = sum;
    }
    __syncthreads();
    if (t < F) pdf[t] = (pdf[t] / sc[0] + 1e-37f) * a.one_minus_exploration + a.exploration_over_f;
    __syncthreads();
    if (t == 0) {
        float tot = 0.0f;
        for (int k = 0; k < F; ++k) tot += pdf[k];
        sc[1] = tot + 1e-30f;
    }
    __syncthreads();
    if (t < F) {
        const float p = pdf[t] / sc[1];
        pdf[t] = p;
        entl[t] = -p * logf(p);
    }
    __syncthreads();
    if (t == 0) {
        float ent = 0.0f;
        for (int k = 0; k < F; ++k) ent += entl[k];
        float s2 = 0.0f;
        for (int k = 0; k < F; ++k) s2 += pdf[k];
        s2 += 1e-36f;
        const float u = a.noise[(long)b * a.noise_stride];
        int cnt = 0, amax = 0;
        float run = 0.0f;
        for (int k = 0; k < F; ++k) {
            const float pk = pdf[k] / s2;
            run += pk;
            if (run - pk < u) ++cnt;
            if (pdf[k] > pdf[amax]) amax = k;
        }
        const int sel = a.forced_id >= 0 ? a.forced_id : (a.sample ? cnt - 1 : amax);
        sel_sh = sel;
        a.selected[b] = (long long)sel;
        a.op_ids[b] = (sel >= 0 && sel < F) ? a.reg[sel].op : ADAISP_OP_ZERO;
        for (int k = 0; k < F; ++k) a.pdf[(long)b * F + k] = pdf[k];
        a.surrogate[b] = (sel >= 0 && sel < F) ? logf(pdf[sel] + 1e-10f) : 0.0f;
        const int S = 3 + F;
        const float* st = a.states + (long)b * S;
        float* ns = a.new_states + (long)b * S;
        const float last = fabsf(st[2] + 1.0f - a.test_steps) < 1e-4f ? 1.0f : 0.0f;
        ns[0] = last; ns[1] = last; ns[2] = st[2] + 1.0f;
        float usage_pen = 0.0f;
        for (int k = 0; k < F; ++k) {
            const float oh = (k == sel) ? 1.0f : 0.0f;
            usage_pen += st[3 + k] * oh;
            ns[3 + k] = fmaxf(st[3 + k], oh);
        }
        const float entropy_pen = (a.entropy_coef_dev ? *a.entropy_coef_dev : a.entropy_coef) * (-ent + a.log_num_filters);
        const float early = (1.0f - last) * last * a.early_stop_penalty;
        float runtime_pen = 0.0f;
        if (a.runtime && sel >= 0 && sel < F) runtime_pen = a.runtime_lambda * a.runtime[sel];
        a.penalty[b] = 0.0f + entropy_pen + usage_pen * a.filter_usage_penalty + early + runtime_pen;
    }
    __syncthreads();
    const int sel = sel_sh;
    for (int s = t; s < PW; s += 256) {
        float v = 0.0f;
        if (sel >= 0 && sel < F && s < a.reg[sel].n) v = regress(a.reg[sel], xb + sel * PW, s);
        a.packed[(long)b * PW + s] = v;
    }
}

__global__ __launch_bounds__(256) void k_policy_tail_bwd(adaisp_policy_tail_args a) {
    const int b = blockIdx.x, t = threadIdx.x;
    const int F = a.num_filters, PW = a.param_width;
    const int sel = (int)a.selected[b];
    const bool live = sel >= 0 && sel < F;
    const float* xb = a.x + (long)b * F * PW;
    float* dx = a.d_x + (long)b * F * PW;
    const float* dp = a.d_packed ? a.d_packed + (long)b * PW : nullptr;
    for (int i = t; i < F * PW; i += 256) {
        const int f = i / PW, s = i - f * PW;
        float g = 0.0f;
        if (dp && live && f == sel && s < a.reg[f].n) {
            const adaisp_regressor rg = a.reg[f];
            const float* xr = xb + f * PW;
            const float x = xr[s];
            switch (rg.kind) {
                case ADAISP_REG_TANH_RANGE: { const float th = tanhf(x + rg.bias); g = dp[s] * rg.scale * 0.5f * (1.0f - th * th); } break;
                case ADAISP_REG_EXP_TANH_RANGE: {
                    const float th = tanhf(x + rg.bias);
                    g = dp[s] * expf((th * 0.5f + 0.5f) * rg.scale + rg.lo) * rg.scale * 0.5f * (1.0f - th * th);
                } break;
                case ADAISP_REG_SIGMOID: { const float sg = 1.0f / (1.0f + expf(-x)); g = dp[s] * sg * (1.0f - sg); } break;
                case ADAISP_REG_TANH: { const float th = tanhf(x); g = dp[s] * (1.0f - th * th); } break;
                default: {      // white balance: out_k = o_k / lum, o_k = exp(tanh_range(x_k keep_k)), lum = 1e-5 + w . o
                    float o[3], th[3];
                    for (int c = 0; c < 3; ++c) {
                        th[c] = tanhf(xr[c] * (c == 0 ? 0.0f : 1.0f) + rg.bias);
                        o[c] = expf((th[c] * 0.5f + 0.5f) * rg.scale + rg.lo);
                    }
                    const float lum = ((1e-5f + 0.27f * o[0]) + 0.67f * o[1]) + 0.06f * o[2];
                    const float w[3] = {0.27f, 0.67f, 0.06f};
                    float dot = 0.0f;
                    for (int c = 0; c < 3; ++c) dot += dp[c] * o[c];
                    const float d_o = dp[s] / lum - dot / (lum * lum) * w[s];
                    g = s == 0 ? 0.0f : d_o * o[s] * rg.scale * 0.5f * (1.0f - th[s] * th[s]);
                } break;
            }
        }
        dx[i] = g;
    }
    if (t == 0) {
        const float* lg = a.logits + (long)b * F;
        const float* pdf = a.pdf + (long)b * F;
        float sm[ADAISP_POLICY_MAX_FILTERS], dpdf[ADAISP_POLICY_MAX_FILTERS];
        float mx = lg[0];
        for (int k = 1; k < F; ++k) mx = fmaxf(mx, lg[k]);
        float sum = 0.0f;
        for (int k = 0; k < F; ++k) { sm[k] = expf(lg[k] - mx); sum += sm[k]; }
        float tot = 0.0f;
        for (int k = 0; k < F; ++k) {
            sm[k] = sm[k] / sum;
            tot += (sm[k] + 1e-37f) * a.one_minus_exploration + a.exploration_over_f;
        }
        tot += 1e-30f;
        const float dsur = a.d_surrogate ? a.d_surrogate[b] : 0.0f;
        const float dpen = (a.d_penalty ? a.d_penalty[b] : 0.0f) * (a.entropy_coef_dev ? *a.entropy_coef_dev : a.entropy_coef);
        float dotp = 0.0f;
        for (int k = 0; k < F; ++k) {
            dpdf[k] = dpen * (logf(pdf[k]) + 1.0f);
            if (live && k == sel) dpdf[k] += dsur / (pdf[k] + 1e-10f);
            dotp += dpdf[k] * pdf[k];
        }
        float dots = 0.0f;
        for (int k = 0; k < F; ++k) {
            dpdf[k] = (dpdf[k] - dotp) / tot * a.one_minus_exploration;      // now d softmax_k
            dots += dpdf[k] * sm[k];
        }
        for (int k = 0; k < F; ++k) a.d_logits[(long)b * F + k] = sm[k] * (dpdf[k] - dots);
    }
}

// ---- per-image mean and non-finite count of a batch (the TD target's brightness test, train.py:287-291, and the replay
// guard, train.py:374-381): 64 chunks per image summed in lane / wave / chunk order, then the chunks in index order ---------
constexpr int kStatChunks = 64;
__global__ __launch_bounds__(256) void k_image_stats_partial(const float* __restrict__ img, float* __restrict__ partial,
                                                             long n) {
    __shared__ float red[4];
    const int b = blockIdx.y, ch = blockIdx.x;
    const long per = (((n + 3) / 4 + kStatChunks - 1) / kStatChunks) * 4;      // floats per chunk, a multiple of 4; 64 chunks cover n (also n % 4 != 0)
    const long lo = ch * per, hi = lo + per < n ? lo + per : n;
    const float* p = img + (long)b * n;
    float sum = 0.0f, bad = 0.0f;
    const bool vec = (n % 4 == 0) && ((reinterpret_cast<uintptr_t>(p) & 15) == 0);
    if (vec) {
        for (long i = lo + 4 * threadIdx.x; i < hi; i += 4 * 256) {
            const float4 v = *reinterpret_cast<const float4*>(p + i);
            sum += (v.x + v.y) + (v.z + v.w);
            bad += (float)((!isfinite(v.x)) + (!isfinite(v.y)) + (!isfinite(v.z)) + (!isfinite(v.w)));
        }
    } else {
        for (long i = lo + threadIdx.x; i < hi; i += 256) {
            const float v = p[i];
            sum += v;
            bad += isfinite(v) ? 0.0f : 1.0f;
        }
    }
    sum = block_sum256(sum, red);
    bad = block_sum256(bad, red);
    if (threadIdx.x == 0) {
        partial[((long)b * kStatChunks + ch) * 2] = sum;
        partial[((long)b * kStatChunks + ch) * 2 + 1] = bad;
    }
}
__global__ __launch_bounds__(64) void k_image_stats_finish(const float* __restrict__ partial, float* __restrict__ stats, int B,
                                                           long n) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    float sum = 0.0f, bad = 0.0f;
    for (int c = 0; c < kStatChunks; ++c) {
        sum += partial[((long)b * kStatChunks + c) * 2];
        bad += partial[((long)b * kStatChunks + c) * 2 + 1];
    }
    stats[2 * b] = sum / (float)n;
    stats[2 * b + 1] = bad;
}

// ---- gradient-norm clip + Adam over a table of tensors (train.py:341-351: clip_grad_norm_(1e-5) then Adam.step) -------------
// torch does this in ~14 launches per model (foreach norms, stack, norm, reciprocal / clamp / mul, foreach mul, two fused-Adam
// launches at 1.3 TB/s); here three: squares per 4096-element chunk in a fixed order, one workgroup that adds the chunks in index
// order and makes the clip coefficient, and one streaming pass that applies the coefficient and the update. Arithmetic of
// torch's fused Adam (lerp for the first moment, bias corrections from the step count in double, eps outside the root).
constexpr int kChunk = 4096;

__device__ __forceinline__ int find_tensor(const adaisp_adam_tensor* t, int n, long chunk) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (t[mid].chunk0 <= chunk) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(256) void k_gradsq_partial(const adaisp_adam_tensor* __restrict__ table, int ntensors,
                                                        float* __restrict__ ws) {
    __shared__ float red[4];
    const long chunk = blockIdx.x;
    const adaisp_adam_tensor t = table[find_tensor(table, ntensors, chunk)];
    const long lo = (chunk - t.chunk0) * kChunk, hi = lo + kChunk < t.n ? lo + kChunk : t.n;
    float s = 0.0f;
    for (long i = lo + threadIdx.x; i < hi; i += 256) { const float g = t.g[i]; s += g * g; }
    s = block_sum256(s, red);
    if (threadIdx.x == 0) ws[chunk] = s;
}

__global__ __launch_bounds__(256) void k_gradnorm_finish(float* __restrict__ ws, long nchunks, float max_norm) {
    __shared__ float red[4];
    float s = 0.0f;
    for (long i = threadIdx.x; i < nchunks; i += 256) s += ws[i];
    s = block_sum256(s, red);
    if (threadIdx.x == 0) {
        const float total = sqrtf(s);
        // clip_grad_norm_'s coefficient (clamped to 1). A non-finite gradient makes the norm NaN: torch.clamp propagates it and every
        // gradient turns NaN — fminf alone would return 1 and step the finite elements unclipped
        ws[nchunks] = (total == total) ? fminf(max_norm / (total + 1e-6f), 1.0f) : total;
        ws[nchunks + 1] = total;
    }
}

__global__ __launch_bounds__(256) void k_adam(const adaisp_adam_tensor* __restrict__ table, int ntensors, const float* __restrict__ coefp,
                                              double lr, const double* __restrict__ lr_dev, double beta1, double beta2, double eps) {
    if (lr_dev) lr = *lr_dev;                                             // (a captured launch: this iteration's learning rate)
    const long chunk = blockIdx.x;
    const adaisp_adam_tensor t = table[find_tensor(table, ntensors, chunk)];
    const long lo = (chunk - t.chunk0) * kChunk, hi = lo + kChunk < t.n ? lo + kChunk : t.n;
    const float coef = coefp ? *coefp : 1.0f;
    const double step = (double)*t.step;
    // torch's functor keeps lr / betas / eps as doubles and the tensors' values as floats: the mixed expressions below round
    // where its do (the second moment's update and the eps addition run in double)
    const float bc1 = (float)(1.0 - pow(beta1, step)), bc2 = (float)(1.0 - pow(beta2, step));
    const float step_size = (float)(lr / (double)bc1), bc2_sqrt = sqrtf(bc2), w = (float)(1.0 - beta1);
    const double omb2 = 1.0 - beta2;
    auto one = [&](float g, float& m, float& v, float& p) {
        g *= coef;
        m = w < 0.5f ? m + w * (g - m) : g - (g - m) * (1.0f - w);          // at::lerp
        v = (float)(beta2 * (double)v + omb2 * (double)g * (double)g);
        p -= step_size * m / (float)((double)(sqrtf(v) / bc2_sqrt) + eps);
    };
    // 16-byte streaming accesses where the four arrays allow it (4 reads + 3 writes per element: HBM-bound)
    const bool vec = ((reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(t.g) | reinterpret_cast<uintptr_t>(t.m) |
                       reinterpret_cast<uintptr_t>(t.v)) & 15) == 0;
    const long nvec = vec ? (hi - lo) / 4 : 0;
    for (long q = threadIdx.x; q < nvec; q += 256) {
        const long i = lo + 4 * q;
        const float4 g = ld4_nt(reinterpret_cast<const float4*>(t.g + i));
        float4 m = *reinterpret_cast<const float4*>(t.m + i), v = *reinterpret_cast<const float4*>(t.v + i);
        float4 p = *reinterpret_cast<const float4*>(t.p + i);
        one(g.x, m.x, v.x, p.x); one(g.y, m.y, v.y, p.y); one(g.z, m.z, v.z, p.z); one(g.w, m.w, v.w, p.w);
        *reinterpret_cast<float4*>(t.m + i) = m;
        *reinterpret_cast<float4*>(t.v + i) = v;
        *reinterpret_cast<float4*>(t.p + i) = p;
    }
    for (long i = lo + 4 * nvec + threadIdx.x; i < hi; i += 256) {
        float m = t.m[i], v = t.v[i], p = t.p[i];
        one(t.g[i], m, v, p);
        t.m[i] = m; t.v[i] = v; t.p[i] = p;
    }
}

PlanesIO planes_io(const adaisp_critic_planes_args& a) {
    PlanesIO io{};
    for (int g = 0; g < a.G; ++g) {
        io.small[g] = a.small[g]; io.states[g] = a.states[g]; io.svec[g] = a.svec[g];
        io.dsvec[g] = a.dsvec[g]; io.dsmall_in[g] = a.dsmall_in[g]; io.dsmall[g] = a.dsmall[g];
    }
    return io;
}

}  // namespace
}  // namespace adaisp

using namespace adaisp;

extern "C" {

int adaisp_critic_planes_fwd(const adaisp_critic_planes_args* a, void* stream) {
    if (!a || a->G < 1 || a->G > kMaxG || a->B < 1 || a->n_state < 0) return ADAISP_EINVAL;
    for (int g = 0; g < a->G; ++g)
        if (!a->small[g] || !a->svec[g] || (a->n_state && !a->states[g])) return ADAISP_EINVAL;
    hipLaunchKernelGGL(k_critic_planes_fwd, dim3(a->B, a->G), dim3(256), 0, static_cast<hipStream_t>(stream), planes_io(*a),
                       a->n_state);
    return hipGetLastError() == hipSuccess ? ADAISP_OK : ADAISP_ELAUNCH;
}

int adaisp_critic_planes_bwd(const adaisp_critic_planes_args* a, void* stream) {
    if (!a || a->G < 1 || a->G > kMaxG || a->B < 1 || a->n_state < 0) return ADAISP_EINVAL;
    for (int g = 0; g < a->G; ++g)
        if (a->dsmall[g] && (!a->small[g] || !a->svec[g] || !a->dsvec[g])) return ADAISP_EINVAL;
    hipLaunchKernelGGL(k_critic_planes_bwd, dim3(a->B, a->G), dim3(256), 0, static_cast<hipStream_t>(stream), planes_io(*a),
                       a->n_state);
    return hipGetLastError() == hipSuccess ? ADAISP_OK : ADAISP_ELAUNCH;
}

static int td_check(const adaisp_td_args* a) {
    if (!a || a->B < 1 || a->state_dim < 3) return ADAISP_EINVAL;
    if (!a->l_in || !a->l_re || !a->surrogate || !a->new_states || !a->old_value || !a->new_value) return ADAISP_EINVAL;
    if ((a->use_penalty && !a->penalty) || (a->use_truncated && !a->retouch_mean)) return ADAISP_EINVAL;
    return ADAISP_OK;
}

int adaisp_td_fwd(const adaisp_td_args* a, void* stream) {
    const int rc = td_check(a);
    if (rc != ADAISP_OK) return rc;
    if (!a->reward || !a->q_value || !a->advantage || !a->losses) return ADAISP_EINVAL;
    hipLaunchKernelGGL(k_td_fwd, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), *a);
    return hipGetLastError() == hipSuccess ? ADAISP_OK : ADAISP_ELAUNCH;
}

int adaisp_td_bwd(const adaisp_td_args* a, void* stream) {
    const int rc = td_check(a);
    if (rc != ADAISP_OK) return rc;
    if (!a->dlosses || !a->d_l_re || !a->d_penalty || !a->d_surrogate || !a->d_old_value || !a->d_new_value) return ADAISP_EINVAL;
    hipLaunchKernelGGL(k_td_bwd, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), *a);
    return hipGetLastError() == hipSuccess ? ADAISP_OK : ADAISP_ELAUNCH;
}

int adaisp_image_stats(const float* img, float* stats, float* workspace, int B, long n, void* stream) {
    if (!img || !stats || !workspace || B < 1 || n < 1) return ADAISP_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_image_stats_partial, dim3(kStatChunks, B), dim3(256), 0, s, img, workspace, n);
    hipLaunchKernelGGL(k_image_stats_finish, dim3((B + 63) / 64), dim3(64), 0, s, workspace, stats, B, n);
    return hipGetLastError() == hipSuccess ? ADAISP_OK : ADAISP_ELAUNCH;
}

static int clip_adam(const adaisp_adam_tensor* table, int ntensors, long nchunks, float* workspace, float max_norm, double lr,
                     const double* lr_dev, double beta1, double beta2, double eps, void* stream) {
    if (!table || !workspace || ntensors < 1 || nchunks < 1 || nchunks > 0x7fffffffL) return ADAISP_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool clip = max_norm > 0.0f;
    if (clip) {
        hipLaunchKernelGGL(k_gradsq_partial, dim3((unsigned)nchunks), dim3(256), 0, s, table, ntensors, workspace);
        hipLaunchKernelGGL(k_gradnorm_finish, dim3(1), dim3(256), 0, s, workspace, nchunks, max_norm);
    }
    hipLaunchKernelGGL(k_adam, dim3((unsigned)nchunks), dim3(256), 0, s, table, ntensors,
                       clip ? workspace + nchunks : static_cast<const float*>(nullptr), lr, lr_dev, beta1, beta2, eps);
    return hipGetLastError() == hipSuccess ? ADAISP_OK : ADAISP_ELAUNCH;
}

int adaisp_clip_adam_step(const adaisp_adam_tensor* table, int ntensors, long nchunks, float* workspace, float max_norm, double lr,
                          double beta1, double beta2, double eps, void* stream) {
    return clip_adam(table, ntensors, nchunks, workspace, max_norm, lr, nullptr, beta1, beta2, eps, stream);
}

int adaisp_clip_adam_step_dev(const adaisp_adam_tensor* table, int ntensors, long nchunks, float* workspace, float max_norm,
                              const double* lr_dev, double beta1, double beta2, double eps, void* stream) {
    if (!lr_dev) return ADAISP_EINVAL;
    return clip_adam(table, ntensors, nchunks, workspace, max_norm, 0.0, lr_dev, beta1, beta2, eps, stream);
}

static int tail_check(const adaisp_policy_tail_args* a) {
    if (!a || a->B < 1) return ADAISP_EINVAL;
    if (a->num_filters < 1 || a->num_filters > ADAISP_POLICY_MAX_FILTERS || a->param_width < 1 ||
        a->param_width > ADAISP_MAX_PARAMS || a->noise_stride < 1 || a->forced_id >= a->num_filters)
        return ADAISP_ESHAPE;
    if (!a->x || !a->logits || !a->noise || !a->states || !a->pdf || !a->selected) return ADAISP_EINVAL;
    return ADAISP_OK;
}

int adaisp_policy_tail_fwd(const adaisp_policy_tail_args* a, void* stream) {
    const int rc = tail_check(a);
    if (rc != ADAISP_OK) return rc;
    if (!a->table || !a->packed || !a->op_ids || !a->surrogate || !a->new_states || !a->penalty) return ADAISP_EINVAL;
    hipLaunchKernelGGL(k_policy_tail_fwd, dim3(a->B), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
    return hipGetLastError() == hipSuccess ? ADAISP_OK : ADAISP_ELAUNCH;
}

int adaisp_policy_tail_bwd(const adaisp_policy_tail_args* a, void* stream) {
    const int rc = tail_check(a);
    if (rc != ADAISP_OK) return rc;
    if (!a->d_x || !a->d_logits) return ADAISP_EINVAL;
    hipLaunchKernelGGL(k_policy_tail_bwd, dim3(a->B), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
    return hipGetLastError() == hipSuccess ? ADAISP_OK : ADAISP_ELAUNCH;
}

}  // extern "C"
