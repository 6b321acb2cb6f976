// C-ABI of libadayolo.so (include/adayolo.h): argument checks + launches. No allocation, no sync.
#include "yolo_internal.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

using namespace adayolo;

// n / d == umulhi(n, magic) >> sh for every 0 <= n < 2^31 (magic = ceil(2^(31+L) / d), L = ceil(log2 d), sh = L - 1)
static void magic_div(unsigned d, unsigned* magic, int* sh) {
    if (d <= 1) { *magic = 0; *sh = -1; return; }
    int L = 0;
    while ((1ull << L) < d) ++L;
    const unsigned long long num = 1ull << (31 + L);
    *magic = (unsigned)((num + d - 1) / d);
    *sh = L - 1;
}

// Shared checks + derived fields of a conv launch (every conv entry point below). ADAYOLO_OK or the error to return.
static int conv_args(ConvArgs& a, const void* in, int in_cstride, const void* weight, const float* bias, const void* residual,
                     int res_cstride, void* out, int out_cstride, int B, int H, int W, int Cin, int Cout, int ksize, int stride,
                     int act) {
    if (!in || !weight || !bias || !out) return ADAYOLO_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return ADAYOLO_EINVAL;
    if ((ksize != 1 && ksize != 3) || (stride != 1 && stride != 2)) return ADAYOLO_ESHAPE;
    if (Cin % 8 || Cout % 8 || in_cstride % 8 || out_cstride % 8 || in_cstride < Cin || out_cstride < Cout)
        return ADAYOLO_ESHAPE;
    if (residual && (res_cstride % 8 || res_cstride < Cout)) return ADAYOLO_ESHAPE;
    if (act != ADAYOLO_ACT_NONE && act != ADAYOLO_ACT_SILU) return ADAYOLO_EINVAL;
    a.in = static_cast<const unsigned short*>(in); a.in_cs = in_cstride;
    a.w = static_cast<const unsigned short*>(weight); a.bias = bias;
    a.res = static_cast<const unsigned short*>(residual); a.res_cs = res_cstride;
    a.out = static_cast<unsigned short*>(out); a.out_cs = out_cstride;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.ks = ksize; a.stride = stride; a.pad = ksize / 2; a.act = act;
    a.Ho = (H + 2 * a.pad - ksize) / stride + 1;
    a.Wo = (W + 2 * a.pad - ksize) / stride + 1;
    const long M = (long)B * a.Ho * a.Wo;
    if (M > 0x7fffffffL || (long)B * H * W * in_cstride > 0x7fffffffffL) return ADAYOLO_ESHAPE;
    a.M = (int)M; a.mtiles = a.ntiles = 0;
    a.w2 = nullptr; a.bias2 = nullptr; a.out2 = nullptr; a.out2_cs = 0;
    a.pre = nullptr; a.pre_cs = 0;
    magic_div((unsigned)(a.Ho * a.Wo), &a.magic_hw, &a.sh_hw);
    magic_div((unsigned)a.Wo, &a.magic_w, &a.sh_w);
    return ADAYOLO_OK;
}

extern "C" {

int adayolo_abi_version(void) { return ADAYOLO_ABI_VERSION; }

const char* adayolo_strerror(int code) {
    switch (code) {
        case ADAYOLO_OK: return "ok";
        case ADAYOLO_EINVAL: return "invalid argument (null pointer, non-positive size or unknown kernel variant)";
        case ADAYOLO_ESHAPE: return "shape not supported (channels/strides must be multiples of 8, ksize 1 or 3, stride 1 or 2)";
        case ADAYOLO_ELAUNCH: return "kernel launch failed";
        default: return "unknown error";
    }
}

#define ADAYOLO_DEFAULT_VARIANT 2

int adayolo_conv_fwd_variant(const void* in, int in_cstride, const void* weight, const float* bias,
                             const void* residual, int res_cstride, void* out, int out_cstride, int B, int H, int W,
                             int Cin, int Cout, int ksize, int stride, int act, int variant, void* stream) {
    ConvArgs a;
    const int rc = conv_args(a, in, in_cstride, weight, bias, residual, res_cstride, out, out_cstride, B, H, W, Cin, Cout,
                             ksize, stride, act);
    if (rc != ADAYOLO_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (variant == 0) variant = ADAYOLO_DEFAULT_VARIANT;
    hipError_t e = hipErrorInvalidValue;
    bool known = false;
    switch (variant) {
        case 2: known = true; e = launch_conv_dma(a, s, variant); break;
        case 5: case 22: case 26: case 27: known = true; e = launch_conv_dma2(a, s, variant); break;
        case 40: known = true; e = launch_conv_small(a, s, variant); break;       // 3x3, Cin 32 / 64 only
        case 50: known = true; e = launch_conv_pp(a, s, variant); break;          // Cin % 64 == 0, Cout % 256 == 0 only
        case 60: known = true; e = launch_conv_pp128(a, s, variant); break;       // Cin % 64 == 0, Cout % 128 == 0 only
        case 80: case 85: known = true; e = launch_conv_pq(a, s, variant); break; // Cin % 32 == 0, Cout % 128 == 0 (85: 128-pixel tile)
        case 90: known = true; e = launch_conv_ws(a, s, variant); break;          // 3x3 s1, Cin 32 / 64, Cout % 64 == 0, SiLU
        default: break;
    }
#ifdef ADAYOLO_MEASURE
    if (!known) {
        known = true;
        if (variant >= 90) e = launch_conv_ws(a, s, variant);
        else if (variant >= 80) e = launch_conv_pq(a, s, variant);
        else if (variant >= 60) e = launch_conv_pp128(a, s, variant);
        else if (variant >= 50) e = launch_conv_pp(a, s, variant);
        else e = launch_conv_dma2(a, s, variant);
    }
#endif
    if (!known) return ADAYOLO_EINVAL;                                            // not a kernel of this library
    if (e == hipErrorInvalidValue && variant >= 40) e = launch_conv_dma(a, s, ADAYOLO_DEFAULT_VARIANT);   // shape not served
    return e == hipSuccess ? ADAYOLO_OK : ADAYOLO_ELAUNCH;
}

int adayolo_conv_fused1x1_fwd(const void* in, int in_cstride, const void* weight, const float* bias, const void* residual,
                              int res_cstride, void* out, int out_cstride, int B, int H, int W, int Cin, int Cout, int ksize,
                              int stride, int act, const void* weight2, const float* bias2, void* out2, int out2_cstride,
                              int Cout2, void* stream) {
    if (!in || !weight || !bias || !out || !weight2 || !bias2 || !out2) return ADAYOLO_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return ADAYOLO_EINVAL;
    if ((ksize != 1 && ksize != 3) || (stride != 1 && stride != 2)) return ADAYOLO_ESHAPE;
    // the first layer's tile must hold every output channel (256) and be served by the 256 x 256 kernel; the second
    // layer is Conv(256 -> 128, k1) + SiLU
    if (Cout != 256 || Cout2 != 128 || Cin % 64 || in_cstride % 8 || out_cstride % 8 || out2_cstride % 8 ||
        in_cstride < Cin || out_cstride < Cout || out2_cstride < Cout2)
        return ADAYOLO_ESHAPE;
    if (residual && (res_cstride % 8 || res_cstride < Cout)) return ADAYOLO_ESHAPE;
    if (act != ADAYOLO_ACT_NONE && act != ADAYOLO_ACT_SILU) return ADAYOLO_EINVAL;
    ConvArgs a;
    a.in = static_cast<const unsigned short*>(in); a.in_cs = in_cstride;
    a.w = static_cast<const unsigned short*>(weight); a.bias = bias;
    a.res = static_cast<const unsigned short*>(residual); a.res_cs = res_cstride;
    a.out = static_cast<unsigned short*>(out); a.out_cs = out_cstride;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.ks = ksize; a.stride = stride; a.pad = ksize / 2; a.act = act;
    a.Ho = (H + 2 * a.pad - ksize) / stride + 1;
    a.Wo = (W + 2 * a.pad - ksize) / stride + 1;
    const long M = (long)B * a.Ho * a.Wo;
    if (M > 0x7fffffffL || (long)B * H * W * in_cstride > 0x7fffffffffL) return ADAYOLO_ESHAPE;
    a.M = (int)M; a.mtiles = a.ntiles = 0;
    magic_div((unsigned)(a.Ho * a.Wo), &a.magic_hw, &a.sh_hw);
    magic_div((unsigned)a.Wo, &a.magic_w, &a.sh_w);
    a.w2 = static_cast<const unsigned short*>(weight2); a.bias2 = bias2;
    a.out2 = static_cast<unsigned short*>(out2); a.out2_cs = out2_cstride;
    a.pre = nullptr; a.pre_cs = 0;
    return launch_conv_pp(a, static_cast<hipStream_t>(stream), 50) == hipSuccess ? ADAYOLO_OK : ADAYOLO_ELAUNCH;
}

int adayolo_conv1x1_stream_fwd(const void* in, int in_cstride, const void* weight_fragments, const float* bias, void* out,
                               int out_cstride, int B, int H, int W, int Cin, int Cout, int act, void* stream) {
    ConvArgs a;
    const int rc = conv_args(a, in, in_cstride, weight_fragments, bias, nullptr, 0, out, out_cstride, B, H, W, Cin, Cout, 1, 1, act);
    if (rc != ADAYOLO_OK) return rc;
    if (in == out) return ADAYOLO_EINVAL;
    const hipError_t e = launch_conv_k1(a, static_cast<hipStream_t>(stream));
    if (e == hipErrorInvalidValue) return ADAYOLO_ESHAPE;                 // Cin not in {256, 512} or Cout % 256 != 0
    return e == hipSuccess ? ADAYOLO_OK : ADAYOLO_ELAUNCH;
}

// ---- persistent chain: host-side tables ------------------------------------------------------------------------------------
namespace {
struct ChainPlan {
    std::vector<ConvArgs> layers;
    std::vector<ChainHead> heads;
    std::vector<ChainDeps> deps;
    int ndone = 0;
    size_t off_layers = 0, off_heads = 0, off_deps = 0, bytes = 0;
};
struct Span { const char* lo; const char* hi; };
inline bool overlap(const Span& x, const Span& y) { return x.lo && y.lo && x.lo < y.hi && y.lo < x.hi; }
inline size_t up64(size_t v) { return (v + 63) & ~(size_t)63; }

// ADAYOLO_OK and the tables, or the reason this list is not served
int chain_plan(const adayolo_chain_layer* L, int n, ChainPlan& P, bool tables) {
    if (!L || n < 1 || n > 64) return ADAYOLO_EINVAL;
    P.layers.resize(n);
    std::vector<int> flag_base(n), in_src(n, -1), res_src(n, -1);
    std::vector<Span> s_in(n), s_res(n), s_out(n), s_out2(n);
    int total = 0;
    P.ndone = 0;
    for (int l = 0; l < n; ++l) {
        const adayolo_chain_layer& d = L[l];
        ConvArgs& a = P.layers[l];
        const int rc = conv_args(a, d.in, d.in_cstride, d.weight, d.bias, d.residual, d.res_cstride, d.out, d.out_cstride, d.B, d.H,
                                 d.W, d.Cin, d.Cout, d.ksize, d.stride, d.act);
        if (rc != ADAYOLO_OK) return rc;
        if (d.tile != 0 && d.tile != 1) return ADAYOLO_EINVAL;
        const int bn = d.tile == 1 ? 128 : 256;
        if (a.Cin % 64 || a.Cout % bn) return ADAYOLO_ESHAPE;
        a.chain_tile = d.tile;
        if (d.weight2) {
            if (d.tile != 0) return ADAYOLO_ESHAPE;
            if (!d.bias2 || !d.out2) return ADAYOLO_EINVAL;
            if (a.Cout != 256 || d.Cout2 != 128 || d.out2_cstride % 8 || d.out2_cstride < d.Cout2) return ADAYOLO_ESHAPE;
            a.w2 = static_cast<const unsigned short*>(d.weight2); a.bias2 = d.bias2;
            a.out2 = static_cast<unsigned short*>(d.out2); a.out2_cs = d.out2_cstride;
        }
        a.mtiles = (a.M + 255) / 256;
        a.ntiles = a.Cout / bn;
        // the written-through stores address their tensor with 32-bit byte offsets
        if ((long)a.M * a.out_cs * 2 >= 0x7FFFFFFFL || (a.out2 && (long)a.M * a.out2_cs * 2 >= 0x7FFFFFFFL)) return ADAYOLO_ESHAPE;
        // the bytes a tensor view really touches: `rows` rows of `c` channels at a stride of `cs` — the LAST row ends after its c
        // channels, not after a whole stride (a channel slice that ends its parent buffer would otherwise reach (cs - c) elements
        // into whatever the allocator placed behind it: in a long-lived process with a split cached block that is another tensor
        // of the same chain, and the chain was refused for an overlap that does not exist — round 6)
        auto span = [](const void* p, long rows, int cs, int c) {
            return p ? Span{(const char*)p, (const char*)p + ((rows - 1) * cs + c) * 2} : Span{nullptr, nullptr};
        };
        s_in[l] = span(a.in, (long)a.B * a.H * a.W, a.in_cs, a.Cin);
        s_res[l] = span(a.res, a.M, a.res_cs, a.Cout);
        s_out[l] = span(a.out, a.M, a.out_cs, a.Cout);
        s_out2[l] = span(a.out2, a.M, a.out2_cs, d.weight2 ? d.Cout2 : 0);
        if (overlap(s_out[l], s_in[l]) || overlap(s_out[l], s_res[l]) || overlap(s_out2[l], s_in[l]) || overlap(s_out2[l], s_res[l]) ||
            overlap(s_out[l], s_out2[l]))
            return ADAYOLO_ESHAPE;
        for (int k = l - 1; k >= 0; --k) {
            const ConvArgs& p = P.layers[k];
            // a tensor of this layer may meet an earlier layer's OUTPUT only as that layer's exact output (a dependency)
            auto link = [&](const unsigned short* ptr, int cs, long rows, const Span& sp, int& src) -> bool {
                if (!ptr) return true;
                const bool m_out = ptr == p.out && cs == p.out_cs && rows == p.M;
                const bool m_out2 = p.out2 && ptr == p.out2 && cs == p.out2_cs && rows == p.M;
                if (m_out || m_out2) { if (src < 0) src = k; return true; }
                return !(overlap(sp, s_out[k]) || overlap(sp, s_out2[k]));
            };
            if (!link(a.in, a.in_cs, (long)a.B * a.H * a.W, s_in[l], in_src[l])) return ADAYOLO_ESHAPE;
            if (!link(a.res, a.res_cs, a.M, s_res[l], res_src[l])) return ADAYOLO_ESHAPE;
            // ... and this layer's outputs meet nothing an earlier layer reads or writes
            for (const Span* o : {&s_out[l], &s_out2[l]})
                if (overlap(*o, s_in[k]) || overlap(*o, s_res[k]) || overlap(*o, s_out[k]) || overlap(*o, s_out2[k])) return ADAYOLO_ESHAPE;
        }
        flag_base[l] = P.ndone;
        P.ndone += a.mtiles;
        total += a.mtiles * a.ntiles;
    }
    P.off_layers = up64(64 + (size_t)P.ndone * 4);
    P.off_heads = up64(P.off_layers + (size_t)n * sizeof(ConvArgs));
    P.off_deps = up64(P.off_heads + (size_t)total * sizeof(ChainHead));
    P.bytes = up64(P.off_deps + (size_t)total * sizeof(ChainDeps));
    if (P.bytes >= 0x7FFFFFFFu) return ADAYOLO_ESHAPE;
    P.heads.clear(); P.deps.clear();
    if (!tables) return ADAYOLO_OK;                                  // (only the sizes are needed)
    P.heads.reserve(total); P.deps.reserve(total);
    // Hand-out order of a layer's m-tiles. ADAYOLO_CHAIN_ORDER=snake (measurement): the m-tiles are cut into 8 contiguous chunks (one
    // per XCD, the launch-per-layer kernels' xcd_remap), handed out round-robin over the chunks — workgroup b of the launch sits
    // on XCD b % 8 and draws item ~b first, and workgroups that finish in their starting order keep drawing "their" chunk —
    // ascending inside even chunks, descending inside odd ones, so that the two tiles on either side of a chunk boundary are
    // handed out at the same end of the layer (a consumer next to a boundary does not wait for a producer handed out 450 items
    // after its own neighbours). Default: ascending.
    static const bool snake = [] { const char* e = getenv("ADAYOLO_CHAIN_ORDER"); return e && !strcmp(e, "snake"); }();
    for (int l = 0; l < n; ++l) {
        const ConvArgs& a = P.layers[l];
        std::vector<int> order(a.mtiles);
        for (int i = 0; i < a.mtiles; ++i) order[i] = i;
        if (snake && a.mtiles >= 16) {
            const int q = a.mtiles / 8, r = a.mtiles % 8;
            int start[9];
            start[0] = 0;
            for (int c = 0; c < 8; ++c) start[c + 1] = start[c] + q + (c < r ? 1 : 0);
            int k = 0;
            for (int pos = 0; pos <= q; ++pos)
                for (int c = 0; c < 8; ++c) {
                    const int size = start[c + 1] - start[c];
                    if (pos < size) order[k++] = (c & 1) ? start[c + 1] - 1 - pos : start[c] + pos;
                }
        }
        for (int it = 0; it < a.mtiles * a.ntiles; ++it) {
            const int mt = order[it / a.ntiles], lid = mt * a.ntiles + it % a.ntiles;
            const int m0 = mt * 256, m1 = std::min(m0 + 255, a.M - 1);
            ChainDeps dp{0, 0, 0, 0};
            if (in_src[l] >= 0) {
                const ConvArgs& p = P.layers[in_src[l]];
                const int hw = a.Ho * a.Wo;
                const int b0 = m0 / hw, ho0 = (m0 % hw) / a.Wo, wo0 = m0 % a.Wo, b1 = m1 / hw, ho1 = (m1 % hw) / a.Wo, wo1 = m1 % a.Wo;
                // the lowest / highest input pixel (flat index) any tap of the tile reads: the first pixel's top-left tap and the
                // last pixel's bottom-right tap — later output rows only reach further down, earlier ones further up — unless that
                // tap's row is outside the image (clamped): then a neighbouring output row's taps on the border row may reach
                // further along it, and the whole border row is taken
                const int top = ho0 * a.stride - a.pad, bot = ho1 * a.stride - a.pad + a.ks - 1;
                const int r_lo = std::max(top, 0), r_hi = std::min(bot, a.H - 1);
                const int c_lo = top >= 0 ? std::max(wo0 * a.stride - a.pad, 0) : 0;
                const int c_hi = bot <= a.H - 1 ? std::min(wo1 * a.stride - a.pad + a.ks - 1, a.W - 1) : a.W - 1;
                const long px_lo = ((long)b0 * a.H + r_lo) * a.W + c_lo, px_hi = ((long)b1 * a.H + r_hi) * a.W + c_hi;
                const int t_lo = (int)(px_lo / 256), t_hi = std::min((int)(px_hi / 256), p.mtiles - 1);
                const int cnt = t_hi - t_lo + 1;
                if (cnt < 1 || cnt > 32 || p.ntiles > 0xFFFF) return ADAYOLO_ESHAPE;
                dp.in_lo = flag_base[in_src[l]] + t_lo;
                dp.in_n_need = (cnt << 16) | p.ntiles;
            }
            if (res_src[l] >= 0) {
                const ConvArgs& p = P.layers[res_src[l]];
                const int t_lo = m0 / 256, t_hi = std::min(m1 / 256, p.mtiles - 1);
                dp.res_lo = flag_base[res_src[l]] + t_lo;
                dp.res_n_need = ((t_hi - t_lo + 1) << 16) | p.ntiles;
            }
            P.heads.push_back(ChainHead{l, lid, flag_base[l] + mt, 0});
            P.deps.push_back(dp);
        }
    }
    return ADAYOLO_OK;
}

int device_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
            v = 256;
        cus = v;
    }
    return cus;
}
}  // namespace

size_t adayolo_conv_chain_workspace_bytes(const adayolo_chain_layer* layers, int n) {
    ChainPlan P;                                         // (with the tables: a dependency window the kernel cannot hold is "not served")
    return chain_plan(layers, n, P, true) == ADAYOLO_OK ? P.bytes : 0;
}

int adayolo_conv_chain_tables(const adayolo_chain_layer* layers, int n, void* host_image, size_t bytes, int32_t* info) {
    ChainPlan P;
    const int rc = chain_plan(layers, n, P, true);
    if (rc != ADAYOLO_OK) return rc;
    if (!host_image || bytes < P.bytes) return ADAYOLO_EINVAL;
    unsigned char* img = static_cast<unsigned char*>(host_image);
    memset(img, 0, P.bytes);
    memcpy(img + P.off_layers, P.layers.data(), P.layers.size() * sizeof(ConvArgs));
    memcpy(img + P.off_heads, P.heads.data(), P.heads.size() * sizeof(ChainHead));
    memcpy(img + P.off_deps, P.deps.data(), P.deps.size() * sizeof(ChainDeps));
    if (info) {
        info[0] = (int32_t)P.heads.size(); info[1] = P.ndone; info[2] = (int32_t)P.off_layers; info[3] = (int32_t)P.off_heads;
        info[4] = (int32_t)P.off_deps; info[5] = (int32_t)sizeof(ConvArgs);
    }
    return ADAYOLO_OK;
}

// What `prepare` leaves on the host for a workspace: the launch arguments, a fingerprint of the layer list they were built from
// and the pinned word the kernel mirrors a give-up code to. `fwd` is then a checked launch: one map lookup + one hash of the
// caller's list, no tables, no allocation.
namespace {
struct ChainDesc {
    ChainArgs args;
    unsigned long long fingerprint;
    size_t bytes;
};
std::mutex g_chain_mu;
std::unordered_map<const void*, ChainDesc> g_chain_desc;
int* g_chain_flags = nullptr;                 // one pinned page: 1024 give-up words, handed out round-robin per prepare
int g_chain_flags_next = 0;

unsigned long long chain_fingerprint(const adayolo_chain_layer* L, int n) {
    unsigned long long h = 1469598103934665603ull;             // FNV-1a over the FIELDS (struct padding is not the caller's to define)
    auto mix = [&](unsigned long long v) { for (int i = 0; i < 8; ++i) { h ^= (v >> (8 * i)) & 0xFF; h *= 1099511628211ull; } };
    mix((unsigned long long)n);
    for (int l = 0; l < n; ++l) {
        const adayolo_chain_layer& d = L[l];
        for (const void* p : {d.in, d.weight, (const void*)d.bias, d.residual, (const void*)d.out, d.weight2, (const void*)d.bias2,
                              (const void*)d.out2})
            mix((unsigned long long)(uintptr_t)p);
        for (int v : {d.in_cstride, d.res_cstride, d.out_cstride, d.B, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.act,
                      d.out2_cstride, d.Cout2, d.tile})
            mix((unsigned long long)(unsigned)v);
    }
    return h;
}
}  // namespace

int adayolo_conv_chain_prepare(const adayolo_chain_layer* layers, int n, void* workspace, size_t workspace_bytes) {
    ChainPlan P;
    const int rc = chain_plan(layers, n, P, true);
    if (rc != ADAYOLO_OK) return rc;
    if (!workspace || workspace_bytes < P.bytes || ((uintptr_t)workspace & 63)) return ADAYOLO_EINVAL;
    std::vector<unsigned char> img(P.bytes, 0);
    memcpy(img.data() + P.off_layers, P.layers.data(), P.layers.size() * sizeof(ConvArgs));
    memcpy(img.data() + P.off_heads, P.heads.data(), P.heads.size() * sizeof(ChainHead));
    memcpy(img.data() + P.off_deps, P.deps.data(), P.deps.size() * sizeof(ChainDeps));
    if (hipMemcpy(workspace, img.data(), P.bytes, hipMemcpyHostToDevice) != hipSuccess) return ADAYOLO_ELAUNCH;
    std::lock_guard<std::mutex> lock(g_chain_mu);
    if (!g_chain_flags) {
        void* p = nullptr;
        if (hipHostMalloc(&p, 4096, hipHostMallocMapped) != hipSuccess) return ADAYOLO_ELAUNCH;
        memset(p, 0, 4096);
        g_chain_flags = static_cast<int*>(p);
    }
    ChainDesc d;
    d.args.ws = static_cast<unsigned char*>(workspace);
    auto old = g_chain_desc.find(workspace);
    d.args.host_err = old != g_chain_desc.end() ? old->second.args.host_err : g_chain_flags + (g_chain_flags_next++ & 1023);
    *d.args.host_err = 0;
    d.args.off_layers = (int)P.off_layers; d.args.off_heads = (int)P.off_heads; d.args.off_deps = (int)P.off_deps;
    d.args.total = (int)P.heads.size(); d.args.ndone = P.ndone; d.args.stagger = 0;
    d.fingerprint = chain_fingerprint(layers, n);
    d.bytes = P.bytes;
    g_chain_desc[workspace] = d;
    return ADAYOLO_OK;
}

int adayolo_conv_chain_fwd(const adayolo_chain_layer* layers, int n, void* workspace, size_t workspace_bytes, void* stream) {
    if (!layers || n < 1 || !workspace) return ADAYOLO_EINVAL;
    ChainArgs c;
    {
        std::lock_guard<std::mutex> lock(g_chain_mu);
        auto it = g_chain_desc.find(workspace);
        // the workspace must have been prepared for THIS layer list: tables of another list would be read at wrong offsets
        if (it == g_chain_desc.end() || it->second.fingerprint != chain_fingerprint(layers, n) || workspace_bytes < it->second.bytes)
            return ADAYOLO_EINVAL;
        c = it->second.args;
    }
    static const int stagger = [] { const char* e = getenv("ADAYOLO_CHAIN_STAGGER"); return e ? atoi(e) : 0; }();
    // ADAYOLO_CHAIN_GRID (measurement): fewer persistent workgroups than CUs — what is left over is free for another stream's kernels
    static const int grid_cap = [] { const char* e = getenv("ADAYOLO_CHAIN_GRID"); return e ? atoi(e) : 0; }();
    c.stagger = stagger;
    const int grid = grid_cap > 0 && grid_cap < device_cus() ? grid_cap : device_cus();
    return launch_conv_chain(c, grid, static_cast<hipStream_t>(stream)) == hipSuccess ? ADAYOLO_OK : ADAYOLO_ELAUNCH;
}

int adayolo_conv_chain_status(const void* workspace) {
    if (!workspace) return ADAYOLO_EINVAL;
    int w[4] = {0, 0, 0, 0};                             // head, err (this launch), exit, err (sticky: the last launch that gave up)
    if (hipDeviceSynchronize() != hipSuccess) return ADAYOLO_ELAUNCH;
    if (hipMemcpy(w, workspace, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess) return ADAYOLO_ELAUNCH;
    return w[3] ? w[3] : w[1];
}

int adayolo_conv_chain_poll(const void* workspace) {
    std::lock_guard<std::mutex> lock(g_chain_mu);
    auto it = g_chain_desc.find(workspace);
    if (it == g_chain_desc.end()) return ADAYOLO_EINVAL;
    return __atomic_load_n(it->second.args.host_err, __ATOMIC_RELAXED);       // pinned host memory: no device call
}

int adayolo_bottleneck256_fwd(const void* x, int x_cstride, const void* weight1, const float* bias1, const void* weight2,
                              const float* bias2, void* out, int out_cstride, int B, int H, int W, void* stream) {
    if (!x || !weight1 || !bias1 || !weight2 || !bias2 || !out || x == out) return ADAYOLO_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0) return ADAYOLO_EINVAL;
    if (x_cstride % 8 || out_cstride % 8 || x_cstride < 256 || out_cstride < 256) return ADAYOLO_ESHAPE;
    if ((long)B * H * W * (x_cstride > out_cstride ? x_cstride : out_cstride) > 0x7fffffffffL ||
        (long)B * ((H + 15) / 16) * ((W + 15) / 16) > 0x7fffffffL)
        return ADAYOLO_ESHAPE;
    return launch_bottleneck256(x, x_cstride, weight1, bias1, weight2, bias2, out, out_cstride, B, H, W,
                                static_cast<hipStream_t>(stream)) == hipSuccess ? ADAYOLO_OK : ADAYOLO_ELAUNCH;
}

int adayolo_bottleneck_ws_fwd(const void* x, int x_cstride, const void* weight1, const float* bias1, const void* weight2,
                              const float* bias2, void* out, int out_cstride, int B, int H, int W, int C, void* stream) {
    if (!x || !weight1 || !bias1 || !weight2 || !bias2 || !out || x == out) return ADAYOLO_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0) return ADAYOLO_EINVAL;
    if ((C != 64 && C != 128) || x_cstride % 8 || out_cstride % 8 || x_cstride < C || out_cstride < C) return ADAYOLO_ESHAPE;
    const hipError_t e = launch_bottleneck_ws(x, x_cstride, weight1, bias1, weight2, bias2, out, out_cstride, B, H, W, C,
                                              static_cast<hipStream_t>(stream));
    if (e == hipErrorInvalidValue) return ADAYOLO_ESHAPE;                  // a tensor beyond 32-bit byte offsets
    return e == hipSuccess ? ADAYOLO_OK : ADAYOLO_ELAUNCH;
}

int adayolo_conv_keep_fwd(const void* in, int in_cstride, const void* weight, const float* bias, const void* residual,
                          int res_cstride, void* out, int out_cstride, void* pre, int pre_cstride, int B, int H, int W, int Cin,
                          int Cout, int ksize, int stride, int act, int variant, void* stream) {
    if (!pre) return ADAYOLO_EINVAL;
    ConvArgs a;
    const int rc = conv_args(a, in, in_cstride, weight, bias, residual, res_cstride, out, out_cstride, B, H, W, Cin, Cout,
                             ksize, stride, act);
    if (rc != ADAYOLO_OK) return rc;
    if (pre_cstride % 8 || pre_cstride < Cout) return ADAYOLO_ESHAPE;
    const bool dma2 = variant == 5 || variant == 22 || variant == 26 || variant == 27;
    const bool pq = variant == 80 || variant == 85;
    if (!dma2 && !pq && variant != 60) return ADAYOLO_EINVAL;       // the kernels whose epilogue has the second output
    a.pre = static_cast<unsigned short*>(pre); a.pre_cs = pre_cstride;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const hipError_t e = dma2 ? launch_conv_dma2(a, s, variant) : pq ? launch_conv_pq(a, s, variant) : launch_conv_pp128(a, s, variant);
    if (e == hipErrorInvalidValue) return ADAYOLO_ESHAPE;    // this kernel does not serve the shape: the caller keeps two launches
    return e == hipSuccess ? ADAYOLO_OK : ADAYOLO_ELAUNCH;
}

size_t adayolo_conv_splitk_workspace_bytes(int B, int H, int W, int Cin, int Cout, int ksize, int stride, int variant) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || ksize < 1 || ksize > 3 || (stride != 1 && stride != 2))
        return 0;
    if (variant < ADAYOLO_SPLITK_BASE + 2 || variant > ADAYOLO_SPLITK_BASE + 16) return 0;
    ConvArgs a;
    a.Cin = Cin; a.Cout = Cout; a.ks = ksize;
    const int pad = ksize / 2;
    // ksize 2 = the stride-2 data gradient's form (adayolo_conv_s2grad_fwd: H x W is its Ho x Wo grid, Cout = 4 x channels)
    const long M = ksize == 2 ? (long)B * H * W
                              : (long)B * ((H + 2 * pad - ksize) / stride + 1) * ((W + 2 * pad - ksize) / stride + 1);
    if (M > 0x7fffffffL) return 0;
    a.M = (int)M;
    return conv_pp128_splitk_bytes(a, variant - ADAYOLO_SPLITK_BASE);
}

int adayolo_conv_splitk_fwd(const void* in, int in_cstride, const void* weight, const float* bias, const void* residual,
                            int res_cstride, void* out, int out_cstride, void* pre, int pre_cstride, int B, int H, int W,
                            int Cin, int Cout, int ksize, int stride, int act, int variant, void* workspace,
                            size_t workspace_bytes, void* stream) {
    if (!workspace) return ADAYOLO_EINVAL;
    if (variant < ADAYOLO_SPLITK_BASE + 2 || variant > ADAYOLO_SPLITK_BASE + 16) return ADAYOLO_EINVAL;
    ConvArgs a;
    const int rc = conv_args(a, in, in_cstride, weight, bias, residual, res_cstride, out, out_cstride, B, H, W, Cin, Cout,
                             ksize, stride, act);
    if (rc != ADAYOLO_OK) return rc;
    if (pre && (pre_cstride % 8 || pre_cstride < Cout)) return ADAYOLO_ESHAPE;
    a.pre = static_cast<unsigned short*>(pre); a.pre_cs = pre ? pre_cstride : 0;
    const size_t need = conv_pp128_splitk_bytes(a, variant - ADAYOLO_SPLITK_BASE);
    if (need == 0) return ADAYOLO_ESHAPE;                    // this split does not serve the shape
    if (workspace_bytes < need) return ADAYOLO_EINVAL;
    const hipError_t e = launch_conv_pp128_splitk(a, static_cast<hipStream_t>(stream), variant - ADAYOLO_SPLITK_BASE, workspace,
                                                  workspace_bytes);
    return e == hipSuccess ? ADAYOLO_OK : ADAYOLO_ELAUNCH;
}

int adayolo_conv_dsilu_fwd(const void* in, int in_cstride, const void* weight, const float* bias, const void* residual,
                           int res_cstride, void* out, int out_cstride, const void* pre, int pre_cstride, void* grad_pre,
                           int gp_cstride, int B, int H, int W, int Cin, int Cout, int ksize, int stride, int variant,
                           void* workspace, size_t workspace_bytes, void* stream) {
    if (!pre || !grad_pre) return ADAYOLO_EINVAL;
    ConvArgs a;
    const int rc = conv_args(a, in, in_cstride, weight, bias, residual, res_cstride, out ? out : grad_pre,
                             out ? out_cstride : gp_cstride, B, H, W, Cin, Cout, ksize, stride, ADAYOLO_ACT_NONE);
    if (rc != ADAYOLO_OK) return rc;
    if (pre_cstride % 8 || pre_cstride < Cout || gp_cstride % 8 || gp_cstride < Cout) return ADAYOLO_ESHAPE;
    a.out = static_cast<unsigned short*>(out); a.out_cs = out ? out_cstride : 0;
    a.pre = static_cast<unsigned short*>(const_cast<void*>(pre)); a.pre_cs = pre_cstride;      // an input here
    a.gpre = static_cast<unsigned short*>(grad_pre); a.gpre_cs = gp_cstride;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e;
    if (variant >= ADAYOLO_SPLITK_BASE + 2 && variant <= ADAYOLO_SPLITK_BASE + 16) {
        const size_t need = conv_pp128_splitk_bytes(a, variant - ADAYOLO_SPLITK_BASE);
        if (need == 0) return ADAYOLO_ESHAPE;
        if (!workspace || workspace_bytes < need) return ADAYOLO_EINVAL;
        e = launch_conv_pp128_splitk(a, s, variant - ADAYOLO_SPLITK_BASE, workspace, workspace_bytes);
    } else if (variant == 5 || variant == 22 || variant == 26 || variant == 27) {
        e = launch_conv_dma2(a, s, variant);
    } else if (variant == 60) {
        e = launch_conv_pp128(a, s, variant);
    } else if (variant == 80 || variant == 85) {
        e = launch_conv_pq(a, s, variant);
    } else {
        return ADAYOLO_EINVAL;                               // the kernels whose epilogue has this form
    }
    if (e == hipErrorInvalidValue) return ADAYOLO_ESHAPE;    // the named kernel does not serve the shape
    return e == hipSuccess ? ADAYOLO_OK : ADAYOLO_ELAUNCH;
}

int adayolo_conv_s2grad_fwd(const void* grad_out, int go_cstride, const void* weight4, const float* bias4, const void* residual,
                            int res_cstride, void* grad_in, int gi_cstride, const void* pre, int pre_cstride, void* grad_pre,
                            int gp_cstride, int B, int Ho, int Wo, int Cout, int Cin, int variant, void* workspace,
                            size_t workspace_bytes, void* stream) {
    if (!grad_out || !weight4 || !bias4 || (!grad_in && !grad_pre) || (!pre != !grad_pre)) return ADAYOLO_EINVAL;
    if (B <= 0 || Ho <= 0 || Wo <= 0 || Cin <= 0 || Cout <= 0) return ADAYOLO_EINVAL;
    if (Cin % 8 || Cout % 8 || go_cstride % 8 || go_cstride < Cout) return ADAYOLO_ESHAPE;
    if (grad_in && (gi_cstride % 8 || gi_cstride < Cin)) return ADAYOLO_ESHAPE;
    if (residual && (res_cstride % 8 || res_cstride < Cin)) return ADAYOLO_ESHAPE;
    if (pre && (pre_cstride % 8 || pre_cstride < Cin || gp_cstride % 8 || gp_cstride < Cin)) return ADAYOLO_ESHAPE;
    const long M = (long)B * Ho * Wo;
    if (4 * M > 0x7fffffffL || 4 * M * (gi_cstride > gp_cstride ? gi_cstride : gp_cstride) > 0x7fffffffffL) return ADAYOLO_ESHAPE;
    ConvArgs a;                                               // a 2x2 stride-1 conv over the Ho x Wo grid, zero beyond its far edges
    a.in = static_cast<const unsigned short*>(grad_out); a.in_cs = go_cstride;
    a.w = static_cast<const unsigned short*>(weight4); a.bias = bias4;
    a.res = static_cast<const unsigned short*>(residual); a.res_cs = residual ? res_cstride : 0;
    a.out = static_cast<unsigned short*>(grad_in); a.out_cs = grad_in ? gi_cstride : 0;
    a.B = B; a.H = Ho; a.W = Wo; a.Cin = Cout; a.Cout = 4 * Cin;
    a.ks = 2; a.stride = 1; a.pad = 0; a.act = ADAYOLO_ACT_NONE;
    a.Ho = Ho; a.Wo = Wo; a.M = (int)M; a.mtiles = a.ntiles = 0;
    a.w2 = nullptr; a.bias2 = nullptr; a.out2 = nullptr; a.out2_cs = 0;
    a.pre = static_cast<unsigned short*>(const_cast<void*>(pre)); a.pre_cs = pre ? pre_cstride : 0;
    a.gpre = static_cast<unsigned short*>(grad_pre); a.gpre_cs = grad_pre ? gp_cstride : 0;
    a.d2s_c = Cin;
    magic_div((unsigned)(Ho * Wo), &a.magic_hw, &a.sh_hw);
    magic_div((unsigned)Wo, &a.magic_w, &a.sh_w);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e;
    if (variant >= ADAYOLO_SPLITK_BASE + 2 && variant <= ADAYOLO_SPLITK_BASE + 16) {
        const size_t need = conv_pp128_splitk_bytes(a, variant - ADAYOLO_SPLITK_BASE);
        if (need == 0) return ADAYOLO_ESHAPE;
        if (!workspace || workspace_bytes < need) return ADAYOLO_EINVAL;
        e = launch_conv_pp128_splitk(a, s, variant - ADAYOLO_SPLITK_BASE, workspace, workspace_bytes);
    } else if (variant == 5 || variant == 22 || variant == 26 || variant == 27) {
        e = launch_conv_dma2(a, s, variant);
    } else if (variant == 60) {
        e = launch_conv_pp128(a, s, variant);
    } else {
        return ADAYOLO_EINVAL;
    }
    if (e == hipErrorInvalidValue) return ADAYOLO_ESHAPE;
    return e == hipSuccess ? ADAYOLO_OK : ADAYOLO_ELAUNCH;
}

int adayolo_conv_fwd(const void* in, int in_cstride, const void* weight, const float* bias, const void* residual,
                     int res_cstride, void* out, int out_cstride, int B, int H, int W, int Cin, int Cout, int ksize,
                     int stride, int act, void* stream) {
    return adayolo_conv_fwd_variant(in, in_cstride, weight, bias, residual, res_cstride, out, out_cstride, B, H, W,
                                    Cin, Cout, ksize, stride, act, 0, stream);
}

int adayolo_stem_fwd(const float* img, const float* weight, const float* bias, void* out, int out_cstride, int B,
                     int H, int W, int Hp, int pad_top, float pad_value, int Cout, void* stream) {
    if (!img || !weight || !bias || !out) return ADAYOLO_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0 || Hp < H || pad_top < 0 || pad_top + H > Hp) return ADAYOLO_EINVAL;
    if (Cout != 32 || out_cstride % 8 || out_cstride < Cout || B > 65535) return ADAYOLO_ESHAPE;
    return launch_stem(img, weight, bias, out, out_cstride, B, H, W, Hp, pad_top, pad_value, ADAYOLO_ACT_SILU,
                       static_cast<hipStream_t>(stream)) == hipSuccess ? ADAYOLO_OK : ADAYOLO_ELAUNCH;
}

int adayolo_stem_fwd_act(const float* img, const float* weight, const float* bias, void* out, int out_cstride, int B,
                         int H, int W, int Hp, int pad_top, float pad_value, int Cout, int act, void* stream) {
    if (!img || !weight || !bias || !out) return ADAYOLO_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0 || Hp < H || pad_top < 0 || pad_top + H > Hp) return ADAYOLO_EINVAL;
    if (act != ADAYOLO_ACT_NONE && act != ADAYOLO_ACT_SILU) return ADAYOLO_EINVAL;
    if (Cout != 32 || out_cstride % 8 || out_cstride < Cout || B > 65535) return ADAYOLO_ESHAPE;
    return launch_stem(img, weight, bias, out, out_cstride, B, H, W, Hp, pad_top, pad_value, act,
                       static_cast<hipStream_t>(stream)) == hipSuccess ? ADAYOLO_OK : ADAYOLO_ELAUNCH;
}

int adayolo_stem_keep_fwd(const float* img, const float* weight, const float* bias, void* out, int out_cstride, void* pre,
                          int pre_cstride, int B, int H, int W, int Hp, int pad_top, float pad_value, int Cout, void* stream) {
    if (!img || !weight || !bias || !out || !pre) return ADAYOLO_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0 || Hp < H || pad_top < 0 || pad_top + H > Hp) return ADAYOLO_EINVAL;
    if (Cout != 32 || out_cstride % 8 || out_cstride < Cout || pre_cstride % 8 || pre_cstride < Cout || B > 65535) return ADAYOLO_ESHAPE;
    return launch_stem(img, weight, bias, out, out_cstride, B, H, W, Hp, pad_top, pad_value, ADAYOLO_ACT_SILU,
                       static_cast<hipStream_t>(stream), pre, pre_cstride) == hipSuccess ? ADAYOLO_OK : ADAYOLO_ELAUNCH;
}

int adayolo_letterbox_pack(const float* img, void* out, int out_cstride, int B, int H, int W, int Hp, int pad_top,
                           float pad_value, void* stream) {
    if (!img || !out) return ADAYOLO_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0 || Hp < H || pad_top < 0 || pad_top + H > Hp) return ADAYOLO_EINVAL;
    if (out_cstride % 8 || out_cstride < 8 || B > 65535 || Hp > 65535) return ADAYOLO_ESHAPE;
    return launch_letterbox_pack(img, out, out_cstride, B, H, W, Hp, pad_top, pad_value,
                                 static_cast<hipStream_t>(stream)) == hipSuccess ? ADAYOLO_OK : ADAYOLO_ELAUNCH;
}

int adayolo_stem_down_fwd(const float* img, const float* w_stem, const float* b_stem, const void* w_down,
                          const float* b_down, void* out, int out_cstride, int B, int H, int W, int Hp, int pad_top,
                          float pad_value, const void* w_next, const float* b_next, void* out_next, int out_next_cstride,
                          void* stream) {
    if (!img || !w_stem || !b_stem || !w_down || !b_down || !out) return ADAYOLO_EINVAL;
    if (w_next && (!b_next || !out_next)) return ADAYOLO_EINVAL;
    if (w_next && (out_next_cstride % 8 || out_next_cstride < 32)) return ADAYOLO_ESHAPE;
    if (B <= 0 || H <= 0 || W <= 0 || Hp < H || pad_top < 0 || pad_top + H > Hp) return ADAYOLO_EINVAL;
    if ((Hp & 1) || (W & 1) || out_cstride % 8 || out_cstride < 64 || B > 65535) return ADAYOLO_ESHAPE;
    return launch_stem_down(img, w_stem, b_stem, w_down, b_down, out, out_cstride, B, H, W, Hp, pad_top, pad_value,
                            w_next, b_next, out_next, out_next_cstride, static_cast<hipStream_t>(stream)) == hipSuccess ? ADAYOLO_OK : ADAYOLO_ELAUNCH;
}

static bool ok8(int C, int cs) { return C > 0 && C % 8 == 0 && cs % 8 == 0 && cs >= C; }

int adayolo_silu_fwd(const void* pre, int pre_cstride, const void* residual, int res_cstride, void* out,
                     int out_cstride, long npix, int C, void* stream) {
    if (!pre || !out || npix <= 0) return ADAYOLO_EINVAL;
    if (!ok8(C, pre_cstride) || !ok8(C, out_cstride) || (residual && !ok8(C, res_cstride))) return ADAYOLO_ESHAPE;
    return launch_silu_fwd(pre, pre_cstride, residual, res_cstride, out, out_cstride, npix, C,
                           static_cast<hipStream_t>(stream)) == hipSuccess ? ADAYOLO_OK : ADAYOLO_ELAUNCH;
}

int adayolo_silu_bwd(const void* grad_out, int go_cstride, const void* pre, int pre_cstride, void* grad_pre,
                     int gp_cstride, void* grad_res, int gr_cstride, int accumulate_res, long npix, int C, void* stream) {
    if (!grad_out || npix <= 0 || (!grad_pre && !grad_res) || (grad_pre && !pre)) return ADAYOLO_EINVAL;
    if (!ok8(C, go_cstride) || (grad_pre && (!ok8(C, pre_cstride) || !ok8(C, gp_cstride))) ||
        (grad_res && !ok8(C, gr_cstride))) return ADAYOLO_ESHAPE;
    return launch_silu_bwd(grad_out, go_cstride, pre, pre_cstride, grad_pre, gp_cstride, grad_res, gr_cstride,
                           accumulate_res, npix, C, static_cast<hipStream_t>(stream)) == hipSuccess ? ADAYOLO_OK : ADAYOLO_ELAUNCH;
}

int adayolo_zero_insert2x(const void* in, int in_cstride, void* out, int out_cstride, int B, int Ho, int Wo, int H, int W,
                          int C, void* stream) {
    if (!in || !out || B <= 0 || Ho <= 0 || Wo <= 0 || H <= 0 || W <= 0) return ADAYOLO_EINVAL;
    if (!ok8(C, in_cstride) || !ok8(C, out_cstride) || (H + 1) / 2 != Ho || (W + 1) / 2 != Wo) return ADAYOLO_ESHAPE;
    return launch_zero_insert(in, in_cstride, out, out_cstride, B, Ho, Wo, H, W, C, static_cast<hipStream_t>(stream)) ==
                   hipSuccess ? ADAYOLO_OK : ADAYOLO_ELAUNCH;
}

int adayolo_upsample2x_bwd(const void* grad_out, int go_cstride, void* grad_in, int gi_cstride, int accumulate, int B,
                           int H, int W, int C, void* stream) {
    if (!grad_out || !grad_in || B <= 0 || H <= 0 || W <= 0) return ADAYOLO_EINVAL;
    if (!ok8(C, go_cstride) || !ok8(C, gi_cstride)) return ADAYOLO_ESHAPE;
    return launch_upsample_bwd(grad_out, go_cstride, grad_in, gi_cstride, accumulate, B, H, W, C,
                               static_cast<hipStream_t>(stream)) == hipSuccess ? ADAYOLO_OK : ADAYOLO_ELAUNCH;
}

int adayolo_image_grad(const void* grad_nhwc, int g_cstride, float* grad_img, int B, int H, int W, int Hp, int pad_top,
                       void* stream) {
    if (!grad_nhwc || !grad_img || B <= 0 || H <= 0 || W <= 0 || Hp < H || pad_top < 0 || pad_top + H > Hp) return ADAYOLO_EINVAL;
    if (g_cstride < 4 || g_cstride % 2) return ADAYOLO_ESHAPE;
    return launch_image_grad(grad_nhwc, g_cstride, grad_img, B, H, W, Hp, pad_top, static_cast<hipStream_t>(stream)) ==
                   hipSuccess ? ADAYOLO_OK : ADAYOLO_ELAUNCH;
}

int adayolo_upsample2x(const void* in, int in_cstride, void* out, int out_cstride, int B, int H, int W, int C,
                       void* stream) {
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0) return ADAYOLO_EINVAL;
    if (C % 8 || in_cstride % 8 || out_cstride % 8 || in_cstride < C || out_cstride < C) return ADAYOLO_ESHAPE;
    return launch_upsample2x(in, in_cstride, out, out_cstride, B, H, W, C, static_cast<hipStream_t>(stream)) ==
                   hipSuccess ? ADAYOLO_OK : ADAYOLO_ELAUNCH;
}

int adayolo_detect_decode(const void* raw, int raw_cstride, float* pred, int pred_rows, int row_offset,
                          const float* anchors_px, float det_stride, int B, int ny, int nx, int na, int no,
                          void* stream) {
    if (!raw || !pred || !anchors_px || B <= 0 || ny <= 0 || nx <= 0 || na <= 0 || no < 5) return ADAYOLO_EINVAL;
    if (raw_cstride < na * no || row_offset < 0 || row_offset + na * ny * nx > pred_rows) return ADAYOLO_ESHAPE;
    return launch_detect_decode(raw, raw_cstride, pred, pred_rows, row_offset, anchors_px, det_stride, B, ny, nx, na,
                                no, static_cast<hipStream_t>(stream)) == hipSuccess ? ADAYOLO_OK : ADAYOLO_ELAUNCH;
}

size_t adayolo_nms_workspace_bytes(int n) {
    const size_t nb = (size_t)((n > 0 ? n : 0) + 63) / 64;
    return (size_t)(n > 0 ? n : 0) * nb * 8 + 8;
}

int adayolo_nms(const float* boxes_xyxy, int n, float iou_thres, int max_det, void* workspace, int32_t* keep,
                int32_t* num_keep, void* stream) {
    if (n < 0 || max_det <= 0 || !keep || !num_keep || (n > 0 && (!boxes_xyxy || !workspace))) return ADAYOLO_EINVAL;
    if (n > 30000 * 4) return ADAYOLO_ESHAPE;        // LDS bit set of the scan: n/64 words
    if (!(iou_thres >= 0.0f && iou_thres <= 1.0f)) return ADAYOLO_EINVAL;
    return launch_nms(boxes_xyxy, n, iou_thres, max_det, static_cast<unsigned long long*>(workspace), keep, num_keep,
                      static_cast<hipStream_t>(stream)) == hipSuccess ? ADAYOLO_OK : ADAYOLO_ELAUNCH;
}

static int detloss_check(const adayolo_loss_args* a, bool bwd) {
    if (!a || a->nl < 1 || a->nl > 4 || a->B < 1 || a->B > 65535 || a->na < 1 || a->nc < 1 || a->no != a->nc + 5 || !a->loss || !a->ticket)
        return ADAYOLO_EINVAL;
    if (bwd && !a->grad_loss) return ADAYOLO_EINVAL;
    if (a->nc > 128) return ADAYOLO_ESHAPE;                                       // two classes per lane in the per-match pass
    for (int i = 0; i < a->nl; ++i) {
        const adayolo_loss_layer& L = a->layer[i];
        if (!L.raw || !L.tobj || !L.cnt || !L.part || L.n < 0 || (L.n > 0 && (!L.idx || !L.box))) return ADAYOLO_EINVAL;
        if (L.ny < 1 || L.nx < 1 || L.cs < a->na * a->no) return ADAYOLO_ESHAPE;
        if (bwd && (!L.grad || L.grad_cs % 8 || L.grad_cs < a->na * a->no)) return ADAYOLO_ESHAPE;
    }
    return ADAYOLO_OK;
}

int adayolo_detloss_fwd(const adayolo_loss_args* args, void* stream) {
    const int rc = detloss_check(args, false);
    if (rc != ADAYOLO_OK) return rc;
    return launch_detloss_fwd(*args, static_cast<hipStream_t>(stream)) == hipSuccess ? ADAYOLO_OK : ADAYOLO_ELAUNCH;
}

int adayolo_detloss_bwd(const adayolo_loss_args* args, void* stream) {
    const int rc = detloss_check(args, true);
    if (rc != ADAYOLO_OK) return rc;
    return launch_detloss_bwd(*args, static_cast<hipStream_t>(stream)) == hipSuccess ? ADAYOLO_OK : ADAYOLO_ELAUNCH;
}

}  // extern "C"
