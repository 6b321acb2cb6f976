// C-ABI of libadaisp.so (see include/adaisp.h). Argument checking + kernel-family dispatch only:
// never allocates, never synchronises, enqueues on the caller's stream, keeps no global state.
#include "isp_internal.h"

using namespace adaisp;

namespace {

int check_common(const float* img, const float* out, const float* params, int pstride, int B, int H, int W) {
    if (!img || !out || !params) return ADAISP_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0 || pstride <= 0) return ADAISP_EINVAL;
    if (B > 65535) return ADAISP_ESHAPE;
    return ADAISP_OK;
}

bool ranges_overlap(const float* a, const float* b, long n) {
    return (a < b + n) && (b < a + n);
}

}  // namespace

extern "C" {

int adaisp_abi_version(void) { return ADAISP_ABI_VERSION; }

const char* adaisp_strerror(int code) {
    switch (code) {
        case ADAISP_OK: return "ok";
        case ADAISP_EINVAL: return "invalid argument (null pointer, non-positive size or stride)";
        case ADAISP_EOP: return "unknown op code";
        case ADAISP_EALIAS: return "out aliases img for a stencil op";
        case ADAISP_ESHAPE: return "shape not supported by this op";
        case ADAISP_ELAUNCH: return "kernel launch failed";
        default: return "unknown error";
    }
}

int adaisp_num_params(int op) {
    switch (op) {
        case ADAISP_OP_ZERO: return 0;
        case ADAISP_OP_EXPOSURE: case ADAISP_OP_GAMMA: case ADAISP_OP_SHARPEN: case ADAISP_OP_NLM:
        case ADAISP_OP_CONTRAST: case ADAISP_OP_SATPLUS: case ADAISP_OP_WNB: case ADAISP_OP_SHARPEN_V2:
            return 1;
        case ADAISP_OP_USM: return 2;
        case ADAISP_OP_WB: return 3;
        case ADAISP_OP_TONE: return 8;
        case ADAISP_OP_CCM: return 9;
        case ADAISP_OP_COLOR: return 24;
        default: return -1;
    }
}

int adaisp_pool64(const float* img, float* pooled, int B, int H, int W, void* stream) {
    if (!img || !pooled || B <= 0 || H <= 0 || W <= 0) return ADAISP_EINVAL;
    if (B > 65535 || W > 16384) return ADAISP_ESHAPE;
    return launch_pool64(img, pooled, B, H, W, static_cast<hipStream_t>(stream)) == hipSuccess ? ADAISP_OK
                                                                                                : ADAISP_ELAUNCH;
}

int adaisp_pool64_backward(const float* grad_pooled, float* grad_img, int B, int H, int W, void* stream) {
    if (!grad_pooled || !grad_img || B <= 0 || H <= 0 || W <= 0) return ADAISP_EINVAL;
    if ((long)B * 3 > 65535 || H > 65535) return ADAISP_ESHAPE;
    return launch_pool64_bwd(grad_pooled, grad_img, B, H, W, static_cast<hipStream_t>(stream)) == hipSuccess
               ? ADAISP_OK : ADAISP_ELAUNCH;
}

int adaisp_demosaic(const uint16_t* raw, float* out, int B, int H, int W, int pattern, float black_level,
                    float white_level, void* stream) {
    if (!raw || !out || B <= 0 || H <= 0 || W <= 0) return ADAISP_EINVAL;
    if (pattern < 0 || pattern > 3 || !(white_level > black_level)) return ADAISP_EINVAL;
    if ((H & 1) || (W & 1) || H < 2 || W < 2 || B > 65535) return ADAISP_ESHAPE;   // whole 2x2 cells
    return launch_demosaic(raw, out, B, H, W, pattern, black_level, white_level, static_cast<hipStream_t>(stream)) ==
                   hipSuccess ? ADAISP_OK : ADAISP_ELAUNCH;
}

int adaisp_process(int op, const float* img, float* out, const float* params, int param_stride, int B, int H, int W,
                   unsigned flags, void* stream) {
    int rc = check_common(img, out, params, param_stride, B, H, W);
    if (rc) return rc;
    const int np = adaisp_num_params(op);
    if (np < 0) return ADAISP_EOP;
    if (np > param_stride) return ADAISP_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    Batch a{img, out, nullptr, op, params, param_stride, B, H, W, flags};
    hipError_t e;
    if (op_is_pointwise(op)) {
        e = launch_pointwise(a, s);
    } else {
        if (ranges_overlap(img, out, (long)B * 3 * H * W)) return ADAISP_EALIAS;
        if (op == ADAISP_OP_NLM) {
            e = launch_nlm(a, s);
        } else {
            if (H < 3 || W < 3) return ADAISP_ESHAPE;   // valid 3x3 conv / reflect pad 2 need >= 3
            e = launch_conv(a, s);
        }
    }
    return e == hipSuccess ? ADAISP_OK : ADAISP_ELAUNCH;
}

size_t adaisp_nlm_general_workspace_bytes(int B, int H, int W) {
    return (B > 0 && H > 0 && W > 0) ? (size_t)B * H * W * sizeof(float) : 0;
}

int adaisp_nlm_general(const float* img, float* out, const float* h, int h_stride, void* workspace, size_t workspace_bytes,
                       int B, int H, int W, int search_window_size, int patch_size, void* stream) {
    if (!img || !out || !h || !workspace || B <= 0 || H <= 0 || W <= 0 || h_stride <= 0) return ADAISP_EINVAL;
    if (search_window_size < 1 || patch_size < 1 || !(search_window_size & 1) || !(patch_size & 1)) return ADAISP_EINVAL;
    if (search_window_size > 63 || patch_size > 31 || B > 65535 || (H + 3) / 4 > 65535) return ADAISP_ESHAPE;
    if (workspace_bytes < adaisp_nlm_general_workspace_bytes(B, H, W)) return ADAISP_EINVAL;
    if (ranges_overlap(img, out, (long)B * 3 * H * W)) return ADAISP_EALIAS;
    return launch_nlm_general(img, out, h, h_stride, static_cast<float*>(workspace), B, H, W, search_window_size, patch_size,
                              static_cast<hipStream_t>(stream)) == hipSuccess ? ADAISP_OK : ADAISP_ELAUNCH;
}

namespace {

// One RL step for a batch: ids on the device (`filter_id`) or one host-known op (`filter_id` == NULL). With `pooled` the
// next step's 64x64 planes come out of the SAME launches for the pointwise and stencil ops (window-aligned cut, PoolGeom);
// NLM images (its 60 x 24 tiles cannot follow the 11.25-row pool windows without 7 % more tiles) are pooled by a launch
// of their own, which returns at once for every other image — and is not enqueued at all when the host knows the op.
int step_impl(const float* img, float* out, float* pooled, const int32_t* filter_id, int uniform_op, const float* params,
              int param_stride, int B, int H, int W, unsigned flags, hipStream_t s) {
    if (ranges_overlap(img, out, (long)B * 3 * H * W)) return ADAISP_EALIAS;
    if (pooled && W > 16384) return ADAISP_ESHAPE;
    if (H < 3 || W < 3) return ADAISP_ESHAPE;     // as adaisp_process: an image whose op is a stencil could not be served
    Batch a{img, out, filter_id, uniform_op, params, param_stride, B, H, W, flags};
    const bool host_op = filter_id == nullptr;
    const bool pw = !host_op || op_is_pointwise(uniform_op), cv = !host_op || op_is_conv(uniform_op),
               nl = !host_op || uniform_op == ADAISP_OP_NLM;
    const PoolGeom g = pooled ? pool_geom(H, W, img, out) : PoolGeom{0, 0, false};
    // The ids live on the device, so each kernel family is enqueued for the whole batch and its workgroups return at
    // once for images whose op belongs to another family.
    if (pw && (g.ok ? launch_pointwise_pool(a, pooled, g, s) : launch_pointwise(a, s)) != hipSuccess) return ADAISP_ELAUNCH;
    if (cv && (g.ok ? launch_conv_pool(a, pooled, g, s) : launch_conv(a, s)) != hipSuccess) return ADAISP_ELAUNCH;
    if (nl && launch_nlm(a, s) != hipSuccess) return ADAISP_ELAUNCH;
    if (pooled) {
        if (!g.ok) {
            if (launch_pool64(out, pooled, B, H, W, s) != hipSuccess) return ADAISP_ELAUNCH;
        } else if (nl && launch_pool64_sel(out, pooled, filter_id, uniform_op, true, flags, B, H, W, s) != hipSuccess) {
            return ADAISP_ELAUNCH;
        }
    }
    return ADAISP_OK;
}

}  // namespace

int adaisp_forward(const float* img, float* out, float* pooled64_next, const int32_t* filter_id, const float* params,
                   int param_stride, int B, int H, int W, unsigned flags, void* stream) {
    int rc = check_common(img, out, params, param_stride, B, H, W);
    if (rc) return rc;
    if (!filter_id) return ADAISP_EINVAL;
    return step_impl(img, out, pooled64_next, filter_id, 0, params, param_stride, B, H, W, flags,
                     static_cast<hipStream_t>(stream));
}

int adaisp_forward_uniform(int op, const float* img, float* out, float* pooled64_next, const float* params,
                           int param_stride, int B, int H, int W, unsigned flags, void* stream) {
    int rc = check_common(img, out, params, param_stride, B, H, W);
    if (rc) return rc;
    const int np = adaisp_num_params(op);
    if (np < 0) return ADAISP_EOP;
    if (np > param_stride) return ADAISP_EINVAL;
    return step_impl(img, out, pooled64_next, nullptr, op, params, param_stride, B, H, W, flags,
                     static_cast<hipStream_t>(stream));
}

int adaisp_backward_params(const float* img, const float* grad_out, const int32_t* filter_id, const float* params,
                           int param_stride, float* grad_params, int B, int H, int W, unsigned flags, void* stream) {
    if (!img || !grad_out || !filter_id || !params || !grad_params) return ADAISP_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0 || param_stride <= 0) return ADAISP_EINVAL;
    if (B > 65535) return ADAISP_ESHAPE;
    return launch_backward_params(img, grad_out, filter_id, params, param_stride, grad_params, B, H, W, flags,
                                  static_cast<hipStream_t>(stream)) == hipSuccess
               ? ADAISP_OK
               : ADAISP_ELAUNCH;
}

int adaisp_policy_conv(const float* in, const float* states, int n_state, const float* w, const float* bias, float* out,
                       int G, int B, int Cin, int Hin, int Cout, void* stream) {
    if (!in || !w || !bias || !out || G <= 0 || B <= 0 || Cin <= 0 || Hin <= 1 || Cout <= 0) return ADAISP_EINVAL;
    if (Cout % 8 || Hin % 2 || G > 65535 || Cin > 128 || (states && (Cin < 3 || n_state != Cin - 3))) return ADAISP_ESHAPE;
    return launch_policy_conv(in, states, n_state, w, bias, out, G, B, Cin, Hin, Cout,
                              static_cast<hipStream_t>(stream)) == hipSuccess ? ADAISP_OK : ADAISP_ELAUNCH;
}

int adaisp_policy_fc1(const float* feats, const int32_t* head_src, const float* w1, const float* b1, float* hidden,
                      int B, int D, int NH, int HID, void* stream) {
    if (!feats || !head_src || !w1 || !b1 || !hidden || B <= 0 || D <= 0 || NH <= 0 || HID <= 0) return ADAISP_EINVAL;
    if (D % 1024 || HID % 4) return ADAISP_ESHAPE;     // 4 waves x 64 lanes x float4 per trip; 4 neurons per workgroup
    return launch_policy_fc1(feats, head_src, w1, b1, hidden, B, D, NH, HID, static_cast<hipStream_t>(stream)) ==
                   hipSuccess ? ADAISP_OK : ADAISP_ELAUNCH;
}

int adaisp_policy_finish(const adaisp_policy_finish_args* a, int B, void* stream) {
    if (!a || B <= 0) return ADAISP_EINVAL;
    if (!a->hidden || !a->w_filter || !a->b_filter || !a->row_filter || !a->row_slot || !a->w_sel || !a->b_sel ||
        !a->noise || !a->states || !a->params_all || !a->packed || !a->op_ids || !a->selected || !a->pdf_out ||
        !a->surrogate || !a->new_states || !a->penalty)
        return ADAISP_EINVAL;
    if (a->num_filters <= 0 || a->num_filters > ADAISP_POLICY_MAX_FILTERS || a->param_width <= 0 ||
        a->param_width > ADAISP_MAX_PARAMS || a->hid <= 0 || a->num_rows <= 0 || B > 65535)
        return ADAISP_ESHAPE;
    return launch_policy_finish(*a, B, static_cast<hipStream_t>(stream)) == hipSuccess ? ADAISP_OK : ADAISP_ELAUNCH;
}

}  // extern "C"
