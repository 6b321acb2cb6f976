"""ctypes binding of csrc/libadaisp.so — the C-ABI declared in include/adaisp.h.

PyTorch is plumbing here: it owns device memory and the HIP stream; every image operation below is a
kernel from the shared library. Nothing in this module falls back to eager PyTorch or to the CPU.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# ADAISP_LIB: another BUILD of the same library (tools/build_variant.py); never a fallback
LIB_PATH = os.environ.get("ADAISP_LIB") or os.path.join(_HERE, "csrc", "libadaisp.so")

OP_ZERO, OP_EXPOSURE, OP_GAMMA, OP_CCM, OP_SHARPEN, OP_NLM, OP_TONE = -1, 0, 1, 2, 3, 4, 5
OP_CONTRAST, OP_SATPLUS, OP_WNB, OP_WB, OP_USM, OP_SHARPEN_V2, OP_COLOR = 6, 7, 8, 9, 10, 11, 12
MAX_PARAMS = 24
CLIP01 = 1
NLM_EXACT = 2      # NLM patch sums in the reference's running-sum order (slower); default is the separable kernel
NLM_SEP_V1 = 4     # the compiler-scheduled form of the separable kernel (cross-check / measurement)
NO_USM = 8         # adaisp_forward: no image selects the unsharp mask (its empty launch is skipped)
NLM_TILE32 = 16    # the 32-row tile of the default NLM kernel (cross-check / measurement)
ABI_VERSION = 8

EXPORTS = ("adaisp_forward", "adaisp_forward_uniform", "adaisp_process", "adaisp_backward_params", "adaisp_pool64", "adaisp_pool64_backward", "adaisp_demosaic", "adaisp_nlm_general", "adaisp_nlm_general_workspace_bytes", "adaisp_num_params",
           "adaisp_policy_conv", "adaisp_policy_fc1", "adaisp_policy_finish",
           "adaisp_trunk_train_fwd", "adaisp_trunk_train_bwd", "adaisp_trunk_train_workspace_bytes", "adaisp_trunk_train_scratch_bytes",
           "adaisp_critic_planes_fwd", "adaisp_critic_planes_bwd", "adaisp_td_fwd", "adaisp_td_bwd",
           "adaisp_policy_tail_fwd", "adaisp_policy_tail_bwd", "adaisp_image_stats", "adaisp_clip_adam_step", "adaisp_clip_adam_step_dev", "adaisp_heads_fwd", "adaisp_heads_bwd",
           "adaisp_strerror", "adaisp_abi_version")

_lib = None


class AdaispError(RuntimeError):
    pass


def load():
    """Load libadaisp.so (once). Raises — never degrades — if the library is absent or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AdaispError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU/eager fallback for the ISP path.")
    L = ctypes.CDLL(LIB_PATH)
    vp, ci, cu = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint
    L.adaisp_forward.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, cu, vp]
    L.adaisp_forward_uniform.argtypes = [ci, vp, vp, vp, vp, ci, ci, ci, ci, cu, vp]
    L.adaisp_forward_uniform.restype = ci
    L.adaisp_process.argtypes = [ci, vp, vp, vp, ci, ci, ci, ci, cu, vp]
    L.adaisp_backward_params.argtypes = [vp, vp, vp, vp, ci, vp, ci, ci, ci, cu, vp]
    L.adaisp_pool64.argtypes = [vp, vp, ci, ci, ci, vp]
    L.adaisp_pool64_backward.argtypes = [vp, vp, ci, ci, ci, vp]
    L.adaisp_pool64_backward.restype = ci
    L.adaisp_demosaic.argtypes = [vp, vp, ci, ci, ci, ci, ctypes.c_float, ctypes.c_float, vp]
    L.adaisp_demosaic.restype = ci
    L.adaisp_nlm_general.argtypes = [vp, vp, vp, ci, vp, ctypes.c_size_t, ci, ci, ci, ci, ci, vp]
    L.adaisp_nlm_general.restype = ci
    L.adaisp_nlm_general_workspace_bytes.argtypes = [ci, ci, ci]
    L.adaisp_nlm_general_workspace_bytes.restype = ctypes.c_size_t
    for name in ("adaisp_trunk_train_fwd", "adaisp_trunk_train_bwd", "adaisp_critic_planes_fwd", "adaisp_critic_planes_bwd",
                 "adaisp_td_fwd", "adaisp_td_bwd", "adaisp_policy_tail_fwd", "adaisp_policy_tail_bwd", "adaisp_heads_fwd", "adaisp_heads_bwd"):
        getattr(L, name).argtypes = [vp, vp]
        getattr(L, name).restype = ci
    for name in ("adaisp_trunk_train_workspace_bytes", "adaisp_trunk_train_scratch_bytes"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = ctypes.c_size_t
    L.adaisp_image_stats.argtypes = [vp, vp, vp, ci, ctypes.c_long, vp]
    L.adaisp_clip_adam_step.argtypes = [vp, ci, ctypes.c_long, vp, ctypes.c_float] + [ctypes.c_double] * 4 + [vp]
    L.adaisp_clip_adam_step.restype = ci
    L.adaisp_clip_adam_step_dev.argtypes = [vp, ci, ctypes.c_long, vp, ctypes.c_float, vp] + [ctypes.c_double] * 3 + [vp]
    L.adaisp_clip_adam_step_dev.restype = ci
    L.adaisp_image_stats.restype = ci
    L.adaisp_num_params.argtypes = [ci]
    L.adaisp_strerror.argtypes = [ci]
    L.adaisp_strerror.restype = ctypes.c_char_p
    for name in ("adaisp_forward", "adaisp_process", "adaisp_backward_params", "adaisp_pool64", "adaisp_num_params",
                 "adaisp_abi_version"):
        getattr(L, name).restype = ci
    if L.adaisp_abi_version() != ABI_VERSION:
        raise AdaispError(f"libadaisp.so ABI {L.adaisp_abi_version()} != expected {ABI_VERSION}: rebuild")
    _lib = L
    return L


def _check(rc, what):
    if rc != 0:
        raise AdaispError(f"{what} failed: {load().adaisp_strerror(rc).decode()} ({rc})")


def _dev_f32(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise AdaispError(f"{name} is on {t.device}: the ISP kernels run on the HIP device only (no CPU fallback)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


def _wrote(t):
    """A kernel wrote `t` through its raw pointer: tell torch (version counter), so that anything keyed on the tensor's
    version — autograd's saved-tensor checks, Agent's cached pooling of its last output — sees the change."""
    if t is not None:
        torch.autograd.graph.increment_version(t)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _img_shape(img):
    if img.dim() != 4 or img.shape[1] != 3:
        raise ValueError(f"expected a [B,3,H,W] image, got {tuple(img.shape)}")
    return int(img.shape[0]), int(img.shape[2]), int(img.shape[3])


def image_stats(img):
    """[B,2] = (mean, number of non-finite values) per image of a [B,...] fp32 batch (adaisp_image_stats)."""
    L = load()
    img = _dev_f32(img.detach(), "img")
    B = int(img.shape[0])
    buf = torch.empty((B, 2 + 128), dtype=torch.float32, device=img.device)
    with torch.cuda.device(img.device):
        rc = L.adaisp_image_stats(img.data_ptr(), buf.data_ptr(), buf.data_ptr() + 8 * B, B, img.numel() // B, _stream())
    _check(rc, "adaisp_image_stats")
    return buf.view(-1)[:2 * B].view(B, 2)


def process(op, img, params, clip=False, out=None, nlm_exact=False, nlm_v1=False, nlm_tile32=False):
    """adaisp_process: one host-known op for the whole batch. params [B,n] (regressed)."""
    L = load()
    img = _dev_f32(img, "img")
    B, H, W = _img_shape(img)
    params = _dev_f32(params.reshape(B, -1), "params")
    if out is None:
        out = torch.empty_like(img)
    with torch.cuda.device(img.device):
        rc = L.adaisp_process(int(op), img.data_ptr(), out.data_ptr(), params.data_ptr(), params.shape[1], B, H, W,
                              (CLIP01 if clip else 0) | (NLM_EXACT if nlm_exact else 0) | (NLM_SEP_V1 if nlm_v1 else 0) |
                              (NLM_TILE32 if nlm_tile32 else 0),
                              _stream())
    _check(rc, "adaisp_process")
    _wrote(out)
    return out


def forward(img, op_ids, params, clip=True, pooled=None, out=None, nlm_exact=False, no_usm=False, host_op=None):
    """adaisp_forward: image b is filtered by op_ids[b] (int32, device). params [B,stride]. `pooled` ([B,3,64,64]) receives
    the 64x64 pooling of the result (fused into the filter launch where the geometry allows). `host_op`: the caller KNOWS
    that every op_ids[b] == host_op (a teacher-forced step) -> adaisp_forward_uniform, one launch instead of one per family."""
    L = load()
    img = _dev_f32(img, "img")
    B, H, W = _img_shape(img)
    params = _dev_f32(params.reshape(B, -1), "params")
    if host_op is None:
        if op_ids.dtype != torch.int32 or not op_ids.is_cuda:
            raise TypeError("op_ids must be an int32 device tensor")
        op_ids = op_ids.contiguous()
    if pooled is not None and (tuple(pooled.shape) != (B, 3, 64, 64) or pooled.dtype != torch.float32 or
                               pooled.device != img.device or not pooled.is_contiguous()):
        raise ValueError(f"pooled must be a contiguous float32 [{B},3,64,64] tensor on {img.device}")
    if out is None:
        out = torch.empty_like(img)
    elif (out.shape != img.shape or out.dtype != torch.float32 or out.device != img.device or not out.is_contiguous()):
        raise ValueError(f"out must be a contiguous float32 {tuple(img.shape)} tensor on {img.device}, got {out.dtype} "
                         f"{tuple(out.shape)} on {out.device}")
    flags = (CLIP01 if clip else 0) | (NLM_EXACT if nlm_exact else 0) | (NO_USM if no_usm else 0)
    with torch.cuda.device(img.device):
        if host_op is not None:
            rc = L.adaisp_forward_uniform(int(host_op), img.data_ptr(), out.data_ptr(),
                                          pooled.data_ptr() if pooled is not None else None, params.data_ptr(),
                                          params.shape[1], B, H, W, flags, _stream())
        else:
            rc = L.adaisp_forward(img.data_ptr(), out.data_ptr(), pooled.data_ptr() if pooled is not None else None,
                                  op_ids.data_ptr(), params.data_ptr(), params.shape[1], B, H, W, flags, _stream())
    _check(rc, "adaisp_forward")
    _wrote(out)
    _wrote(pooled)
    return out


def nlm_general(img, h, search_window_size, patch_size, out=None):
    """adaisp_nlm_general: NonLocalMeansGray(search_window_size, patch_size).forward(img, h) for any odd sizes (the ISP's
    11 / 5 goes through OP_NLM's tuned kernel). h: one value per image."""
    L = load()
    img = _dev_f32(img, "img")
    B, H, W = _img_shape(img)
    h = _dev_f32(h.reshape(B, -1), "h")
    if out is None:
        out = torch.empty_like(img)
    nbytes = int(L.adaisp_nlm_general_workspace_bytes(B, H, W))
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=img.device)
    with torch.cuda.device(img.device):
        rc = L.adaisp_nlm_general(img.data_ptr(), out.data_ptr(), h.data_ptr(), h.shape[1], ws.data_ptr(), nbytes, B, H, W,
                                  int(search_window_size), int(patch_size), _stream())
    _check(rc, "adaisp_nlm_general")
    _wrote(out)
    return out


def backward_params(img, grad_out, op_ids, params, clip=True):
    L = load()
    img = _dev_f32(img, "img")
    grad_out = _dev_f32(grad_out, "grad_out")
    B, H, W = _img_shape(img)
    params = _dev_f32(params.reshape(B, -1), "params")
    grad = torch.empty_like(params)
    with torch.cuda.device(img.device):
        rc = L.adaisp_backward_params(img.data_ptr(), grad_out.data_ptr(), op_ids.contiguous().data_ptr(),
                                      params.data_ptr(), params.shape[1], grad.data_ptr(), B, H, W,
                                      CLIP01 if clip else 0, _stream())
    _check(rc, "adaisp_backward_params")
    return grad


def pool64(img):
    """AdaptiveAvgPool2d((64,64)) of a [B,3,H,W] device image."""
    L = load()
    img = _dev_f32(img, "img")
    B, H, W = _img_shape(img)
    out = torch.empty((B, 3, 64, 64), dtype=torch.float32, device=img.device)
    with torch.cuda.device(img.device):
        rc = L.adaisp_pool64(img.data_ptr(), out.data_ptr(), B, H, W, _stream())
    _check(rc, "adaisp_pool64")
    return out


def pool64_backward(grad_pooled, H, W):
    """grad_pooled [B,3,64,64] -> gradient w.r.t. the [B,3,H,W] image that adaisp_pool64 pooled."""
    L = load()
    g = _dev_f32(grad_pooled, "grad_pooled")
    B = g.shape[0]
    if tuple(g.shape[1:]) != (3, 64, 64):
        raise AdaispError(f"grad_pooled must be [B,3,64,64], got {tuple(g.shape)}")
    out = torch.empty((B, 3, H, W), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        rc = L.adaisp_pool64_backward(g.data_ptr(), out.data_ptr(), B, H, W, _stream())
    _check(rc, "adaisp_pool64_backward")
    return out


CFA = {"RGGB": 0, "GRBG": 1, "GBRG": 2, "BGGR": 3}


def demosaic(raw, pattern="RGGB", black_level=0.0, white_level=65535.0, out=None):
    """Bayer front-end: raw uint16 [B,H,W] on the device -> planar fp32 [B,3,H,W] in [0,1] (include/adaisp.h)."""
    L = load()
    if raw.device.type != "cuda":
        raise AdaispError("demosaic: raw must live on a HIP device (there is no CPU path)")
    if raw.dtype not in (torch.uint16, torch.int16) or raw.dim() != 3:
        raise AdaispError(f"demosaic: raw must be uint16 [B,H,W], got {raw.dtype} {tuple(raw.shape)}")
    raw = raw.contiguous()
    B, H, W = raw.shape
    if out is None:
        out = torch.empty((B, 3, H, W), dtype=torch.float32, device=raw.device)
    pat = CFA[pattern.upper()] if isinstance(pattern, str) else int(pattern)
    with torch.cuda.device(raw.device):
        rc = L.adaisp_demosaic(raw.data_ptr(), out.data_ptr(), B, H, W, pat, float(black_level), float(white_level),
                               _stream())
    _check(rc, "adaisp_demosaic")
    _wrote(out)
    return out


def pack_rggb_to_plane(packed):
    """The reference's 4-channel Bayer packing [..., H/2, W/2, 4] = (R, Gr, Gb, B) (`mosaic`, isp/unprocess_np.py:82-98)
    -> the flat colour-filter-array plane [..., H, W] (`reconstruct_bayer` :111-128 for 'rggb')."""
    h2, w2 = packed.shape[-3], packed.shape[-2]
    plane = packed.new_empty(packed.shape[:-3] + (2 * h2, 2 * w2))
    plane[..., 0::2, 0::2] = packed[..., 0]
    plane[..., 0::2, 1::2] = packed[..., 1]
    plane[..., 1::2, 0::2] = packed[..., 2]
    plane[..., 1::2, 1::2] = packed[..., 3]
    return plane
