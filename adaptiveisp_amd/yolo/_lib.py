"""ctypes binding of csrc/libadayolo.so (C-ABI in include/adayolo.h). No eager/CPU fallback."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# ADAYOLO_LIB: another BUILD of the same library (measurement builds of tools/build_variant.py); never a fallback
LIB_PATH = os.environ.get("ADAYOLO_LIB") or os.path.join(_HERE, "csrc", "libadayolo.so")
ABI_VERSION = 9
ACT_NONE, ACT_SILU = 0, 1
EXPORTS = ("adayolo_conv_fwd", "adayolo_conv_fwd_variant", "adayolo_conv_fused1x1_fwd", "adayolo_conv1x1_stream_fwd", "adayolo_bottleneck256_fwd", "adayolo_bottleneck_ws_fwd", "adayolo_conv_keep_fwd", "adayolo_conv_splitk_fwd", "adayolo_conv_dsilu_fwd", "adayolo_conv_s2grad_fwd", "adayolo_conv_splitk_workspace_bytes", "adayolo_conv_chain_workspace_bytes", "adayolo_conv_chain_prepare", "adayolo_conv_chain_fwd", "adayolo_conv_chain_status", "adayolo_conv_chain_poll", "adayolo_conv_chain_tables", "adayolo_stem_fwd", "adayolo_upsample2x", "adayolo_detect_decode", "adayolo_nms", "adayolo_nms_workspace_bytes", "adayolo_stem_fwd_act", "adayolo_stem_keep_fwd", "adayolo_stem_down_fwd", "adayolo_letterbox_pack", "adayolo_silu_fwd", "adayolo_silu_bwd",
           "adayolo_zero_insert2x", "adayolo_upsample2x_bwd", "adayolo_image_grad", "adayolo_detloss_fwd", "adayolo_detloss_bwd",
           "adayolo_strerror",
           "adayolo_abi_version")
_lib = None


class AdayoloError(RuntimeError):
    pass


class LossLayer(ctypes.Structure):                 # adayolo_loss_layer (include/adayolo.h)
    _fields_ = [("raw", ctypes.c_void_p), ("cs", ctypes.c_int), ("ny", ctypes.c_int), ("nx", ctypes.c_int),
                ("balance", ctypes.c_float), ("idx", ctypes.c_void_p), ("box", ctypes.c_void_p), ("n", ctypes.c_int),
                ("part", ctypes.c_void_p), ("tobj", ctypes.c_void_p), ("cnt", ctypes.c_void_p),
                ("grad", ctypes.c_void_p), ("grad_cs", ctypes.c_int)]


class ChainLayer(ctypes.Structure):                # adayolo_chain_layer
    _fields_ = [("in_", ctypes.c_void_p), ("in_cstride", ctypes.c_int), ("weight", ctypes.c_void_p), ("bias", ctypes.c_void_p),
                ("residual", ctypes.c_void_p), ("res_cstride", ctypes.c_int), ("out", ctypes.c_void_p), ("out_cstride", ctypes.c_int),
                ("B", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("Cin", ctypes.c_int), ("Cout", ctypes.c_int),
                ("ksize", ctypes.c_int), ("stride", ctypes.c_int), ("act", ctypes.c_int), ("weight2", ctypes.c_void_p),
                ("bias2", ctypes.c_void_p), ("out2", ctypes.c_void_p), ("out2_cstride", ctypes.c_int), ("Cout2", ctypes.c_int),
                ("tile", ctypes.c_int)]


class LossArgs(ctypes.Structure):                  # adayolo_loss_args
    _fields_ = [("layer", LossLayer * 4), ("nl", ctypes.c_int), ("B", ctypes.c_int), ("na", ctypes.c_int),
                ("nc", ctypes.c_int), ("no", ctypes.c_int), ("hyp_box", ctypes.c_float), ("hyp_obj", ctypes.c_float),
                ("hyp_cls", ctypes.c_float), ("cp", ctypes.c_float), ("cn", ctypes.c_float), ("cls_pw", ctypes.c_float),
                ("obj_pw", ctypes.c_float), ("loss", ctypes.c_void_p), ("ticket", ctypes.c_void_p),
                ("grad_loss", ctypes.c_void_p)]


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AdayoloError(f"{LIB_PATH} not found: run __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                           "the detector has no eager fallback")
    L = ctypes.CDLL(LIB_PATH)
    vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    L.adayolo_conv_fwd.argtypes = [vp, ci, vp, vp, vp, ci, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, vp]
    L.adayolo_conv_fwd_variant.argtypes = [vp, ci, vp, vp, vp, ci, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, vp]
    L.adayolo_conv_fwd_variant.restype = ci
    L.adayolo_conv_fused1x1_fwd.argtypes = [vp, ci, vp, vp, vp, ci, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, vp, vp, vp, ci, ci, vp]
    L.adayolo_conv_fused1x1_fwd.restype = ci
    L.adayolo_bottleneck256_fwd.argtypes = [vp, ci, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp]
    L.adayolo_bottleneck256_fwd.restype = ci
    L.adayolo_bottleneck_ws_fwd.argtypes = [vp, ci, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]
    L.adayolo_bottleneck_ws_fwd.restype = ci
    L.adayolo_conv_keep_fwd.argtypes = [vp, ci, vp, vp, vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, vp]
    L.adayolo_conv_keep_fwd.restype = ci
    L.adayolo_stem_fwd.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, cf, ci, vp]
    L.adayolo_upsample2x.argtypes = [vp, ci, vp, ci, ci, ci, ci, ci, vp]
    L.adayolo_detect_decode.argtypes = [vp, ci, vp, ci, ci, vp, cf, ci, ci, ci, ci, ci, vp]
    cl = ctypes.c_long
    L.adayolo_stem_fwd_act.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, cf, ci, ci, vp]
    L.adayolo_stem_keep_fwd.argtypes = [vp, vp, vp, vp, ci, vp, ci, ci, ci, ci, ci, ci, cf, ci, vp]
    L.adayolo_stem_down_fwd.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, cf, vp, vp, vp, ci, vp]
    L.adayolo_stem_down_fwd.restype = ci
    L.adayolo_letterbox_pack.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, cf, vp]
    L.adayolo_letterbox_pack.restype = ci
    L.adayolo_silu_fwd.argtypes = [vp, ci, vp, ci, vp, ci, cl, ci, vp]
    L.adayolo_silu_bwd.argtypes = [vp, ci, vp, ci, vp, ci, vp, ci, ci, cl, ci, vp]
    L.adayolo_zero_insert2x.argtypes = [vp, ci, vp, ci, ci, ci, ci, ci, ci, ci, vp]
    L.adayolo_upsample2x_bwd.argtypes = [vp, ci, vp, ci, ci, ci, ci, ci, ci, vp]
    L.adayolo_image_grad.argtypes = [vp, ci, vp, ci, ci, ci, ci, ci, vp]
    for n in ("adayolo_stem_fwd_act", "adayolo_stem_keep_fwd", "adayolo_stem_down_fwd", "adayolo_letterbox_pack", "adayolo_silu_fwd", "adayolo_silu_bwd", "adayolo_zero_insert2x",
              "adayolo_upsample2x_bwd", "adayolo_image_grad"):
        getattr(L, n).restype = ci
    for n in ("adayolo_detloss_fwd", "adayolo_detloss_bwd"):
        getattr(L, n).argtypes = [ctypes.POINTER(LossArgs), vp]
        getattr(L, n).restype = ci
    L.adayolo_nms.argtypes = [vp, ci, cf, ci, vp, vp, vp, vp]
    L.adayolo_nms.restype = ci
    L.adayolo_conv_splitk_workspace_bytes.argtypes = [ci] * 8
    L.adayolo_conv_splitk_workspace_bytes.restype = ctypes.c_size_t
    L.adayolo_conv_splitk_fwd.argtypes = [vp, ci, vp, vp, vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, vp,
                                          ctypes.c_size_t, vp]
    L.adayolo_conv_splitk_fwd.restype = ci
    L.adayolo_conv_dsilu_fwd.argtypes = [vp, ci, vp, vp, vp, ci, vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, vp,
                                         ctypes.c_size_t, vp]
    L.adayolo_conv_dsilu_fwd.restype = ci
    L.adayolo_conv_s2grad_fwd.argtypes = [vp, ci, vp, vp, vp, ci, vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, ci, ci, vp, ctypes.c_size_t, vp]
    L.adayolo_conv_s2grad_fwd.restype = ci
    L.adayolo_conv_chain_workspace_bytes.argtypes = [ctypes.POINTER(ChainLayer), ci]
    L.adayolo_conv_chain_workspace_bytes.restype = ctypes.c_size_t
    L.adayolo_conv_chain_prepare.argtypes = [ctypes.POINTER(ChainLayer), ci, vp, ctypes.c_size_t]
    L.adayolo_conv_chain_prepare.restype = ci
    L.adayolo_conv_chain_fwd.argtypes = [ctypes.POINTER(ChainLayer), ci, vp, ctypes.c_size_t, vp]
    L.adayolo_conv_chain_fwd.restype = ci
    L.adayolo_conv_chain_tables.argtypes = [ctypes.POINTER(ChainLayer), ci, vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int32)]
    L.adayolo_conv_chain_tables.restype = ci
    L.adayolo_conv_chain_status.argtypes = [vp]
    L.adayolo_conv_chain_status.restype = ci
    L.adayolo_conv1x1_stream_fwd.argtypes = [vp, ci, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp]
    L.adayolo_conv1x1_stream_fwd.restype = ci
    L.adayolo_conv_chain_poll.argtypes = [vp]
    L.adayolo_conv_chain_poll.restype = ci
    L.adayolo_nms_workspace_bytes.argtypes = [ci]
    L.adayolo_nms_workspace_bytes.restype = ctypes.c_size_t
    L.adayolo_strerror.argtypes = [ci]
    L.adayolo_strerror.restype = ctypes.c_char_p
    for n in ("adayolo_conv_fwd", "adayolo_stem_fwd", "adayolo_upsample2x", "adayolo_detect_decode",
              "adayolo_abi_version"):
        getattr(L, n).restype = ci
    if L.adayolo_abi_version() != ABI_VERSION:
        raise AdayoloError("libadayolo.so ABI mismatch: rebuild")
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        raise AdayoloError(f"{what} failed: {load().adayolo_strerror(rc).decode()} ({rc})")


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
