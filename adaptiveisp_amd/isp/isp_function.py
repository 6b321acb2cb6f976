"""Autograd bridge between PyTorch and the C-ABI ISP kernels.

forward  : adaisp_process (one host-known op) or adaisp_forward (per-image op ids on the device)
backward : adaisp_backward_params — gradient w.r.t. the regressed filter parameters only. The image
           gradient is not produced (reference training treats images as constants, train.py:255-258,
           341-342); asking for it raises instead of silently returning zeros.
"""
import torch

from .. import _lib


def _flat_params(img, param):
    B = img.shape[0]
    p = param.reshape(param.shape[0], -1)
    if p.shape[0] == 1 and B > 1:
        p = p.expand(B, -1)
    if p.shape[0] != B:
        raise ValueError(f"param batch {p.shape[0]} does not match image batch {B}")
    return p.to(torch.float32).contiguous()


class _IspFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, params, op_ids, uniform_op, clip):
        if op_ids is None:
            out = _lib.process(uniform_op, img, params, clip=clip)
        else:
            out = _lib.forward(img, op_ids, params, clip=clip)
        ctx.save_for_backward(img, params, op_ids)
        ctx.uniform_op, ctx.clip = uniform_op, clip
        return out

    @staticmethod
    def backward(ctx, grad_out):
        img, params, op_ids = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("d(out)/d(img) of the ISP kernels is not built: the reference's training "
                                      "only differentiates w.r.t. the filter parameters (train.py:341-342)")
        if op_ids is None:
            op_ids = torch.full((img.shape[0],), ctx.uniform_op, dtype=torch.int32, device=img.device)
        grad_p = _lib.backward_params(img, grad_out, op_ids, params, clip=ctx.clip)
        return None, grad_p, None, None, None


def isp_apply(img, param, op, clip):
    """One host-known op for the whole batch (Filter.process / Filter.forward)."""
    p = _flat_params(img, param)
    n = _lib.load().adaisp_num_params(int(op))
    if n < 0 or p.shape[1] < n:
        raise ValueError(f"op {op} needs {n} parameters per image, got {p.shape[1]}")
    if torch.is_grad_enabled() and (p.requires_grad or img.requires_grad):
        return _IspFunction.apply(img, p, None, int(op), bool(clip))
    return _lib.process(int(op), img, p, clip=clip)


def isp_apply_selected(img, packed_params, op_ids, clip=True):
    """Per-image ops chosen on the device (Agent.forward): op_ids int32 [B], packed_params [B,stride]."""
    if torch.is_grad_enabled() and (packed_params.requires_grad or img.requires_grad):
        return _IspFunction.apply(img, packed_params.contiguous(), op_ids, 0, bool(clip))
    return _lib.forward(img, op_ids, packed_params, clip=clip)
