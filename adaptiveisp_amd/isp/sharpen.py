"""Sharpening entry points — mirror of the reference's isp/sharpen.py, backed by the LDS-tiled HIP
stencil kernels (csrc/isp_conv.hip)."""
import torch

from .. import _lib
from .isp_function import isp_apply


def _per_image(v, B, device):
    v = torch.as_tensor(v, dtype=torch.float32, device=device).reshape(-1, 1)
    return v.expand(B, 1) if v.shape[0] == 1 else v


def adjust_sharpness(image, factor):
    """clamp(image*f + blur3x3(image)*(1-f)), 1-px frame keeps the image (reference isp/sharpen.py:105-142)."""
    return isp_apply(image, _per_image(factor, image.shape[0], image.device), _lib.OP_SHARPEN, clip=False)


def sharpness(image, factor):
    """clamp(image + (image - blur3x3(image))*f) (reference isp/sharpen.py:145-182)."""
    return isp_apply(image, _per_image(factor, image.shape[0], image.device), _lib.OP_SHARPEN_V2, clip=False)


def unsharp_mask(img, sigma, amount, kernel_size=(5, 5), clip=True):
    """img + (img - gaussian5x5(img; sigma))*amount, reflect border (reference isp/sharpen.py:84-102)."""
    if tuple(kernel_size) != (5, 5) or not clip:
        raise NotImplementedError("the HIP unsharp-mask kernel is built for kernel_size=(5,5), clip=True")
    B = img.shape[0]
    p = torch.cat([_per_image(sigma, B, img.device), _per_image(amount, B, img.device)], dim=1)
    return isp_apply(img, p, _lib.OP_USM, clip=False)
