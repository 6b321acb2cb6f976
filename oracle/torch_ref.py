"""torch-CPU restatement of the reference's ISP op chain — TEST INFRASTRUCTURE ONLY (the `cpu_baseline` leg of bench.py
and tests/ import it; the product never does).

Where oracle/isp_oracle.c restates each filter per pixel (gather form, one pass), this file keeps the reference's
FORMULATION: whole-tensor ops in the reference's order — roll-based non-local means (121 shifted copies, 25-term box
sums by rolling), the tone curve as 8 clamp passes, HSV by sequential masked overwrite, and the policy step that runs
ALL filters on the batch and keeps one per image with a one-hot multiply-sum. That formulation is what the reference
costs on a CPU, so it is what `cpu_baseline` times as "reference-faithful"; `selected_only=True` is the same chain
computing only the filter each image selected. Pinned: tests/test_oracle_golden.py checks every function here against
tests/golden/filters.npz / nlm.npz (outputs of the reference itself).

Citations are to the reference checkout (isp/filters.py, isp/denoise.py, isp/sharpen.py, agent.py).
"""
import math

import torch
import torch.nn.functional as F

# default filter order of config.py:19-22 and the op codes of include/adaisp.h
E, G, CCM, SHR, NLM, T, CT, SP, BW, W, USM, SHRV2, COLOR = range(13)


def _lum(img):                                             # filters.py:32-34 rgb2lum (keeps dim)
    return 0.27 * img[:, 0:1] + 0.67 * img[:, 1:2] + 0.06 * img[:, 2:3]


def _p(param, n=1):
    return param.reshape(param.shape[0], n, 1, 1) if n == 1 else param


def exposure(img, p):                                      # filters.py:215-224
    return img * torch.exp(_p(p) * math.log(2))


def gamma(img, p):                                         # filters.py:235-245
    return torch.pow(torch.clamp(img, min=0.001), _p(p))


def white_balance(img, p):                                 # filters.py:253-272
    return img * p[:, :3, None, None]


def ccm(img, p):                                           # filters.py:666-672,694-708: rows normalised, no epsilon
    m = p.reshape(-1, 3, 3)
    m = m / m.sum(dim=2, keepdim=True)
    x = img.permute(0, 2, 3, 1)[..., None, :]              # [B,H,W,1,3]
    return (x * m[:, None, None]).sum(dim=-1).permute(0, 3, 1, 2)


def tone(img, p):                                          # filters.py:326-347: eight clamp passes, this order
    p = p.reshape(p.shape[0], 8, 1, 1, 1)
    total = p.sum(dim=1) + 1e-30
    acc = torch.zeros_like(img)
    for i in range(8):
        acc = acc + torch.clamp(img - 1.0 * i / 8, 0, 1.0 / 8) * p[:, i]
    return acc * (8 / total)


def color_curve(img, p):                                   # filters.py:281-303: per-channel curves
    p = p.reshape(p.shape[0], 8, 3, 1, 1)
    total = p.sum(dim=1) + 1e-30
    acc = torch.zeros_like(img)
    for i in range(8):
        acc = acc + torch.clamp(img - 1.0 * i / 8, 0, 1.0 / 8) * p[:, i]
    return acc * (8 / total)


def contrast(img, p):                                      # filters.py:406-419
    lum = torch.clamp(_lum(img), 0.0, 1.0)
    cl = -torch.cos(math.pi * lum) * 0.5 + 0.5
    ci = img / (lum + 1e-6) * cl
    a = _p(p)
    return (1 - a) * img + a * ci


def wnb(img, p):                                           # filters.py:427-437
    a = _p(p)
    return (1 - a) * img + a * _lum(img)


def _rgb2hsv(x):                                           # filters.py:445-478: sequential masked overwrite
    r, g, b = x[:, 0], x[:, 1], x[:, 2]
    mx, mn = x.max(dim=1)[0], x.min(dim=1)[0]
    d = mx - mn + 1e-8
    h = torch.zeros_like(mx)
    m = b == mx
    h[m] = 4.0 + ((r - g) / d)[m]
    m = g == mx
    h[m] = 2.0 + ((b - r) / d)[m]
    m = r == mx
    h[m] = torch.remainder((g - b) / d, 6.0)[m]
    h[mn == mx] = 0.0
    h = h / 6.0
    s = (mx - mn) / (mx + 1e-8)
    s[mx == 0] = 0.0
    return h, s, mx


def _hsv2rgb(h, s, v):                                     # filters.py:481-533
    h = torch.remainder(h, 1.0)
    s, v = torch.clamp(s, 0, 1), torch.clamp(v, 0, 1)
    hi = torch.floor(h * 6)
    f = h * 6 - hi
    p, q, t = v * (1 - s), v * (1 - f * s), v * (1 - (1 - f) * s)
    out = torch.zeros((h.shape[0], 3) + tuple(h.shape[1:]), dtype=h.dtype)
    table = ((v, t, p), (q, v, p), (p, v, t), (p, q, v), (t, p, v), (v, p, q))
    for k, (rr, gg, bb) in enumerate(table):
        m = hi == k
        out[:, 0][m], out[:, 1][m], out[:, 2][m] = rr[m], gg[m], bb[m]
    return out


def saturation_plus(img, p):                               # filters.py:536-560
    x = torch.clamp(img, 0.0, 1.0)
    h, s, v = _rgb2hsv(x)
    s2 = s + (1 - s) * (0.5 - torch.abs(0.5 - v)) * 0.8
    full = _hsv2rgb(h, s2, v)
    a = _p(p)
    return x * (1 - a) + full * a


def _blur3(img):                                           # sharpen.py:105-142: K = ones; K[1,1] = 5; K /= 13; 1-px frame kept
    k = torch.ones(3, 3)
    k[1, 1] = 5.0
    k = (k / k.sum()).expand(3, 1, 3, 3)
    inner = F.conv2d(img, k, groups=3)
    out = img.clone()
    out[:, :, 1:-1, 1:-1] = inner
    return out


def sharpen(img, p):                                       # adjust_sharpness: img*f + blur*(1-f), clamped (sharpen.py:105-142)
    f = _p(p)
    return torch.clamp(img * f + _blur3(img) * (1 - f), 0, 1)


def sharpen_v2(img, p):                                    # sharpness: img + (img - blur)*f, clamped (sharpen.py:145-182)
    return torch.clamp(img + (img - _blur3(img)) * _p(p), 0, 1)


def usm(img, p):                                           # sharpen.py:15-31,63-102: per-image 5x5 Gaussian, reflect pad 2
    outs = []
    for b in range(img.shape[0]):
        sigma, amount = p[b, 0], p[b, 1]
        g = torch.exp(-0.5 * (torch.linspace(-2.0, 2.0, 5) / sigma) ** 2)
        g = g / g.sum()
        k = (g[:, None] * g[None, :]).expand(3, 1, 5, 5)
        x = img[b:b + 1]
        blur = F.conv2d(F.pad(x, (2, 2, 2, 2), mode="reflect"), k, groups=3)
        outs.append(torch.clamp(x + (x - blur) * amount, 0, 1))
    return torch.cat(outs, 0)


def nlm(img, p, search=11, patch=5):                       # denoise.py:93-119 (+ BoxFilter :46-65, luminance :11-17)
    h = torch.clamp(p.reshape(-1, 1, 1, 1), min=0) + 1e-8
    x = torch.clamp(img, 0.0, 1.0)
    y = 0.299 * x[:, 0:1] + 0.587 * x[:, 1:2] + 0.114 * x[:, 2:3]
    r, pr = search // 2, patch // 2
    num, den = torch.zeros_like(x), torch.zeros_like(y)
    for dx in range(-r, r + 1):
        for dy in range(-r, r + 1):
            xs = torch.roll(x, shifts=(dy, dx), dims=(2, 3))
            ys = torch.roll(y, shifts=(dy, dx), dims=(2, 3))
            sq = (y - ys) ** 2
            dist = torch.zeros_like(sq)
            for bx in range(-pr, pr + 1):                   # 25-term box sum by rolling, sequential adds from 0
                for by in range(-pr, pr + 1):
                    dist = dist + torch.roll(sq, shifts=(by, bx), dims=(2, 3))
            w = torch.exp(-torch.sqrt(torch.clamp(dist, min=0)) / h)
            num = num + xs * w
            den = den + w
    return torch.clamp(num / den, 0.0, 1.0)


PROCESS = {E: exposure, G: gamma, CCM: ccm, SHR: sharpen, NLM: nlm, T: tone, CT: contrast, SP: saturation_plus, BW: wnb,
           W: white_balance, USM: usm, SHRV2: sharpen_v2, COLOR: color_curve}
NUM_PARAMS = {E: 1, G: 1, CCM: 9, SHR: 1, NLM: 1, T: 8, CT: 1, SP: 1, BW: 1, W: 3, USM: 2, SHRV2: 1, COLOR: 24}


def process(op, img, param):
    """Filter.process of op (no clip)."""
    return PROCESS[int(op)](img, param.reshape(img.shape[0], -1).float())


def forward(op, img, param):
    """Filter.forward with masking off: clip((1-m)*img + m*process, 0, 1), m = 1 (filters.py:91-126,171-173)."""
    out = process(op, img, param)
    m = torch.ones(1, 1, 1, 1)
    return torch.clip((1 - m) * img + m * out, 0.0, 1.0)


def policy_step(img, params_by_filter, selected, filter_ops=tuple(range(10)), selected_only=False):
    """The pixel side of Agent.forward (agent.py:103-116,154). Reference-faithful: EVERY filter of `filter_ops` runs on
    the whole batch, the results are stacked and one per image survives a one-hot multiply-sum. `selected_only`: only
    the filters some image selected run, each on its own images. `selected` int64 [B] (-1 -> zero image)."""
    B = img.shape[0]
    if selected_only:
        out = torch.zeros_like(img)
        for j, op in enumerate(filter_ops):
            idx = (selected == j).nonzero().flatten()
            if idx.numel():
                out[idx] = forward(op, img[idx], params_by_filter[j][idx])
        return out
    outs = torch.stack([forward(op, img, params_by_filter[j]) for j, op in enumerate(filter_ops)], dim=1)   # [B,F,3,H,W]
    hot = (selected[:, None] == torch.arange(len(filter_ops))[None, :]).to(outs.dtype)
    return (outs * hot[:, :, None, None, None]).sum(dim=1)
