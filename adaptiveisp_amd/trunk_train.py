"""Training-mode CNN trunk of the policy / critic networks on the HIP kernels (csrc/isp_trunk_train.hip).

`FeatureExtractor.layers` in train mode — [Conv2d(k4 s2 p1) -> BatchNorm2d(batch statistics) -> LeakyReLU(0.2)] x 4,
reference agent.py:26-60 / value.py:6-44 — is ~13 ATen ops forward and ~20 autograd nodes backward per trunk, each a few
vendor-library launches on sub-megabyte tensors; the RL iteration (train.py:258,282-283) runs four trunk passes and is
bound by the host's enqueue work. `trunk_features` runs up to two trunk instances through ONE autograd node: one C call
forward (8 launches), one backward (11-15), same parameters / buffers / state-dict keys as the module path (the modules stay
the owners of every tensor; running statistics and `num_batches_tracked` are updated as nn.BatchNorm2d does).

Results agree with the module path to fp32 rounding (other summation orders; every sum here runs in a fixed order, so two
runs are bit-identical): tests/test_gpu_trunk_train.py.
"""
import ctypes
import os

import torch
import torch.nn as nn

from . import _lib

LAYERS = 4
MAX_G = 2


class _Params(ctypes.Structure):
    _fields_ = [(k, ctypes.c_void_p * LAYERS) for k in ("w", "bias", "gamma", "beta", "running_mean", "running_var")]


class _Grads(ctypes.Structure):
    _fields_ = [(k, ctypes.c_void_p * LAYERS) for k in ("w", "bias", "gamma", "beta")]


class _Args(ctypes.Structure):
    _fields_ = [("G", ctypes.c_int32), ("B", ctypes.c_int32), ("n_state", ctypes.c_int32), ("share_params", ctypes.c_int32),
                ("C", ctypes.c_int32 * (LAYERS + 1)),
                ("momentum", ctypes.c_float), ("eps", ctypes.c_float), ("slope", ctypes.c_float),
                ("img", ctypes.c_void_p * MAX_G), ("svec", ctypes.c_void_p * MAX_G), ("p", _Params * MAX_G),
                ("feat", ctypes.c_void_p), ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t),
                ("dfeat", ctypes.c_void_p), ("g", _Grads * MAX_G),
                ("dimg", ctypes.c_void_p * MAX_G), ("dsvec", ctypes.c_void_p * MAX_G),
                ("scratch", ctypes.c_void_p), ("scratch_bytes", ctypes.c_size_t)]


class _PlanesArgs(ctypes.Structure):
    _fields_ = [("G", ctypes.c_int32), ("B", ctypes.c_int32), ("n_state", ctypes.c_int32)] + \
               [(k, ctypes.c_void_p * MAX_G) for k in ("small", "states", "svec", "dsvec", "dsmall_in", "dsmall")]


def enabled():
    return os.environ.get("ADAISP_TRUNK_KERNELS", "1") == "1"


def _stages(trunk):
    """[(conv, bn)] x 4 of a FeatureExtractor the kernels serve, or None (another depth / input size, SyncBatchNorm, a
    BatchNorm without affine parameters / running statistics / momentum: the module path runs those)."""
    mods = list(trunk.layers)
    if len(mods) != 3 * LAYERS:
        return None
    out = []
    for i in range(LAYERS):
        conv, bn, act = mods[3 * i:3 * i + 3]
        if type(conv) is not nn.Conv2d or type(bn) is not nn.BatchNorm2d or type(act) is not nn.LeakyReLU:
            return None
        if (conv.kernel_size, conv.stride, conv.padding, conv.dilation, conv.groups) != ((4, 4), (2, 2), (1, 1), (1, 1), 1):
            return None
        if conv.bias is None or not bn.affine or not bn.track_running_stats or bn.momentum is None:
            return None
        if conv.out_channels % 16 or (i and conv.in_channels != mods[3 * i - 3].out_channels):
            return None
        if act.negative_slope != mods[2].negative_slope or bn.eps != mods[1].eps or bn.momentum != mods[1].momentum:
            return None
        out.append((conv, bn))
    return out


def serves(trunk, img, svec, extra=0):
    """Whether `trunk_features` can run this trunk on these inputs (train mode, HIP device, fp32, 64x64); `extra`: constant
    planes the node appends itself (the critic's three statistics)."""
    if not (enabled() and trunk.training and img.is_cuda and img.dtype == torch.float32 and tuple(img.shape[1:]) == (3, 64, 64)):
        return False
    key = tuple(map(id, trunk.layers))                  # (a converted / edited nn.Sequential is looked at again)
    if getattr(trunk, "_trunk_key", None) != key:
        trunk._trunk_key, trunk._trunk_stages = key, _stages(trunk)
    st = trunk._trunk_stages
    if st is None:
        return False
    n_state = 0 if svec is None else int(svec.shape[1])
    return st[0][0].in_channels == 3 + n_state + extra


def _fill(args, trunks, imgs, svecs, share):
    G, B = len(trunks), int(imgs[0].shape[0])
    st0 = trunks[0]._trunk_stages
    args.G, args.B, args.share_params = G, B, 1 if share else 0
    args.n_state = 0 if svecs[0] is None else int(svecs[0].shape[1])
    args.C[0] = st0[0][0].in_channels
    for l in range(LAYERS):
        args.C[l + 1] = st0[l][0].out_channels
    args.momentum, args.eps, args.slope = float(st0[0][1].momentum), float(st0[0][1].eps), float(trunks[0].layers[2].negative_slope)
    for g in range(G):
        args.img[g] = imgs[g].data_ptr()
        args.svec[g] = None if svecs[g] is None else svecs[g].data_ptr()
        p = args.p[g]
        for l, (conv, bn) in enumerate(trunks[g]._trunk_stages):
            p.w[l], p.bias[l] = conv.weight.data_ptr(), conv.bias.data_ptr()
            p.gamma[l], p.beta[l] = bn.weight.data_ptr(), bn.bias.data_ptr()
            p.running_mean[l], p.running_var[l] = bn.running_mean.data_ptr(), bn.running_var.data_ptr()


def _param_list(trunks, share):
    out = []
    for t in (trunks[:1] if share else trunks):
        for conv, bn in t._trunk_stages:
            out += [conv.weight, conv.bias, bn.weight, bn.bias]
    return out


class _TrunkFn(torch.autograd.Function):
    """forward(trunks, share, n_in, planes, img..., svec..., params...). With `planes` (the critic) the svec inputs are the
    raw state vectors and the three hand statistics of value.py:65-80 are computed from the image planes and appended by a
    kernel of this node (adaisp_critic_planes_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, trunks, share, n_in, planes, *tensors):
        L = _lib.load()
        imgs = [t.contiguous() for t in tensors[:n_in]]
        svecs = [None if t is None else t.contiguous() for t in tensors[n_in:2 * n_in]]
        G = len(trunks)
        dev = imgs[0].device
        B = int(imgs[0].shape[0])
        pargs = None
        with torch.cuda.device(dev):
            if planes:
                S = 0 if svecs[0] is None else int(svecs[0].shape[1])
                pargs = _PlanesArgs()
                pargs.G, pargs.B, pargs.n_state = n_in, B, S
                ext = [torch.empty((B, S + 3), dtype=torch.float32, device=dev) for _ in range(n_in)]
                for g in range(n_in):
                    pargs.small[g], pargs.svec[g] = imgs[g].data_ptr(), ext[g].data_ptr()
                    pargs.states[g] = None if svecs[g] is None else svecs[g].data_ptr()
                _lib._check(L.adaisp_critic_planes_fwd(ctypes.byref(pargs), _lib._stream()), "adaisp_critic_planes_fwd")
                raw_states, svecs = svecs, ext
            else:
                raw_states = None
            if n_in == 1:                                   # one input for every instance (the agent's two trunks)
                imgs, svecs = imgs * G, svecs * G
            args = _Args()
            _fill(args, trunks, imgs, svecs, share)
            ref = ctypes.byref(args)
            D = args.C[LAYERS] * 16
            feat = torch.empty((G, args.B, D), dtype=torch.float32, device=dev)
            ws = torch.empty((L.adaisp_trunk_train_workspace_bytes(ref) // 4,), dtype=torch.float32, device=dev)
            args.feat, args.workspace, args.workspace_bytes = feat.data_ptr(), ws.data_ptr(), ws.numel() * 4
            rc = L.adaisp_trunk_train_fwd(ref, _lib._stream())
        _lib._check(rc, "adaisp_trunk_train_fwd")
        bns = [bn for t in (trunks[:1] if share else trunks) for _, bn in t._trunk_stages]
        for bn in bns:                                      # the kernels wrote the running statistics through raw pointers
            _lib._wrote(bn.running_mean)
            _lib._wrote(bn.running_var)
        torch._foreach_add_([bn.num_batches_tracked for bn in bns], G if share else 1)
        ctx.args, ctx.pargs, ctx.keep, ctx.n_in, ctx.share, ctx.trunks = args, pargs, (imgs, svecs, raw_states, feat, ws), n_in, share, trunks
        ctx.save_for_backward(*tensors[2 * n_in:])          # the parameters: autograd's version check guards in-place edits
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        L = _lib.load()
        if ctx.keep is None:
            raise RuntimeError("trunk_features: a second backward through the same node (retain_graph) is not supported: the "
                               "saved activations are released by the first")
        args, pargs, (imgs, svecs, raw_states, feat, ws), n_in, share, trunks = ctx.args, ctx.pargs, ctx.keep, ctx.n_in, ctx.share, ctx.trunks
        params = ctx.saved_tensors
        dfeat = dfeat.contiguous()
        dev = dfeat.device
        G = len(trunks)
        need_img = [bool(ctx.needs_input_grad[4 + i]) for i in range(n_in)]
        need_sv = [bool(ctx.needs_input_grad[4 + n_in + i]) for i in range(n_in)]
        need_in = [a or b for a, b in zip(need_img, need_sv)]
        if n_in == 1 and G > 1 and need_in[0]:
            raise NotImplementedError("input gradient of a shared trunk input (the agent's pooled planes are constants)")
        grads = [torch.empty_like(p) for p in params]
        dimg, dsvec = [None] * G, [None] * G
        for g in range(G):
            gp = 0 if share else g
            for l in range(LAYERS):
                w, b, ga, be = grads[16 * gp + 4 * l:16 * gp + 4 * l + 4]
                args.g[gp].w[l], args.g[gp].bias[l], args.g[gp].gamma[l], args.g[gp].beta[l] = (
                    w.data_ptr(), b.data_ptr(), ga.data_ptr(), be.data_ptr())
            if n_in == G and need_in[g]:
                dimg[g] = torch.empty_like(imgs[g])
                dsvec[g] = torch.empty_like(svecs[g]) if svecs[g] is not None else None
            args.dimg[g] = None if dimg[g] is None else dimg[g].data_ptr()
            args.dsvec[g] = None if dsvec[g] is None else dsvec[g].data_ptr()
        ref = ctypes.byref(args)
        scratch = torch.empty((L.adaisp_trunk_train_scratch_bytes(ref) // 4,), dtype=torch.float32, device=dev)
        args.dfeat, args.scratch, args.scratch_bytes = dfeat.data_ptr(), scratch.data_ptr(), scratch.numel() * 4
        with torch.cuda.device(dev):
            _lib._check(L.adaisp_trunk_train_bwd(ref, _lib._stream()), "adaisp_trunk_train_bwd")
            if pargs is not None and any(need_in):
                for g in range(n_in):                       # the statistics' gradient joins the image planes' in place
                    on = dimg[g] is not None
                    pargs.dsvec[g] = dsvec[g].data_ptr() if on else None
                    pargs.dsmall_in[g] = pargs.dsmall[g] = dimg[g].data_ptr() if on else None
                _lib._check(L.adaisp_critic_planes_bwd(ctypes.byref(pargs), _lib._stream()), "adaisp_critic_planes_bwd")
        ctx.keep = None
        d_in = [dimg[i] if need_img[i] else None for i in range(n_in)]
        if pargs is not None:                               # the raw state vector is the head of the extended one
            S = pargs.n_state
            d_sv = [dsvec[i][:, :S].contiguous() if need_sv[i] else None for i in range(n_in)]
        else:
            d_sv = [dsvec[i] if need_sv[i] else None for i in range(n_in)]
        return (None, None, None, None, *d_in, *d_sv, *grads)


def trunk_features(trunks, imgs, svecs, share_params=False, critic_planes=False):
    """Features [G, B, output_dim] of G <= 2 trunk instances (before the trunk's dropout).
    trunks: G FeatureExtractor modules — with `share_params` the SAME module G times (the critic's two calls of an
    iteration: statistics per instance, running statistics updated in instance order, parameter gradients summed);
    imgs / svecs: one [B,3,64,64] / [B,S] pair for all instances, or one pair per instance. `critic_planes`: svecs are the
    raw state vectors; the critic's three hand statistics of the image planes (value.py:65-80) are appended inside the node
    (the trunk's first conv then has 3 + S + 3 input channels)."""
    G = len(trunks)
    if not 1 <= G <= MAX_G or len(imgs) not in (1, G) or len(svecs) != len(imgs):
        raise ValueError("trunk_features: 1-2 trunk instances, one input for all or one per instance")
    if share_params and any(t is not trunks[0] for t in trunks):
        raise ValueError("share_params: the instances must be one module")
    for t in trunks:
        if not serves(t, imgs[0], svecs[0], extra=3 if critic_planes else 0):
            raise _lib.AdaispError("trunk_features: this trunk / input is not served by the kernels (check `serves` first)")
    return _TrunkFn.apply(tuple(trunks), bool(share_params), len(imgs), bool(critic_planes), *imgs, *svecs,
                          *_param_list(trunks, share_params))
