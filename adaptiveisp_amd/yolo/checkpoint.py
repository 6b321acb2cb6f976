"""Checkpoint import.

The reference keeps two kinds of file:
  * AdaptiveISP checkpoints — `torch.save({'iter', 'agent_model', 'value_model', 'agent_optimizer',
    'value_optimizer'})` (train.py:475-485); eval reads `['agent_model']` (val_adaptiveisp.py:192). Plain state
    dicts: `load_isp_checkpoint` / `save_isp_checkpoint`.
  * the detector `yolov3.pt` — a PICKLED nn.Module (`ckpt['model']` / `ckpt['ema']`, train.py:109-115,
    models/experimental.py:73-89) whose classes live in the reference's `models.yolo` / `models.common`. Unpickling
    that normally imports and runs the reference's code. `load_detector_checkpoint` never does: a restricted
    unpickler resolves `models.*` (and anything else that is not on a short allow-list of torch / numpy
    reconstruction helpers) to inert placeholder classes, the parameter tree is read out of the placeholders as
    a state dict, and that is loaded into this package's DetectionModel (same key layout, yolo/model.py).
"""
import collections
import pickle

import torch
import torch.nn as nn

from .model import DetectionModel

ISP_KEYS = ("iter", "agent_model", "value_model", "agent_optimizer", "value_optimizer")


# ------------------------------------------------------------------------------------------------ ISP checkpoints
def save_isp_checkpoint(path, iteration, agent, value, agent_optimizer=None, value_optimizer=None):
    torch.save({"iter": int(iteration), "agent_model": agent.state_dict(), "value_model": value.state_dict(),
                "agent_optimizer": agent_optimizer.state_dict() if agent_optimizer is not None else None,
                "value_optimizer": value_optimizer.state_dict() if value_optimizer is not None else None}, path)


def load_isp_checkpoint(path, agent=None, value=None, map_location="cpu"):
    """Loads the dict; fills `agent` / `value` when given (strict key match: the reference's key layout)."""
    ckpt = torch.load(path, map_location=map_location, weights_only=True)
    if agent is not None:
        agent.load_state_dict(ckpt["agent_model"])
    if value is not None and ckpt.get("value_model") is not None:
        value.load_state_dict(ckpt["value_model"])
    return ckpt


# ------------------------------------------------------------------------------------------------ detector pickles
class _Inert:
    """Stands in for any class or callable the allow-list does not know: constructing / calling it runs nothing."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)

    def __reduce_ex__(self, protocol):       # never re-pickled with behaviour
        return (_Inert, ())


class _Shell(nn.Module):
    """Placeholder for a reference module class (models.yolo.*, models.common.*): holds parameters, buffers and
    children exactly as pickled; has no forward."""

    def __init__(self, *a, **k):
        super().__init__()


_SHELLS = {}


def _shell(module, name):
    key = f"{module}.{name}"
    if key not in _SHELLS:
        _SHELLS[key] = type(name, (_Shell,), {"__module__": "adaptiveisp_amd.yolo.checkpoint", "_ref_class": key})
    return _SHELLS[key]


_ALLOWED_EXACT = {
    ("collections", "OrderedDict"), ("builtins", "set"), ("builtins", "frozenset"), ("builtins", "slice"),
    ("builtins", "complex"), ("builtins", "bytearray"), ("builtins", "list"), ("builtins", "dict"), ("builtins", "tuple"),
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_parameter"), ("torch._utils", "_rebuild_tensor"),
    ("torch._utils", "_rebuild_parameter_with_state"), ("torch", "Size"), ("torch", "device"), ("torch", "dtype"),
    ("torch.serialization", "_get_layout"), ("torch._tensor", "_rebuild_from_type_v2"),
    ("numpy.core.multiarray", "scalar"), ("numpy.core.multiarray", "_reconstruct"), ("numpy", "dtype"), ("numpy", "ndarray"),
    ("numpy._core.multiarray", "scalar"), ("numpy._core.multiarray", "_reconstruct"),
    ("_codecs", "encode"),                       # numpy's byte-string payloads (latin1 round trip): a pure function
}
_ALLOWED_TORCH_TYPES = {"FloatStorage", "HalfStorage", "BFloat16Storage", "LongStorage", "IntStorage", "ShortStorage",
                        "CharStorage", "ByteStorage", "BoolStorage", "DoubleStorage", "float16", "float32", "float64",
                        "bfloat16", "int64", "int32", "int16", "int8", "uint8", "bool", "Tensor"}


class _RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == "__builtin__":                                           # protocol-2 spelling of builtins
            module = "builtins"
        if module.split(".")[0] == "models":                                  # the reference's module classes
            return _shell(module, name)
        if (module, name) in _ALLOWED_EXACT or (module == "torch" and name in _ALLOWED_TORCH_TYPES):
            return super().find_class(module, name)
        if module.startswith("torch.nn.modules.") and not name.startswith("_"):
            cls = super().find_class(module, name)
            if isinstance(cls, type) and issubclass(cls, nn.Module):
                return cls
        if module == "torch.nn.parameter" and name == "Parameter":
            return super().find_class(module, name)
        return _Inert                      # opt namespaces, paths, loggers, callbacks...: inert, never imported


class _SafePickle:
    """The `pickle_module` torch.load asks for."""
    __name__ = "adaptiveisp_amd.yolo.checkpoint._SafePickle"
    Unpickler = _RestrictedUnpickler
    UnpicklingError = pickle.UnpicklingError

    @staticmethod
    def load(f, **kw):
        return _RestrictedUnpickler(f, **kw).load()


def _defuse(sd):
    """A checkpoint saved after `fuse()` (conv bias, no bn) -> the unfused key layout with an identity BN."""
    out = collections.OrderedDict(sd)
    for k in list(sd):
        if k.endswith(".conv.bias"):
            base = k[:-len("conv.bias")]
            if base + "bn.weight" in sd:
                continue
            b = out.pop(k).float()
            eps = 1e-3
            out[base + "bn.weight"] = torch.ones_like(b)
            out[base + "bn.bias"] = b
            out[base + "bn.running_mean"] = torch.zeros_like(b)
            out[base + "bn.running_var"] = torch.full_like(b, 1.0 - eps)        # (var + eps) == 1 exactly in fp32
            out[base + "bn.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    return out


def read_detector_pickle(path):
    """-> dict(state_dict (fp32), nc, names, anchors [nl,na,2] in pixels or None, source ('ema'|'model'), epoch)."""
    ckpt = torch.load(path, map_location="cpu", pickle_module=_SafePickle, weights_only=False)
    if isinstance(ckpt, nn.Module):
        ckpt = {"model": ckpt}
    src = "ema" if isinstance(ckpt.get("ema"), nn.Module) else "model"
    mod = ckpt[src]
    if not isinstance(mod, nn.Module):
        raise ValueError(f"{path}: ckpt['{src}'] is not a pickled module")
    sd = collections.OrderedDict((k, v.float() if v.is_floating_point() else v) for k, v in mod.state_dict().items())
    det = None
    for m in mod.modules():
        if getattr(type(m), "_ref_class", "").endswith(".Detect"):
            det = m
    nc = getattr(det, "nc", None) or getattr(mod, "nc", None)
    if nc is None and isinstance(getattr(mod, "yaml", None), dict):
        nc = mod.yaml.get("nc")
    names = getattr(mod, "names", None)
    if isinstance(names, (list, tuple)):
        names = dict(enumerate(names))
    anchors = None
    if det is not None and "anchors" in det._buffers and getattr(det, "stride", None) is not None:
        anchors = det._buffers["anchors"].float() * torch.as_tensor(det.stride).float().view(-1, 1, 1)
    return dict(state_dict=_defuse(sd), nc=int(nc) if nc is not None else None, names=names, anchors=anchors,
                source=src, epoch=ckpt.get("epoch") if isinstance(ckpt.get("epoch"), int) else None)


def load_detector_checkpoint(path, width=None):
    """yolov3.pt -> DetectionModel in eval mode with the checkpoint's weights (fp32). `width`: channel multiple of
    the pickled model (1.0 for the released yolov3.pt; inferred from the widest conv when None)."""
    info = read_detector_pickle(path)
    sd = info["state_dict"]
    if width is None:
        width = sd["model.9.conv.weight"].shape[0] / 1024.0      # the widest layer: channel rounding cannot hide it
    nc = info["nc"] if info["nc"] is not None else sd["model.28.m.0.bias"].shape[0] // 3 - 5
    anchors = None
    if info["anchors"] is not None:
        anchors = tuple(tuple(float(v) for v in a.reshape(-1)) for a in info["anchors"])
    model = DetectionModel(nc=nc, anchors=anchors, width=width)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    unexpected = [k for k in unexpected if not k.endswith("anchor_grid")]
    if missing or unexpected:
        raise ValueError(f"{path}: state dict does not match the YOLOv3 layout (missing {missing[:4]}, "
                         f"unexpected {unexpected[:4]})")
    if info["names"]:
        model.names = info["names"]
    return model.eval()
