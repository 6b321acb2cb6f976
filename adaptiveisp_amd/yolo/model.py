"""YOLOv3 (Darknet-53 backbone, 3-scale head) as a plain parameter tree.

The module classes and their nesting reproduce the state-dict layout of the reference's detector
(`model.N.conv.weight`, `model.N.bn.*`, `model.N.cv1/cv2.*`, `model.N.K.cv1.*` for repeated blocks,
`model.28.m.{0,1,2}.*`, `model.28.anchors`; yolov3/models/yolov3.yaml, common.py:45-59,110-120,
yolo.py:38-89) so a reference checkpoint's state dict loads unchanged. `forward()` here is the
plain-PyTorch fp32 formulation — the numerical reference the HIP engine (engine.py) is tested against,
and what runs when there is no GPU (tests). The product path on the MI355X is YoloEngine.
"""
import torch
import torch.nn as nn

ANCHORS = ((10, 13, 16, 30, 33, 23), (30, 61, 62, 45, 59, 119), (116, 90, 156, 198, 373, 326))
STRIDES = (8.0, 16.0, 32.0)

# (from, repeats, kind, args) — the yolov3.yaml graph. Conv args: (cout, k, s); Bottleneck args: (cout, shortcut)
GRAPH = (
    (-1, 1, "conv", (32, 3, 1)), (-1, 1, "conv", (64, 3, 2)), (-1, 1, "bneck", (64, True)),
    (-1, 1, "conv", (128, 3, 2)), (-1, 2, "bneck", (128, True)), (-1, 1, "conv", (256, 3, 2)),
    (-1, 8, "bneck", (256, True)), (-1, 1, "conv", (512, 3, 2)), (-1, 8, "bneck", (512, True)),
    (-1, 1, "conv", (1024, 3, 2)), (-1, 4, "bneck", (1024, True)),
    (-1, 1, "bneck", (1024, False)), (-1, 1, "conv", (512, 1, 1)), (-1, 1, "conv", (1024, 3, 1)),
    (-1, 1, "conv", (512, 1, 1)), (-1, 1, "conv", (1024, 3, 1)),
    (-2, 1, "conv", (256, 1, 1)), (-1, 1, "up", ()), ((-1, 8), 1, "cat", ()),
    (-1, 1, "bneck", (512, False)), (-1, 1, "bneck", (512, False)), (-1, 1, "conv", (256, 1, 1)),
    (-1, 1, "conv", (512, 3, 1)),
    (-2, 1, "conv", (128, 1, 1)), (-1, 1, "up", ()), ((-1, 6), 1, "cat", ()),
    (-1, 1, "bneck", (256, False)), (-1, 2, "bneck", (256, False)),
    ((27, 22, 15), 1, "detect", ()),
)


class Conv(nn.Module):
    """conv (no bias) -> BatchNorm(eps 1e-3, momentum 0.03) -> SiLU."""

    def __init__(self, c1, c2, k=1, s=1):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, k // 2, bias=False)
        self.bn = nn.BatchNorm2d(c2, eps=1e-3, momentum=0.03)
        self.act = nn.SiLU()

    def forward(self, x):
        return self.act(self.bn(self.conv(x)))

    def folded(self):
        """(weight [Cout,Cin,k,k], bias [Cout]) with the batch-norm folded in (eval statistics)."""
        scale = self.bn.weight / torch.sqrt(self.bn.running_var + self.bn.eps)
        return self.conv.weight * scale[:, None, None, None], self.bn.bias - self.bn.running_mean * scale


class Bottleneck(nn.Module):
    def __init__(self, c1, c2, shortcut=True, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_, c2, 3, 1)
        self.add = shortcut and c1 == c2

    def forward(self, x):
        y = self.cv2(self.cv1(x))
        return x + y if self.add else y


class Concat(nn.Module):
    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension

    def forward(self, xs):
        return torch.cat(xs, self.d)


class Detect(nn.Module):
    def __init__(self, nc=80, anchors=ANCHORS, ch=(256, 512, 1024)):
        super().__init__()
        self.nc, self.no, self.nl, self.na = nc, nc + 5, len(anchors), len(anchors[0]) // 2
        self.stride = torch.tensor(STRIDES)
        a = torch.tensor(anchors).float().view(self.nl, -1, 2)
        self.register_buffer("anchors", a / self.stride.view(-1, 1, 1))      # in grid units, like the reference
        self.m = nn.ModuleList(nn.Conv2d(c, self.no * self.na, 1) for c in ch)

    def forward(self, xs):
        z, raws = [], []
        for i, x in enumerate(xs):
            x = self.m[i](x)
            bs, _, ny, nx = x.shape
            x = x.view(bs, self.na, self.no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
            raws.append(x)
            if not self.training:
                yv, xv = torch.meshgrid(torch.arange(ny, device=x.device, dtype=x.dtype),
                                        torch.arange(nx, device=x.device, dtype=x.dtype), indexing="ij")
                grid = torch.stack((xv, yv), 2).expand(1, self.na, ny, nx, 2) - 0.5
                anchor_grid = (self.anchors[i] * self.stride[i]).view(1, self.na, 1, 1, 2)
                xy, wh, conf = x.sigmoid().split((2, 2, self.nc + 1), 4)
                y = torch.cat(((xy * 2 + grid) * self.stride[i], (wh * 2) ** 2 * anchor_grid, conf), 4)
                z.append(y.view(bs, self.na * nx * ny, self.no))
        return raws if self.training else (torch.cat(z, 1), raws)


def make_divisible(x, divisor=8):
    """Channel rounding of the reference's parse_model (yolov3/utils/general.py make_divisible)."""
    import math
    return math.ceil(x / divisor) * divisor


class DetectionModel(nn.Module):
    def __init__(self, cfg="yolov3.yaml", ch=3, nc=80, anchors=None, width=1.0):
        """`width`: the yaml's width_multiple (1.0 = yolov3.yaml); channels are make_divisible(c * width, 8)
        exactly as the reference's parse_model does (yolov3/models/yolo.py:299-356)."""
        super().__init__()
        layers, chans = [], []
        for frm, n, kind, args in GRAPH:
            if kind in ("conv", "bneck") and width != 1.0:
                args = (make_divisible(args[0] * width, 8),) + tuple(args[1:])
            i = len(layers)
            c_prev = ch if i == 0 else (chans[i + frm if frm < 0 else frm] if isinstance(frm, int) else None)
            if kind == "conv":
                m, c_out = Conv(c_prev, args[0], args[1], args[2]), args[0]
            elif kind == "bneck":
                blocks = [Bottleneck(c_prev if j == 0 else args[0], args[0], args[1]) for j in range(n)]
                m, c_out = (blocks[0] if n == 1 else nn.Sequential(*blocks)), args[0]
            elif kind == "up":
                m, c_out = nn.Upsample(None, 2, "nearest"), c_prev
            elif kind == "cat":
                m = Concat(1)
                c_out = sum(chans[i + j if j < 0 else j] for j in frm)
            else:
                m, c_out = Detect(nc, anchors or ANCHORS, tuple(chans[j] for j in frm)), None
            m.f, m.i = frm, len(layers)
            layers.append(m)
            chans.append(c_out)
        self.model = nn.Sequential(*layers)
        self.stride = torch.tensor(STRIDES)
        self.nc = nc
        self.names = [str(i) for i in range(nc)]
        for mod in self.modules():
            if isinstance(mod, nn.BatchNorm2d):
                mod.eps, mod.momentum = 1e-3, 0.03

    def forward(self, x):
        ys = []
        for m in self.model:
            if m.f != -1:
                x = ys[m.f] if isinstance(m.f, int) else [x if j == -1 else ys[j] for j in m.f]
            x = m(x)
            ys.append(x)
        return x


Model = DetectionModel


def yolov3(nc=80):
    return DetectionModel(nc=nc)
