"""yaml table -> model, and the model-level call convention of the reference
(ultralytics/nn/tasks.py: BaseModel :80-280, DetectionModel :283-318, YOLOv10DetectionModel :645,
YOLOv10_3DDetectionModel :649, parse_model :837-964, yaml_model_load :967).

Same yaml schema, same `model.{i}.…` state_dict keys, same `model(batch_dict) -> (loss*B, items)` /
`model(img) -> preds` dispatch.  Differences, all deliberate:
  * strides are derived from the table instead of probing a 256x256 CPU forward (the modules only run on HIP);
  * `kernel_size_1/2` default to 3 when a 3D yaml omits them (the reference raises TypeError there).
"""
from __future__ import annotations

import contextlib
import math
import os
import re
from copy import deepcopy
from types import SimpleNamespace

import torch
import torch.nn as nn

from ._lib import Y3DError
import yaml

from . import modules as M
from . import ops
from .modules import (C2f, C2fCIB, Concat, Conv, Detect, DWConv, PSA, SCDown, SPPF, Upsample, v10Detect, v10Detect3d)

CFG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cfg", "models")

# hyper-parameters the losses read (reference cfg/default.yaml:102-141), with the offline-only pins of SURVEY §0.5
DEFAULT_HYP = dict(box=5.0, cls=1.0, dfl=1.5, loss2d=2.0, depth=1.0, offset3d=10.0, size3d=1.0, heading=1.0,
                   tal_topk=8, tal_alpha=0.5, tal_beta=1.0, tal_gamma=1.0, tal_3d=True, tal_2d=True, kps_dist_metric="l1",
                   constrain_anchors=True, distillation=False, fgdm_loss=False, fgdm_supervision=False, htl=False)

_REGISTRY = {"Conv": Conv, "DWConv": DWConv, "C2f": C2f, "C2fCIB": C2fCIB, "SCDown": SCDown, "SPPF": SPPF, "PSA": PSA,
             "Concat": Concat, "nn.Upsample": Upsample, "Detect": Detect, "v10Detect": v10Detect, "v10Detect3d": v10Detect3d}


def make_divisible(x, divisor):
    return math.ceil(x / divisor) * divisor


def guess_model_scale(path):
    with contextlib.suppress(AttributeError):
        return re.search(r"yolov\d+([nsblmx])", os.path.splitext(os.path.basename(str(path)))[0]).group(1)
    return ""


def yaml_model_load(path):
    """reference tasks.py:967-985; bare names resolve inside the package's cfg/models tree"""
    p = str(path)
    if not os.path.exists(p):
        for sub in ("v10-3D", "v10", ""):
            q = os.path.join(CFG_DIR, sub, os.path.basename(p))
            if os.path.exists(q):
                p = q
                break
    with open(p) as f:
        d = yaml.safe_load(f)
    d["scale"] = guess_model_scale(p)
    d["yaml_file"] = p
    return d


def parse_model(d, ch, verbose=False):
    """reference tasks.py:837-964 for the module set of the v10 / v10-3D tables"""
    max_channels = float("inf")
    nc, scales = d.get("nc"), d.get("scales")
    depth, width = d.get("depth_multiple", 1.0), d.get("width_multiple", 1.0)
    if scales:
        scale = d.get("scale") or tuple(scales.keys())[0]
        depth, width, max_channels = scales[scale]
    ch = [ch]
    layers, save, c2 = [], [], ch[-1]
    for i, (f, n, m, args) in enumerate(d["backbone"] + d["head"]):
        if m not in _REGISTRY:
            raise NotImplementedError(f"module '{m}' is outside the YOLOv10 / YOLOv10-3D path")
        mod = _REGISTRY[m]
        args = list(args)
        for j, a in enumerate(args):
            if isinstance(a, str):
                args[j] = nc if a == "nc" else (None if a == "None" else a)
        n = n_ = max(round(n * depth), 1) if n > 1 else n
        if mod in (Conv, DWConv, C2f, C2fCIB, SCDown, SPPF, PSA):
            c1, c2 = ch[f], args[0]
            if c2 != nc:
                c2 = make_divisible(min(c2, max_channels) * width, 8)
            args = [c1, c2, *args[1:]]
            if mod in (C2f, C2fCIB):
                args.insert(2, n)
                n = 1
        elif mod is Concat:
            c2 = sum(ch[x] for x in f)
        elif mod in (Detect, v10Detect, v10Detect3d):
            args.append([ch[x] for x in f])
            if mod is v10Detect3d:
                for key in ("dsconv", "channels", "use_predecessors", "detach_predecessors", "deform", "common_head", "num_scales",
                            "half_channels", "fgdm_predictor", "kernel_size_1", "kernel_size_2"):
                    args.append(d.get(key))
        else:
            c2 = ch[f]
        m_ = nn.Sequential(*(mod(*args) for _ in range(n))) if n > 1 else mod(*args)
        m_.np = sum(x.numel() for x in m_.parameters())
        m_.i, m_.f, m_.type = i, f, m
        save.extend(x % i for x in ([f] if isinstance(f, int) else f) if x != -1)
        layers.append(m_)
        if i == 0:
            ch = []
        ch.append(c2)
    return nn.Sequential(*layers), sorted(save)


def table_strides(model: nn.Sequential, nl=None):
    """down-sampling factor of each detect input, derived from the layer table"""
    down = {}
    for m in model:
        f = m.f
        src = 1 if m.i == 0 else down[(f if f >= 0 else m.i + f)] if isinstance(f, int) else None
        if isinstance(m, (Detect, v10Detect3d)):
            out = [float(down[x]) for x in f]
            return out[:nl] if nl else out
        if isinstance(m, Conv):
            cur = src * m.s
        elif isinstance(m, SCDown):
            cur = src * m.cv2.s
        elif isinstance(m, Upsample):
            cur = src / 2
        elif isinstance(m, Concat):
            cur = down[f[0] if f[0] >= 0 else m.i + f[0]]
        else:
            cur = src
        down[m.i] = cur
    raise ValueError("no detect layer in the table")


def fuse_conv_and_bn(w, gamma, beta, mean, var, eps=1e-3):
    """reference utils/torch_utils.py:171-198: W' = diag(gamma / sqrt(var + eps)) W,  b' = beta - gamma * mean / sqrt(var + eps).
    One-off host-side weight transform (deployment export); the eval path itself applies the same scale/shift in the conv
    epilogue (y3d_conv2d_fwd_affine) and never materialises folded weights."""
    s = gamma / torch.sqrt(var + eps)
    return w * s.view(-1, 1, 1, 1), beta - mean * s


def fuse_repvggdw(m, eps=1e-3):
    """reference block.py:716-735 RepVGGDW.fuse: fold both BatchNorms and pad the 3x3 depth-wise filter into the 7x7 one"""
    w7, b7 = fuse_conv_and_bn(m.conv.conv.weight, m.conv.bn.weight, m.conv.bn.bias, m.conv.bn.running_mean, m.conv.bn.running_var, eps)
    w3, b3 = fuse_conv_and_bn(m.conv1.conv.weight, m.conv1.bn.weight, m.conv1.bn.bias, m.conv1.bn.running_mean, m.conv1.bn.running_var, eps)
    return w7 + torch.nn.functional.pad(w3, [2, 2, 2, 2]), b7 + b3


def folded_state_dict(model):
    """{conv-key: folded weight, conv-key-with-.bias: folded bias} for every Conv of the model (BaseModel.fuse, tasks.py:177-205)"""
    out = {}
    with torch.no_grad():
        for name, m in model.named_modules():
            if isinstance(m, Conv):
                w, b = fuse_conv_and_bn(m.conv.weight, m.bn.weight, m.bn.bias, m.bn.running_mean, m.bn.running_var, m.bn.eps)
                out[name + ".conv.weight"], out[name + ".conv.bias"] = w, b
    return out


def fp8_state_dict(model):
    """The 1-byte weight store of the `fp8w` mode (BASELINE configs[4]; csrc/fp8w.hip): {conv key: (codes uint8 OIHW, scale fp32 (Cout,))}
    for every dense / grouped Conv of the model (depth-wise filters and everything else stay in `state_dict()` form).  The codes are
    what `y3d.set_weight_quant("fp8")` multiplies with: value(code) * scale."""
    from . import ops
    out = {}
    seen = set()
    with torch.no_grad():
        for name, m in model.named_modules():
            if isinstance(m, Conv) and not (m.g > 1 and m.g == m.conv.in_channels and m.g == m.conv.out_channels):
                w = m.conv.weight.detach()
                if not w.is_cuda:
                    raise Y3DError("fp8_state_dict: the quantiser runs on the HIP device")
                if w.data_ptr() in seen:
                    continue
                seen.add(w.data_ptr())
                w32 = w.float().contiguous()
                rows, K = w32.shape[0], w32[0].numel()
                codes = torch.empty(w32.shape, dtype=torch.uint8, device=w.device)
                scale = torch.empty(rows, dtype=torch.float32, device=w.device)
                desc = torch.tensor([w32.data_ptr(), 0, codes.data_ptr(), scale.data_ptr(), rows, K], dtype=torch.int64, device=w.device)
                rb = torch.zeros(1, dtype=torch.int32, device=w.device)
                ops.lib().mt_fp8w_quantize(desc.data_ptr(), rb.data_ptr(), 1, rows, ops.stream())
                out[name + ".conv.weight"] = (codes, scale)
    return out


def load_fp8_state_dict(model, fp8_sd):
    """inverse of fp8_state_dict: writes value(code) * scale into the named conv weights (in place)"""
    from . import ops
    named = dict(model.named_parameters())
    with torch.no_grad():
        for k, (codes, scale) in fp8_sd.items():
            p = named[k]
            if not p.is_cuda:
                raise Y3DError("load_fp8_state_dict: the model must live on the HIP device")
            w = torch.empty(p.shape, dtype=torch.float32, device=p.device)
            ops.lib().fp8w_dequantize(codes.to(p.device).contiguous().data_ptr(), scale.to(p.device).float().contiguous().data_ptr(), w.data_ptr(),
                                      p.shape[0], p[0].numel(), ops.stream())
            p.copy_(w)
    return model


def initialize_weights(model):
    """reference utils/torch_utils.py:327-337"""
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.eps = 1e-3
            m.momentum = 0.03
        elif isinstance(m, (nn.SiLU,)):
            m.inplace = True


class BaseModel(nn.Module):
    """reference tasks.py:80-280"""

    def forward(self, x, *args, **kwargs):
        if isinstance(x, dict):
            return self.loss(x, *args, **kwargs)
        return self.predict(x, *args, **kwargs)

    def predict(self, x, profile=False, visualize=False, augment=False, embed=None):
        return self._predict_once(x)

    def _predict_once(self, x):
        y = []
        ops.reset_placement()
        layers = list(self.model)
        # cross-layer placement: the second pass over an input geometry knows every Concat row's channel layout and map size (recorded
        # by the first), so the rows that produce a concat's LATER members (skips from earlier layers) write them in place
        skips = self.__dict__.get("_skip_plan")
        if skips is None:
            skips = self.__dict__["_skip_plan"] = self._skip_members(layers)
        key = (tuple(x.shape), x.dtype, str(x.device), ops.compute_dtype()) if torch.is_tensor(x) else None
        shapes_all = self.__dict__.setdefault("_cat_shapes", {})
        shapes = shapes_all.get(key) if (key is not None and ops.PLACEMENT) else None
        rec, bufs = {}, {}
        for idx, m in enumerate(layers):
            if m.f != -1:
                x = y[m.f] if isinstance(m.f, int) else [x if j == -1 else y[j] for j in m.f]
            fin = (None, 0)
            if shapes is not None and idx in skips and skips[idx][0] in shapes and torch.is_tensor(x) and x.is_cuda:
                k, pos = skips[idx]
                tot, cs, h, w = shapes[k]
                if k not in bufs:
                    bufs[k] = ops.concat_buffer(x, tot, h, w)
                if bufs[k] is not None:
                    fin = (bufs[k], sum(cs[:pos]))
            if type(m) is Concat and isinstance(x, (list, tuple)) and all(torch.is_tensor(t) and t.dim() == 4 for t in x):
                rec[idx] = (sum(t.shape[1] for t in x), [t.shape[1] for t in x], x[0].shape[2], x[0].shape[3])
            slot = bufs.get(idx + 1) if (idx + 1) in bufs and self._first_member_fits(layers, idx, x, bufs[idx + 1], shapes) else self._concat_slot(layers, idx, x, y)
            with ops.place(slot, 0), ops.place_final(*fin):
                x = m(x)
            y.append(x if m.i in self.save else None)
        if key is not None and key not in shapes_all:
            shapes_all[key] = rec
        return x

    @staticmethod
    def _skip_members(layers):
        """{producer row: (concat row, member position)} for the members of every Concat row that come from an earlier row (not -1);
        a producer that feeds several concats is planned for the first one only"""
        plan = {}
        for k, m in enumerate(layers):
            if type(m) is Concat and isinstance(m.f, (list, tuple)):
                for pos, j in enumerate(m.f):
                    if j != -1:
                        src = j if j >= 0 else k + j
                        if 0 <= src < k - 1 and src not in plan:
                            plan[src] = (k, pos)
        return plan

    @staticmethod
    def _first_member_fits(layers, idx, x, buf, shapes):
        """the pre-allocated buffer of the NEXT row's concat also serves its first member (this row's output, a single-kernel producer)"""
        m, nxt = layers[idx], layers[idx + 1]
        if buf is None or shapes is None or (idx + 1) not in shapes or not (type(nxt) is Concat and nxt.f[0] == -1 and m.f == -1):
            return False
        if not (type(m) in (Upsample, Conv) and torch.is_tensor(x) and x.dim() == 4):
            return False
        tot, cs, h, w = shapes[idx + 1]
        c = x.shape[1] if type(m) is Upsample else m.conv.out_channels
        return c == cs[0] and buf.shape[1] == tot and buf.shape[2] == h and buf.shape[3] == w

    @staticmethod
    def _concat_slot(layers, idx, x, y):
        """A single-kernel producer (Upsample, Conv) whose output is the FIRST input of the next row's Concat writes it straight into the
        concat buffer (ops.place); the other inputs are copied in by ConcatFn.  None: nothing to place."""
        m = layers[idx]
        nxt = layers[idx + 1] if idx + 1 < len(layers) else None
        if not (type(nxt) is Concat and isinstance(nxt.f, (list, tuple)) and len(nxt.f) >= 2 and nxt.f[0] == -1 and m.f == -1
                and torch.is_tensor(x) and x.dim() == 4):
            return None
        others = [y[j] if j >= 0 else y[len(y) + 1 + j] if len(y) + 1 + j >= 0 else None for j in nxt.f[1:]]
        if any(o is None or not torch.is_tensor(o) for o in others):
            return None
        if type(m) is Upsample:
            c, h, w = x.shape[1], 2 * x.shape[2], 2 * x.shape[3]
        elif type(m) is Conv:
            c = m.conv.out_channels
            h = (x.shape[2] + 2 * m.p - m.k) // m.s + 1
            w = (x.shape[3] + 2 * m.p - m.k) // m.s + 1
        else:
            return None
        if any(o.shape[2] != h or o.shape[3] != w for o in others):
            return None
        return ops.concat_buffer(x, c + sum(o.shape[1] for o in others), h, w)

    def loss(self, batch, preds=None):
        if not hasattr(self, "criterion"):
            self.criterion = self.init_criterion()
        preds = self.forward(batch["img"]) if preds is None else preds
        return self.criterion(preds, batch)

    def init_criterion(self):
        raise NotImplementedError

    def load(self, weights, verbose=False):
        """reference tasks.py:249-262: transfer by intersecting state_dict keys/shapes"""
        sd = weights if isinstance(weights, dict) else weights.state_dict()
        own = self.state_dict()
        ok = {k: v for k, v in sd.items() if k in own and own[k].shape == v.shape}
        self.load_state_dict(ok, strict=False)
        return len(ok), len(own)


class DetectionModel(BaseModel):
    """reference tasks.py:283-318"""

    def __init__(self, cfg="yolov10s_3D.yaml", ch=3, nc=None, verbose=False):
        super().__init__()
        self.yaml = cfg if isinstance(cfg, dict) else yaml_model_load(cfg)
        ch = self.yaml["ch"] = self.yaml.get("ch", ch)
        if nc and nc != self.yaml["nc"]:
            self.yaml["nc"] = nc
        self.model, self.save = parse_model(deepcopy(self.yaml), ch=ch, verbose=verbose)
        self.names = {i: f"{i}" for i in range(self.yaml["nc"])}
        self.inplace = self.yaml.get("inplace", True)
        m = self.model[-1]
        if isinstance(m, (Detect, v10Detect3d)):
            m.stride = torch.tensor(table_strides(self.model, m.nl))
            self.stride = m.stride
            m.bias_init()
        else:
            self.stride = torch.Tensor([32])
        initialize_weights(self)
        self.args = SimpleNamespace(**DEFAULT_HYP)


class YOLOv10DetectionModel(DetectionModel):
    def init_criterion(self):
        from .loss import v10DetectLoss
        return v10DetectLoss(self)


class YOLOv10_3DDetectionModel(DetectionModel):
    def init_criterion(self):
        from .loss import DetectLoss3d
        return DetectLoss3d(self)
