"""Host-side mirror of the reference's nn.Modules for the YOLOv10 / YOLOv10-3D path.

Same class names, constructor signatures, attribute names and state_dict key layout as
ultralytics/nn/modules/{conv,block,head}.py — so the yaml tables and reference
checkpoints apply unchanged — but every forward/backward body runs on liby3d_hip.so
(see ops.py).  nn.Conv2d / nn.BatchNorm2d objects are kept as *parameter containers* only
(their own forward is never called).
"""
from __future__ import annotations

import copy
import math

import torch
import torch.nn as nn

from . import ops
from ._lib import Y3DError


def autopad(k, p=None, d=1):
    """reference conv.py:28-34"""
    if d > 1:
        k = d * (k - 1) + 1 if isinstance(k, int) else [d * (x - 1) + 1 for x in k]
    if p is None:
        p = k // 2 if isinstance(k, int) else [x // 2 for x in k]
    return p


def _flush_nbt(module, prefix, keep_vars):
    if module._nbt_pending and hasattr(module, "bn"):
        module.bn.num_batches_tracked += module._nbt_pending
        module._nbt_pending = 0


# ---- shape probe ----------------------------------------------------------------------------------------------------------------
# The reference's model constructor finds the detect strides by running forward(torch.zeros(1, ch, 256, 256)) on the HOST
# (nn/tasks.py:300-310) and reading the shapes of what comes back.  There is no host arithmetic in this package, so a module that is
# handed a tensor which is not on a HIP device answers with META tensors of its output shape: the probe (and thop / model.info(), which
# only look at shapes) works, and anything that asks such a result for a VALUE fails inside torch ("Cannot copy out of meta tensor").
# The kernel wrappers in ops.py keep raising Y3DError for host tensors; no value is ever computed on the host.
def _bump_epoch_on_load(module, incompatible_keys):
    """load_state_dict writes through torch (only `_version` counters move): the eval caches notice, a captured hipGraph of the forward
    (graph.GraphedForward keys on ops.PARAM_EPOCH) would not - so a load moves the parameter epoch too (round-3 advisor finding)"""
    ops.bump_weight_epoch()


def _host(x):
    t = x[0] if isinstance(x, (list, tuple)) else x
    return not t.is_cuda


def _meta(B, C, H, W):
    return torch.empty(int(B), int(C), int(H), int(W), device="meta")


class Conv(nn.Module):
    """reference conv.py:103-126: act(bn(conv(x))); one fused HIP sequence (conv -> BN stats -> BN+SiLU[+res])."""

    default_act = nn.SiLU()

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, d=1, act=True, deform=False):
        super().__init__()
        if deform:
            raise NotImplementedError("deform=True is not used by any v10 / v10-3D yaml (SURVEY §8c)")
        if isinstance(k, (tuple, list)):
            assert k[0] == k[1], "square kernels only"
            k = k[0]
        if d != 1:
            raise NotImplementedError("dilation != 1 is not on the YOLOv10 path")
        self.deform = False
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p, d), groups=g, dilation=d, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.act = self.default_act if act is True else act if isinstance(act, nn.Module) else nn.Identity()
        self._nbt_pending = 0
        self.register_state_dict_pre_hook(_flush_nbt)
        self.register_load_state_dict_post_hook(_bump_epoch_on_load)

    # kernel-facing views of the configuration
    @property
    def k(self):
        return self.conv.kernel_size[0]

    @property
    def s(self):
        return self.conv.stride[0]

    @property
    def p(self):
        return self.conv.padding[0]

    @property
    def g(self):
        return self.conv.groups

    @property
    def eps(self):
        return float(self.bn.eps)

    @property
    def momentum(self):
        return float(self.bn.momentum)

    @property
    def has_act(self):
        if isinstance(self.act, nn.SiLU):
            return True
        if isinstance(self.act, nn.Identity):
            return False
        raise Y3DError(f"activation {type(self.act).__name__} has no HIP kernel (SiLU / Identity only)")

    def out_shape(self, x):
        B, _, H, W = x.shape
        return B, self.conv.out_channels, (H + 2 * self.p - self.k) // self.s + 1, (W + 2 * self.p - self.k) // self.s + 1

    def forward(self, x, res=None, res_mode=0):
        """res_mode 1: act(bn(conv(x))) + res   (Bottleneck / CIB / PSA shortcuts)
           res_mode 2: act(bn(conv(x)) + res)   (RepVGGDW)"""
        if not x.is_cuda:
            return _meta(*self.out_shape(x))  # shape probe (see _host)
        return ops.ConvBNActFn.apply(x, self.conv.weight, self.bn.weight, self.bn.bias, res, res_mode if res is not None else 0, self)

    def forward_fuse(self, x, res=None, res_mode=0):
        """The forward the reference's BaseModel.fuse() switches a Conv to (nn/tasks.py:187-192: `m.conv = fuse_conv_and_bn(m.conv, m.bn)`,
        `delattr(m, "bn")`, `m.forward = m.forward_fuse`; conv.py:124-126): act(conv(x) + bias) with the folded weight and bias that
        `self.conv` now carries - the eval kernels' affine epilogue with scale 1 and shift = bias (+ residual as in `forward`)."""
        if not x.is_cuda:
            return _meta(*self.out_shape(x))
        if self.conv.bias is None:
            raise Y3DError("forward_fuse needs the folded conv (weight + bias) that BaseModel.fuse() installs")
        return ops.conv_bias_act_eval(x, self.conv.weight, self.conv.bias, self.k, self.s, self.p, self.g, self.has_act, res,
                                      res_mode if res is not None else 0, self.__dict__.setdefault("_eval_cache", {}))


class DWConv(Conv):
    """reference conv.py:172-177"""

    def __init__(self, c1, c2, k=1, s=1, d=1, act=True):
        super().__init__(c1, c2, k, s, g=math.gcd(c1, c2), d=d, act=act)


class Concat(nn.Module):
    """reference conv.py:394-404"""

    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension

    def forward(self, x):
        assert self.d == 1
        if _host(x):
            return _meta(x[0].shape[0], sum(t.shape[1] for t in x), *x[0].shape[2:])
        return cat(x)


def cat(xs):
    c = ops.ce(ops.compute_dtype())
    if all(t.shape[1] % c == 0 for t in xs):
        return ops.ConcatFn.apply(*xs)
    return torch.cat([t.to(ops.compute_dtype()) for t in xs], 1)  # channel counts that are not 16-byte chunks (plumbing)


class Upsample(nn.Module):
    """nn.Upsample(None, 2, 'nearest') rows of the yaml tables"""

    def __init__(self, size=None, scale_factor=None, mode="nearest"):
        super().__init__()
        if size is not None or int(scale_factor) != 2 or mode != "nearest":
            raise NotImplementedError("only nearest 2x upsampling is on the YOLOv10 path")
        self.scale_factor, self.mode = scale_factor, mode

    def forward(self, x):
        if not x.is_cuda:
            return _meta(x.shape[0], x.shape[1], 2 * x.shape[2], 2 * x.shape[3])
        return ops.Upsample2xFn.apply(x)


class Bottleneck(nn.Module):
    """reference block.py:327-342"""

    def __init__(self, c1, c2, shortcut=True, g=1, k=(3, 3), e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, k[0], 1)
        self.cv2 = Conv(c_, c2, k[1], 1, g=g)
        self.add = shortcut and c1 == c2

    def forward(self, x, place=None):
        """place = (buffer, channel offset): the block's output goes straight into that slice (ops.place)"""
        with ops.place(None, 0):
            h = self.cv1(x)
        with ops.place(*(place or (None, 0))):
            return self.cv2(h, x, 1) if self.add else self.cv2(h)


class C2f(nn.Module):
    """reference block.py:216-239"""

    def __init__(self, c1, c2, n=1, shortcut=False, g=1, e=0.5):
        super().__init__()
        self.c = int(c2 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(Bottleneck(self.c, self.c, shortcut, g, k=((3, 3), (3, 3)), e=1.0) for _ in range(n))

    def forward(self, x):
        if not x.is_cuda:
            return self.cv2(x)  # shape probe: spatial size kept, cv2's output channels
        # cv1 and the blocks write into their slices of ONE buffer: the reference's torch.cat (block.py:236) costs nothing
        buf = ops.concat_buffer(x, (2 + len(self.m)) * self.c)
        with ops.place(buf, 0):
            t = self.cv1(x)
        if t.is_cuda and t.shape[1] == 2 * self.c and self.c % ops.ce(ops.compute_dtype()) == 0:
            y0, y1, feed = ops.C2fSplitFn.apply(t)  # y1 twice: once for the concat, once for the first block
        else:
            y0, y1 = t.chunk(2, 1)
            feed = y1
        y = [y0, y1]
        for i, m in enumerate(self.m):
            feed = m(feed, place=(buf, (2 + i) * self.c)) if isinstance(m, (Bottleneck, CIB)) else m(feed)
            y.append(feed)
        h = cat(y)
        with ops.final_place():  # a later Concat row that takes this block's output gets it written in place (tasks._predict_once)
            return self.cv2(h)


class SPPF(nn.Module):
    """reference block.py:158-178"""

    def __init__(self, c1, c2, k=5):
        super().__init__()
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * 4, c2, 1, 1)
        self.k = k

    def forward(self, x):
        if not x.is_cuda:
            return self.cv2(x)  # shape probe
        c_ = self.cv1.conv.out_channels
        buf = ops.concat_buffer(x, 4 * c_)
        ys = []
        with ops.place(buf, 0):
            ys.append(self.cv1(x))
        for i in range(1, 4):
            with ops.place(buf, i * c_):
                ys.append(ops.MaxPoolFn.apply(ys[-1], self.k))
        return self.cv2(cat(ys))


class RepVGGDW(nn.Module):
    """reference block.py:702-735: SiLU(dw7x7+BN + dw3x3+BN)"""

    def __init__(self, ed):
        super().__init__()
        self.conv = Conv(ed, ed, 7, 1, 3, g=ed, act=False)
        self.conv1 = Conv(ed, ed, 3, 1, 1, g=ed, act=False)
        self.dim = ed
        self.act = nn.SiLU()
        self.conv.act = nn.SiLU()  # the 7x7 branch applies the SiLU after adding the 3x3 branch (res_mode 2)

    def forward(self, x):
        return self.conv(x, self.conv1(x), 2)


class CIB(nn.Module):
    """reference block.py:737-758"""

    def __init__(self, c1, c2, shortcut=True, e=0.5, lk=False):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = nn.Sequential(
            Conv(c1, c1, 3, g=c1),
            Conv(c1, 2 * c_, 1),
            Conv(2 * c_, 2 * c_, 3, g=2 * c_) if not lk else RepVGGDW(2 * c_),
            Conv(2 * c_, c2, 1),
            Conv(c2, c2, 3, g=c2),
        )
        self.add = shortcut and c1 == c2

    def forward(self, x, place=None):
        y = x
        for i in range(4):
            y = self.cv1[i](y)
        with ops.place(*(place or (None, 0))):  # the closing depth-wise conv writes into the enclosing C2fCIB's concat slice
            return self.cv1[4](y, x, 1) if self.add else self.cv1[4](y)


class C2fCIB(C2f):
    """reference block.py:760-768"""

    def __init__(self, c1, c2, n=1, shortcut=False, lk=False, g=1, e=0.5):
        super().__init__(c1, c2, n, shortcut, g, e)
        self.m = nn.ModuleList(CIB(self.c, self.c, shortcut, e=1.0, lk=lk) for _ in range(n))


class Attention(nn.Module):
    """reference block.py:771-797"""

    def __init__(self, dim, num_heads=8, attn_ratio=0.5):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.key_dim = int(self.head_dim * attn_ratio)
        self.scale = self.key_dim ** -0.5
        nh_kd = self.key_dim * num_heads
        h = dim + nh_kd * 2
        self.qkv = Conv(dim, h, 1, act=False)
        self.proj = Conv(dim, dim, 1, act=False)
        self.pe = Conv(dim, dim, 3, 1, g=dim, act=False)

    def forward(self, x, res=None):
        """returns proj(attn(x) + pe(v)) (+ res when given: the `b + attn(b)` of PSA.forward)"""
        qkv = self.qkv(x)
        o, v = ops.AttentionFn.apply(qkv, self.num_heads, self.key_dim, self.head_dim, self.scale)
        y = self.pe(v, o, 1)  # bn(dw3x3(v)) + o
        return self.proj(y, res, 1) if res is not None else self.proj(y)


class PSA(nn.Module):
    """reference block.py:799-818"""

    def __init__(self, c1, c2, e=0.5):
        super().__init__()
        assert c1 == c2
        self.c = int(c1 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1, 1)
        self.cv2 = Conv(2 * self.c, c1, 1)
        self.attn = Attention(self.c, attn_ratio=0.5, num_heads=self.c // 64)
        self.ffn = nn.Sequential(Conv(self.c, self.c * 2, 1), Conv(self.c * 2, self.c, 1, act=False))

    def forward(self, x):
        if not x.is_cuda:
            return self.cv2(x)  # shape probe
        a, b = self.cv1(x).split((self.c, self.c), dim=1)
        b = self.attn(b, res=b)
        b = self.ffn[1](self.ffn[0](b), b, 1)
        h = cat((a, b))
        with ops.final_place():
            return self.cv2(h)


class SCDown(nn.Module):
    """reference block.py:820-827"""

    def __init__(self, c1, c2, k, s):
        super().__init__()
        self.cv1 = Conv(c1, c2, 1, 1)
        self.cv2 = Conv(c2, c2, k=k, s=s, g=c2, act=False)

    def forward(self, x):
        return self.cv2(self.cv1(x))


class DFL(nn.Module):
    """reference block.py:44-62 (parameter container; the expectation is computed in head/loss code)"""

    def __init__(self, c1=16):
        super().__init__()
        self.conv = nn.Conv2d(c1, 1, 1, bias=False).requires_grad_(False)
        self.conv.weight.data[:] = torch.arange(c1, dtype=torch.float).view(1, c1, 1, 1)
        self.c1 = c1

    def forward(self, x):
        b, _, a = x.shape
        return (x.view(b, 4, self.c1, a).softmax(2) * self.conv.weight.view(1, 1, self.c1, 1).to(x.dtype)).sum(2)


def make_anchors(shapes, strides, device, grid_cell_offset=0.5):
    """reference utils/tal.py:300-312 from (h, w) shapes"""
    pts, st = [], []
    for (h, w), s in zip(shapes, strides):
        sx = torch.arange(w, device=device, dtype=torch.float32) + grid_cell_offset
        sy = torch.arange(h, device=device, dtype=torch.float32) + grid_cell_offset
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        pts.append(torch.stack((xx, yy), -1).view(-1, 2))
        st.append(torch.full((h * w, 1), float(s), dtype=torch.float32, device=device))
    return torch.cat(pts), torch.cat(st)


def _proj(branches, feats):
    """final 1x1+bias convs of several branches -> one NHWC map (their torch.cat)"""
    n = len(branches)
    return ops.HeadProjFn.apply(n, *feats, *[b.weight for b in branches], *[b.bias for b in branches])


def _conv_over_cat(m, parts):
    """Conv module `m` over the channel concatenation of `parts` when the total is not a whole number of 16-byte chunks (the
    `use_predecessors` inputs: ch + 3 / 6 / 7 channels): input and weight are zero-padded along the input channels to the next chunk
    multiple (zero weights on zero channels: the same sums), then the ordinary HIP conv -> BatchNorm -> SiLU sequence runs; autograd
    slices the padded gradients back (plumbing around the kernels)."""
    dt = ops.compute_dtype()
    c = ops.ce(dt)
    cin = sum(t.shape[1] for t in parts)
    cp = (cin + c - 1) // c * c
    parts = [t.to(dt) for t in parts]
    if cp != cin:
        B, _, H, W = parts[0].shape
        parts.append(torch.zeros(B, cp - cin, H, W, dtype=dt, device=parts[0].device))
    w = m.conv.weight if cp == cin else torch.nn.functional.pad(m.conv.weight, (0, 0, 0, 0, 0, cp - cin))
    return ops.ConvBNActFn.apply(torch.cat(parts, 1), w, m.bn.weight, m.bn.bias, None, 0, m)


class v10Detect3d(nn.Module):
    """reference head.py:545-975.  The shipped yamls leave every constructor switch off; with `dsconv`, `half_channels` or
    `use_predecessors` on, the head is built and run branch by branch (head.py:629-650, 718-743) on the same HIP ops - the
    sibling-branch fusion of the default head does not apply.  Not available, with the reference's own state of these switches:
    `common_head` (its training forward fails in the reference, head.py:745-746: three layers asserted, `build_small_head` makes two),
    `use_predecessors` in eval mode (the reference's patch path, head.py:709, hands the branches bare feature patches: channel
    mismatch), `deform` / `fgdm_predictor` (DCN / depth-predictor code outside the YOLOv10 path)."""

    max_det = 50
    dynamic = False
    export = False
    shape = None
    fused = True  # sibling-branch fusion of the training forward (set False for the per-branch reference form)
    PREDECESSORS = {"cls": (), "o2d": (), "s2d": (), "o3d": ("cls",), "s3d": ("cls",), "hd": ("cls",), "dep": ("cls", "s3d"),
                    "dep_un": ("cls", "s3d", "dep")}  # head.py:585-594

    def __init__(self, nc=80, ch=(), dsconv=False, channels=None, use_predecessors=False, detach_predecessors=True,
                 deform=False, common_head=False, num_scales=3, half_channels=False, fgdm_predictor=False,
                 kernel_size_1=3, kernel_size_2=3):
        super().__init__()
        assert channels is not None
        for name, flag in (("deform", deform), ("fgdm_predictor", fgdm_predictor)):
            if flag:
                raise NotImplementedError(f"{name}=True is not used by any shipped v10-3D yaml")
        if common_head:
            raise NotImplementedError("common_head=True: the reference's own training forward fails on it (head.py:745-746 asserts three "
                                      "layers per branch, build_small_head makes two); there is no behaviour to reproduce")
        if dsconv and use_predecessors:
            raise NotImplementedError("dsconv with use_predecessors: depth-wise convs over ch + 3 / 6 / 7 channels have no 16-byte-chunk kernel")
        kernel_size_1 = 3 if kernel_size_1 is None else kernel_size_1  # reference bug: tasks.py:940 passes None
        kernel_size_2 = 3 if kernel_size_2 is None else kernel_size_2
        self.nc = nc
        self.nl = num_scales
        self.dsconv, self.half_channels = bool(dsconv), bool(half_channels)
        self.use_predecessors, self.detach_predecessors = bool(use_predecessors), detach_predecessors
        self.generic = self.dsconv or self.half_channels or self.use_predecessors  # branch-by-branch forward
        self.predecessors = dict(self.PREDECESSORS)
        self.output_channels = {"cls": nc, "o2d": 2, "s2d": 2, "o3d": 2, "s3d": 3, "hd": 24, "dep": 1, "dep_un": 1}
        self.no = sum(self.output_channels.values())
        self.stride = torch.zeros(self.nl)
        self.kernel_size_1, self.kernel_size_2 = kernel_size_1, kernel_size_2
        self.patch_size = (kernel_size_1 - 1) + (kernel_size_2 - 1) + 1
        self.dep_norm = 65.0
        ch = [ch[i] for i in range(self.nl)]
        for name, out in self.output_channels.items():
            extra = sum(self.output_channels[q] for q in self.predecessors[name]) if self.use_predecessors else 0
            setattr(self, name, self.build_head([c + extra for c in ch], channels[name + "_c"], out))
        self.o2o_heads = nn.ModuleList([self.cls, self.o2d, self.s2d, self.o3d, self.s3d, self.hd, self.dep, self.dep_un])
        self.o2m_heads = copy.deepcopy(self.o2o_heads)
        self.register_load_state_dict_post_hook(_bump_epoch_on_load)

    def build_head(self, in_channels, mid, out):
        last = mid // 2 if self.half_channels else mid
        return nn.ModuleList(nn.Sequential(self.build_conv(x, mid, self.kernel_size_1, self.dsconv), self.build_conv(mid, last, self.kernel_size_2, self.dsconv),
                                           nn.Conv2d(last, out, 1)) for x in in_channels)

    @staticmethod
    def build_conv(c1, c2, k, dsconv):
        """head.py:645-650"""
        return nn.Sequential(Conv(c1, c1, k, g=c1), Conv(c1, c2, 1)) if dsconv else Conv(c1, c2, k)

    # ---- dense (training) path: head.py:718-753 -------------------------------------------------------------
    def forward_feat(self, x, heads):
        """per-branch form (one head set): the reference's own loop, head.py:718-743.  The default head's training forward uses the fused
        form below; the constructor switches run here."""
        ys, embs = [], []
        names = list(self.output_channels)
        for i in range(self.nl):
            feats, outs, emb = [], {}, None
            for j, module in enumerate(heads):
                br = module[i]
                pre = self.predecessors[names[j]] if self.use_predecessors else ()
                if pre:  # head.py:727-737: level map + detached earlier outputs (depth / 65) along the channels
                    e = _conv_over_cat(br[0], [x[i]] + [(outs[q] / self.dep_norm if q == "dep" else outs[q]).detach() for q in pre])
                else:
                    e = br[0](x[i])
                if j == 6:
                    emb = e
                if self.use_predecessors:
                    outs[names[j]] = _proj([br[2]], [br[1](e)])
                else:
                    feats.append(br[1](e))
            if self.use_predecessors:
                dt = ops.compute_dtype()
                ys.append(torch.cat([o.to(dt) for o in outs.values()], 1))
            else:
                ys.append(_proj([module[i][2] for module in heads], feats))
            embs.append(emb)
        return ys, embs

    def _stacks(self, i):
        """StackedConvs of level i over [o2o branches 0..7, o2m branches 0..7] for layer 1 and layer 2.
        -> (branches, mids, s1, s2, parts).  Uniform widths: s2 is ONE grouped stack, parts is None, the stack order is the branch
        order.  cls wider than the seven regression branches (M-3D: 128 vs 64): `branches` / `mids` are in STACK order
        [o2o cls, o2o regs, o2m regs, o2m cls] (the one-to-one half stays first: its input is detached), s2 is None and parts lists the
        second layer as (lo, hi, stack, groups) over z1's channels: cls | 14 regression branches as one grouped conv | cls;
        `pos[c]` maps the canonical branch index c (o2o 0..7, o2m 8..15) to its stack position."""
        key = (i, id(self.o2o_heads), id(self.o2m_heads))
        cache = self.__dict__.setdefault("_stack_cache", {})
        if key not in cache:
            canon = [h[i] for h in self.o2o_heads] + [h[i] for h in self.o2m_heads]
            cm = [b[0].conv.out_channels for b in canon]
            k2 = [b[1].conv.kernel_size for b in canon]
            if len(set(cm)) == 1:
                s1 = ops.StackedConvs([b[0] for b in canon])
                s2 = ops.StackedConvs([b[1] for b in canon], groups=len(canon))
                cache[key] = (canon, cm, s1, s2, None, list(range(16)))
            elif len(set(cm[1:8] + cm[9:16])) == 1 and cm[0] == cm[8] and len(set(k2)) == 1 and cm[1] % 8 == 0 and cm[0] % 8 == 0:
                order = list(range(0, 8)) + list(range(9, 16)) + [8]
                branches = [canon[c] for c in order]
                mids = [cm[c] for c in order]
                pos = [order.index(c) for c in range(16)]
                s1 = ops.StackedConvs([b[0] for b in branches])
                c, m = mids[0], mids[1]
                parts = [(0, c, ops.StackedConvs([branches[0][1]]), 1),
                         (c, c + 14 * m, ops.StackedConvs([b[1] for b in branches[1:15]], groups=14), 14),
                         (c + 14 * m, 2 * c + 14 * m, ops.StackedConvs([branches[15][1]]), 1)]
                cache[key] = (branches, mids, s1, None, parts, pos)
            else:
                s1 = ops.StackedConvs([b[0] for b in canon])
                cache[key] = (canon, cm, s1, None, None, list(range(16)))
        return cache[key]

    def restack(self):
        """(re)establish the stacked parameter storage of the fused training forward now (e.g. before wrapping the model in
        DistributedDataParallel, after .to(device) / load_state_dict(assign=True) / deepcopy)"""
        if self.generic:
            return  # branch-by-branch head: every Conv keeps its own parameters
        for i in range(self.nl):
            _, _, s1, s2, parts, _ = self._stacks(i)
            s1.tensors()
            if s2 is not None:
                s2.tensors()
            for part in parts or ():
                part[2].tensors()

    def forward_train_fused(self, x):
        """Both head sets of one level as: ONE 3x3 conv Cin -> sum(mid) (the one-to-one half contributes no input gradient:
        it sees x.detach(), head.py:820), ONE grouped conv (16 groups of mid -> mid), ONE projection launch set writing the
        (B, 2*no, H, W) map.  Numerically identical to the per-branch form (BatchNorm is per channel)."""
        o2o, o2m, e_o2o, e_o2m = [], [], [], []
        self._maps = []  # the (B, 2*no, H, W) maps the two head sets are channel halves of: the fused loss takes them whole
        for i in range(self.nl):
            branches, mids, s1, s2, parts, pos = self._stacks(i)
            half = sum(mids[:8])
            with ops.want_fp8_copy():  # fp8 convolutions on: layer 1's BatchNorm + SiLU pass also writes the fp8 copy layer 2 reads
                z1 = ops.FusedConvBNActFn.apply(x[i], s1, 1, (half, sum(mids)), *s1.params())
            offs = [sum(mids[:j]) for j in range(16)]
            if s2 is not None and mids[0] % 64 == 0 and getattr(self, "fuse_bn_proj", True):
                # grouped conv + BatchNorm statistics + projections with BatchNorm/SiLU applied on the fly (no activation tensor)
                out = ops.FusedConvBNProjFn.apply(z1, s2, len(branches), offs, mids, 16, *s2.params(), *[b[2].weight for b in branches],
                                                  *[b[2].bias for b in branches])
            elif s2 is not None:
                z2 = ops.FusedConvBNActFn.apply(z1, s2, len(branches), None, *s2.params())
                out = ops.HeadProjSlicesFn.apply(z2, offs, mids, 16, *[b[2].weight for b in branches], *[b[2].bias for b in branches])
            elif parts is not None:
                # cls | 14 regression branches (one grouped conv) | cls over channel views of z1; every split hands ONE gradient back
                views = ops.SplitChannelsFn.apply(z1, [lo for lo, _, _, _ in parts], [hi - lo for lo, hi, _, _ in parts])
                fuse = getattr(self, "fuse_bn_proj", True)
                outs_s, feats_s, k = [None] * 16, [None] * 16, 0
                for v, (lo, hi, st2, g) in zip(views, parts):
                    brs, w = branches[k:k + g], (hi - lo) // g
                    if fuse and w % 64 == 0:
                        # this part's grouped conv + BatchNorm statistics + its projections with BatchNorm / SiLU on the fly (the matrix-core
                        # kernels of proj_bn_mfma.hip at 64 / 128 channels), as on the uniform heads: no activation tensor, no per-branch
                        # VALU projection launches (M-3D: 32 launches of proj_bwd_weight + 64 slab folds per step)
                        o = ops.FusedConvBNProjFn.apply(v, st2, g, [j * w for j in range(g)], [w] * g, g, *st2.params(), *[b[2].weight for b in brs],
                                                        *[b[2].bias for b in brs])
                        off = 0
                        for j, b in enumerate(brs):
                            co = b[2].weight.shape[0]
                            outs_s[k + j] = o[:, off:off + co]
                            off += co
                    else:
                        z2 = ops.FusedConvBNActFn.apply(v, st2, g, None, *st2.params())
                        fs = [z2] if g == 1 else ops.SplitChannelsFn.apply(z2, [j * w for j in range(g)], [w] * g)
                        for j, f in enumerate(fs):
                            feats_s[k + j] = f
                    k += g
                if all(o is not None for o in outs_s):
                    out = cat([outs_s[pos[c]] for c in range(16)])
                elif all(f is not None for f in feats_s):
                    canon = [branches[pos[c]] for c in range(16)]
                    out = _proj([b[2] for b in canon], [feats_s[pos[c]] for c in range(16)])
                else:  # mixed: project the un-fused branches one by one, then line the 16 outputs up
                    for j in range(16):
                        if outs_s[j] is None:
                            outs_s[j] = _proj([branches[j][2]], [feats_s[j]])
                    out = cat([outs_s[pos[c]] for c in range(16)])
            else:
                feats = [b[1](zj) for b, zj in zip(branches, ops.SplitChannelsFn.apply(z1, offs, mids))]
                out = _proj([b[2] for b in branches], feats)
            self._maps.append(out)
            o2o.append(out[:, : self.no])
            o2m.append(out[:, self.no:])
            e_o2o.append(z1[:, offs[pos[6]]:offs[pos[6]] + mids[pos[6]]])
            e_o2m.append(z1[:, offs[pos[14]]:offs[pos[14]] + mids[pos[14]]])
        return o2o, o2m, e_o2o, e_o2m

    # ---- sparse (eval) path: head.py:656-716 -----------------------------------------------------------------
    def select_candidates(self, scores):
        """(B, K) int32 flat cell indices of the top-`max_det` max-class logits (head.py:686-692), HIP kernel"""
        B, nc, H, W = scores.shape
        if not ops.px_dense(scores):
            scores = ops._dense_any(scores, scores.dtype)
        idx = torch.empty(B, self.max_det, dtype=torch.int32, device=scores.device)
        ops.lib().topk_cells(ops.code(scores.dtype), scores.data_ptr(), scores.stride(3), B, H * W, nc, self.max_det, idx.data_ptr(), ops.stream())
        return idx

    def inference_forward_feat(self, x, heads):
        if self.use_predecessors:
            raise RuntimeError("use_predecessors has no eval path: the reference's patch forward (head.py:694-716) feeds the branches bare "
                               "feature patches and fails on the channel count")
        if ops.EVAL_LEVEL_STREAMS and self.nl > 1 and x[0].is_cuda and torch.cuda.is_current_stream_capturing():
            # the levels are independent chains of ~9 small launches each (1 600 patches, 40x40 / 20x20 maps: none fills 256 CUs for
            # long): inside a capture each level is recorded on its own stream, so the hipGraph holds nl parallel branches between the
            # neck and the decode.  Eagerly the host is the bound and one stream is kept.
            cur = torch.cuda.current_stream()
            pool = ops.level_streams(self.nl)
            ys = [None] * self.nl
            for i in range(self.nl):
                pool[i].wait_stream(cur)
                with torch.cuda.stream(pool[i]):
                    ys[i] = self._inference_level(x, heads, i)
            for i in range(self.nl):
                cur.wait_stream(pool[i])
            return ys
        return [self._inference_level(x, heads, i) for i in range(self.nl)]

    def _inference_level(self, x, heads, i):
        ps = self.patch_size
        L = ops.lib()
        xi = ops.to_nhwc(x[i], ops.compute_dtype())
        B, C, H, W = xi.shape
        if H * W < self.max_det:
            raise ValueError(f"level {i} has {H * W} cells < max_det={self.max_det} (reference head.py:690 needs H*W >= max_det)")
        dt, st = ops.code(xi.dtype), ops.stream()
        cls = _proj([heads[0][i][2]], [heads[0][i][1](heads[0][i][0](xi))])
        idx = self.select_candidates(cls)  # (B, K) int32
        K = idx.shape[1]
        patches = ops.nhwc_empty(B * K, C, ps, ps, xi.dtype, xi.device)
        sb, sh, sw = ops.s3(xi)
        L.patch_gather(dt, xi.data_ptr(), sb, sh, sw, idx.data_ptr(), patches.data_ptr(), B, H, W, C, K, ps, st)
        # a model folded by the reference's BaseModel.fuse() (no .bn on the branch Convs) runs branch by branch on forward_fuse
        folded = not hasattr(heads[1][i][0], "bn") if isinstance(heads[1][i][0], Conv) else False
        _, mids, s1, s2, _, _ = self._stacks(i) if not (self.generic or folded) else (None,) * 6
        if heads is self.o2o_heads and s2 is not None:
            # the 7 regression branches of the one-to-one set as one stacked conv + one grouped conv on the patches
            # (channel rows mid..8*mid of the training-time stacks), both unpadded: patch semantics of head.py:706-708
            mid, c0 = mids[0], heads[1][i][0]
            lo, hi = mid, 8 * mid
            w1, g1, b1, rm1, rv1 = s1.tensors()
            z1 = ops.conv_bn_act_eval(patches, w1[lo:hi], g1[lo:hi], b1[lo:hi], rm1[lo:hi], rv1[lo:hi], self.kernel_size_1, 1, 0, 1,
                                      c0.has_act, c0.eps, s1.__dict__.setdefault("_eval_cache", {}), ver=s1.ver)
            w2, g2, b2, rm2, rv2 = s2.tensors()
            z2 = ops.conv_bn_act_eval(z1, w2[lo:hi], g2[lo:hi], b2[lo:hi], rm2[lo:hi], rv2[lo:hi], self.kernel_size_2, 1, 0, 7,
                                      c0.has_act, c0.eps, s2.__dict__.setdefault("_eval_cache", {}), ver=s2.ver)
            reg = ops.proj_slices_eval(z2, [j * mid for j in range(7)], mid, [heads[j][i][2].weight for j in range(1, 8)],
                                       [heads[j][i][2].bias for j in range(1, 8)])[:, :, 0, 0]
        else:
            feats = []
            for j in range(1, 8):
                br = heads[j][i]
                # patch semantics, head.py:706-708: the branch's TOP-LEVEL Conv layers run unpadded (5x5 patch -> 1x1); the nested
                # Sequentials of `dsconv` keep their padding there, and so here (the 5x5 result is then read at cell (0, 0) like the
                # reference reads it).  Unlike the reference we do not leave the modules mutated.
                convs = [l for l in list(br)[:-1] if isinstance(l, Conv)]
                pads = [l.conv.padding for l in convs]
                for l in convs:
                    l.conv.padding = (0, 0)
                try:
                    f = patches
                    for l in list(br)[:-1]:
                        f = l(f)
                    feats.append(f)
                finally:
                    for l, p0 in zip(convs, pads):
                        l.conv.padding = p0
            reg = _proj([heads[j][i][2] for j in range(1, 8)], feats)[:, :, 0, 0]  # (BK, 35)
        reg = reg.contiguous()
        full = ops.nhwc_empty(B, self.no, H, W, cls.dtype, xi.device)
        L.head3d_scatter(dt, cls.data_ptr(), cls.stride(3), reg.data_ptr(), reg.stride(0), idx.data_ptr(), full.data_ptr(), B, H * W, self.nc,
                         self.no, K, st)
        return full

    def decode(self, ys):
        """head.py:755-797: (B, no, A) fp32 with xyxy px boxes and centre-3d px (HIP kernel over the per-level NHWC maps)."""
        import ctypes
        B = ys[0].shape[0]
        nl = len(ys)
        ys = [y if (y.dtype == ys[0].dtype and ops.px_dense(y) and y.stride(3) == self.no) else ops._dense_any(y, ys[0].dtype) for y in ys]
        A = sum(y.shape[2] * y.shape[3] for y in ys)
        out = torch.empty(B, self.no, A, dtype=torch.float32, device=ys[0].device)
        ops.lib().head3d_decode(ops.code(ys[0].dtype), nl, (ctypes.c_void_p * nl)(*[y.data_ptr() for y in ys]),
                                (ctypes.c_int * nl)(*[y.shape[2] for y in ys]), (ctypes.c_int * nl)(*[y.shape[3] for y in ys]),
                                (ctypes.c_float * nl)(*[float(s) for s in self.stride.tolist()[:nl]]), B, self.nc, out.data_ptr(), ops.stream())
        return out

    def _probe(self, x):
        """shape probe (see _host): the reference's constructor reads `forward(zeros)["one2many"]` shapes, nn/tasks.py:306-307"""
        maps = [_meta(xi.shape[0], self.no, *xi.shape[2:]) for xi in x]
        if not self.training:
            return {"one2one": (torch.empty(x[0].shape[0], self.no, sum(m.shape[2] * m.shape[3] for m in maps), device="meta"), maps), "o2o_embs": None}
        embs = [_meta(xi.shape[0], self.dep[i][0].conv.out_channels if isinstance(self.dep[i][0], Conv) else self.dep[i][0][-1].conv.out_channels,
                      *xi.shape[2:]) for i, xi in enumerate(x)]
        return {"one2many": maps, "one2one": list(maps), "o2m_embs": embs, "o2o_embs": list(embs), "depth_maps": torch.empty(1)}

    def forward(self, x):
        x = list(x[: self.nl])
        if _host(x):
            return self._probe(x)
        if not self.training:
            maps = self.inference_forward_feat([xi.detach() for xi in x], self.o2o_heads)
            return {"one2one": (self.decode(maps), maps), "o2o_embs": None}
        fused = self.fused and not self.generic
        if fused:
            one2one, one2many, o2o_embs, o2m_embs = self.forward_train_fused(x)
        else:
            one2one, o2o_embs = self.forward_feat([xi.detach() for xi in x], self.o2o_heads)
            one2many, o2m_embs = self.forward_feat(x, self.o2m_heads)
        out = {"one2many": one2many, "one2one": one2one, "o2m_embs": o2m_embs, "o2o_embs": o2o_embs, "depth_maps": torch.empty(1)}
        if fused:
            out["_y3d_maps"] = self.__dict__.pop("_maps")  # private extra next to the reference's keys (head.py:833)
        return out

    # head.py:847-871 as data: per number of levels, the depth bias of each level and the uniform range of its depth projection
    # weights; (bias fill, weight init) of the other branches.  The class prior assumes KITTI's 1280 x 384 images, as the reference does.
    DEPTH_PRIOR = {1: ((40.0,), ((-3.5, 3.5),)), 2: ((45.0, 20.0), ((-2.0, 2.0), (-2.0, 2.0))),
                   3: ((45.0, 25.0, 10.0), ((-2.0, 2.0), (-1.5, 1.5), (-1.0, 1.0)))}
    BIAS_FILL = {"s2d": 6.0, "o2d": 0.0, "o3d": 0.0, "s3d": 0.0}
    S3D_WEIGHT_STD = 0.05
    PRIOR_IMAGE = (1280.0, 384.0)

    def bias_init(self):
        """reference head.py:847-871; pinned by tests/golden/bias_init.npz (tests/test_host_logic.py)"""
        if self.nl not in self.DEPTH_PRIOR:
            raise RuntimeError("Initialization only set for 1 and 3 scales")
        deps, ranges = self.DEPTH_PRIOR[self.nl]
        for i in range(self.nl):
            s = float(self.stride[i])
            self.cls[i][-1].bias.data[: self.nc] = math.log(5 / self.nc / ((self.PRIOR_IMAGE[0] / s) * (self.PRIOR_IMAGE[1] / s)))
            for name, v in self.BIAS_FILL.items():
                getattr(self, name)[i][-1].bias.data.fill_(v)
            nn.init.normal_(self.s3d[i][-1].weight, std=self.S3D_WEIGHT_STD)
            self.dep[i][-1].bias.data.fill_(deps[i])
            nn.init.uniform_(self.dep[i][-1].weight, a=ranges[i][0], b=ranges[i][1])
        # the one-to-one set aliases the named branches, the one-to-many set restarts as its copy (head.py:869-870)
        self.o2o_heads = nn.ModuleList([self.cls, self.o2d, self.s2d, self.o3d, self.s3d, self.hd, self.dep, self.dep_un])
        self.o2m_heads = copy.deepcopy(self.o2o_heads)


class Detect(nn.Module):
    """reference head.py:22-109 (YOLOv8 detect head; base of v10Detect).
    Restrictions against the reference: the eval decode (`inference`) is the HIP kernel `y3d_head2d_decode` - head
    maps on a HIP device and reg_max == 16 (the value every shipped yaml uses; the constructor fixes it as the reference does);
    anything else raises instead of falling back to a host formulation."""

    dynamic = False
    export = False
    shape = None
    anchors = torch.empty(0)  # head.py:28-29; BaseModel._apply (nn/tasks.py:243-246) moves stride / anchors / strides with the model
    strides = torch.empty(0)

    def __init__(self, nc=80, ch=()):
        super().__init__()
        self.nc = nc
        self.nl = len(ch)
        self.reg_max = 16
        self.no = nc + self.reg_max * 4
        self.stride = torch.zeros(self.nl)
        c2, c3 = max((16, ch[0] // 4, self.reg_max * 4)), max(ch[0], min(self.nc, 100))
        self.cv2 = nn.ModuleList(nn.Sequential(Conv(x, c2, 3), Conv(c2, c2, 3), nn.Conv2d(c2, 4 * self.reg_max, 1)) for x in ch)
        self.cv3 = nn.ModuleList(nn.Sequential(Conv(x, c3, 3), Conv(c3, c3, 3), nn.Conv2d(c3, self.nc, 1)) for x in ch)
        self.dfl = DFL(self.reg_max)
        self.register_load_state_dict_post_hook(_bump_epoch_on_load)

    def forward_feat(self, x, cv2, cv3):
        ys = []
        for i in range(self.nl):
            f2 = cv2[i][1](cv2[i][0](x[i]))
            f3 = x[i]
            for sub in list(cv3[i])[:-1]:
                f3 = sub(f3)
            ys.append(_proj([cv2[i][2], cv3[i][2]], [f2, f3]))
        return ys

    def inference(self, ys):
        """head.py:53-79: (B, 4+nc, A) fp32: xywh px boxes (DFL expectation + dist2bbox) + sigmoid scores - one HIP launch over the
        per-level NHWC maps (`y3d_head2d_decode`)"""
        import ctypes
        if self.reg_max != 16:
            raise NotImplementedError("the DFL decode kernel is built for reg_max = 16")
        B, nl = ys[0].shape[0], len(ys)
        if not ys[0].is_cuda:
            raise Y3DError("Detect.inference runs on the HIP kernel of post.hip: head maps must live on a HIP device")
        ms = [y if (y.dtype == ys[0].dtype and ops.px_dense(y) and y.stride(3) == self.no) else ops._dense_any(y, ys[0].dtype) for y in ys]
        A = sum(y.shape[2] * y.shape[3] for y in ms)
        out = torch.empty(B, 4 + self.nc, A, dtype=torch.float32, device=ms[0].device)
        ops.lib().head2d_decode(ops.code(ms[0].dtype), nl, (ctypes.c_void_p * nl)(*[y.data_ptr() for y in ms]),
                                (ctypes.c_int * nl)(*[y.shape[2] for y in ms]), (ctypes.c_int * nl)(*[y.shape[3] for y in ms]),
                                (ctypes.c_float * nl)(*[float(s) for s in self.stride.tolist()[:nl]]), B, self.nc, out.data_ptr(), ops.stream())
        return out, ys

    def _probe(self, x, decode):
        """shape probe (see _host): level maps (B, no, H, W); with `decode` also the (B, 4 + nc, A) inference tensor"""
        maps = [_meta(xi.shape[0], self.no, *xi.shape[2:]) for xi in x]
        if not decode:
            return maps
        return torch.empty(x[0].shape[0], 4 + self.nc, sum(m.shape[2] * m.shape[3] for m in maps), device="meta"), maps

    def forward(self, x):
        if _host(x):
            return self._probe(x, not self.training)
        y = self.forward_feat(x, self.cv2, self.cv3)
        return y if self.training else self.inference(y)

    def bias_init(self):
        for a, b, s in zip(self.cv2, self.cv3, self.stride):
            a[-1].bias.data[:] = 1.0
            b[-1].bias.data[: self.nc] = math.log(5 / self.nc / (640 / float(s)) ** 2)


class v10Detect(Detect):
    """reference head.py:505-543"""

    max_det = 300

    def __init__(self, nc=80, ch=()):
        super().__init__(nc, ch)
        c3 = max(ch[0], min(self.nc, 100))
        self.cv3 = nn.ModuleList(nn.Sequential(nn.Sequential(Conv(x, x, 3, g=x), Conv(x, c3, 1)),
                                               nn.Sequential(Conv(c3, c3, 3, g=c3), Conv(c3, c3, 1)),
                                               nn.Conv2d(c3, self.nc, 1)) for x in ch)
        self.one2one_cv2 = copy.deepcopy(self.cv2)
        self.one2one_cv3 = copy.deepcopy(self.cv3)

    def forward(self, x):
        if _host(x):
            return {"one2many": self._probe(x, not self.training), "one2one": self._probe(x, not self.training)}
        one2one = self.forward_feat([xi.detach() for xi in x], self.one2one_cv2, self.one2one_cv3)
        one2many = self.forward_feat(x, self.cv2, self.cv3)
        if self.training:
            return {"one2many": one2many, "one2one": one2one}
        return {"one2many": self.inference(one2many), "one2one": self.inference(one2one)}

    def bias_init(self):
        super().bias_init()
        for a, b, s in zip(self.one2one_cv2, self.one2one_cv3, self.stride):
            a[-1].bias.data[:] = 1.0
            b[-1].bias.data[: self.nc] = math.log(5 / self.nc / (640 / float(s)) ** 2)
