"""TEST INFRASTRUCTURE — CPU restatement (oracle) of the YOLOv10 / YOLOv10-3D hot path.

Plain PyTorch fp32, functional, independent of the product package.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module;
the product path (`yolov10-3d_amd/`) never does.

Parity status: PINNED — every function below is checked in `tests/test_oracle_golden.py`
against golden vectors minted from the reference itself (`oracle/make_golden.py`, which
imports /root/reference through `oracle/ref_shim.py` in the build container) and, when
/root/reference is present, directly against the reference in `tests/test_oracle_vs_reference.py`.

Each function cites the reference file:line it restates (paths relative to
/root/reference/ultralytics/).  The model is described by a *spec* (list of layer dicts
derived from the yaml table) and a flat `state` dict of tensors keyed exactly like the
reference's `state_dict()`, so reference weights drop in unchanged.
"""
from __future__ import annotations

import copy
import math
import re
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

BN_EPS = 1e-3  # utils/torch_utils.py:327-337 (initialize_weights)
BN_MOM = 0.03

HEAD3D_BRANCHES = ("cls", "o2d", "s2d", "o3d", "s3d", "hd", "dep", "dep_un")  # head.py:563-572


def head3d_out_channels(nc: int) -> List[int]:
    return [nc, 2, 2, 2, 3, 24, 1, 1]


# --------------------------------------------------------------------------------------
# yaml table -> spec  (nn/tasks.py:837-964 parse_model)
# --------------------------------------------------------------------------------------
def make_divisible(x, divisor):
    return math.ceil(x / divisor) * divisor


def guess_scale(name: str) -> str:
    m = re.search(r"yolov\d+([nsblmx])", name)  # nn/tasks.py:1003
    return m.group(1) if m else ""


def build_spec(cfg: dict, ch: int = 3) -> dict:
    """Resolve the `[from, repeats, module, args]` rows into concrete layer descriptions."""
    d = copy.deepcopy(cfg)
    nc = d["nc"]
    scales = d.get("scales")
    depth, width, max_ch = 1.0, 1.0, float("inf")
    if scales:
        scale = d.get("scale") or tuple(scales.keys())[0]
        depth, width, max_ch = scales[scale]
    chs = [ch]
    layers, save = [], []
    for i, (f, n, m, args) in enumerate(d["backbone"] + d["head"]):
        args = [nc if a == "nc" else (None if a == "None" else a) for a in args]
        n = max(round(n * depth), 1) if n > 1 else n
        L = {"i": i, "f": f, "type": m}
        if m in ("Conv", "C2f", "C2fCIB", "SCDown", "SPPF", "PSA"):
            c1, c2 = chs[f], args[0]
            if c2 != nc:
                c2 = make_divisible(min(c2, max_ch) * width, 8)
            rest = list(args[1:])
            if m == "Conv":
                k = rest[0] if len(rest) > 0 else 1
                s = rest[1] if len(rest) > 1 else 1
                L.update(c1=c1, c2=c2, k=k, s=s)
            elif m == "C2f":
                L.update(c1=c1, c2=c2, n=n, shortcut=bool(rest[0]) if rest else False)
            elif m == "C2fCIB":
                L.update(c1=c1, c2=c2, n=n, shortcut=bool(rest[0]) if rest else False,
                         lk=bool(rest[1]) if len(rest) > 1 else False)
            elif m == "SCDown":
                L.update(c1=c1, c2=c2, k=rest[0], s=rest[1])
            elif m == "SPPF":
                L.update(c1=c1, c2=c2, k=rest[0] if rest else 5)
            elif m == "PSA":
                assert c1 == c2  # block.py:803
                L.update(c1=c1, c2=c2)
        elif m == "nn.Upsample":
            c2 = chs[f]
            L.update(scale=args[1], mode=args[2])
        elif m == "Concat":
            c2 = sum(chs[x] for x in f)
        elif m in ("v10Detect", "v10Detect3d"):
            c2 = None
            L.update(nc=nc, ch=[chs[x] for x in f])
            if m == "v10Detect3d":
                k1 = d.get("kernel_size_1") or 3  # default fix, SURVEY §0.5
                k2 = d.get("kernel_size_2") or 3
                L.update(channels=d["channels"], nl=d.get("num_scales", 3), k1=k1, k2=k2, dsconv=bool(d.get("dsconv")),
                         half=bool(d.get("half_channels")), pred=bool(d.get("use_predecessors")))
                for flag in ("deform", "common_head", "fgdm_predictor"):  # common_head: the reference's own forward fails (head3d_dense)
                    if d.get(flag):
                        raise NotImplementedError(f"{flag}=True is not used by any shipped yaml")
        else:
            raise NotImplementedError(m)
        layers.append(L)
        save.extend(x % i for x in ([f] if isinstance(f, int) else f) if x != -1)
        if i == 0:
            chs = []
        chs.append(c2)
    return {"layers": layers, "save": sorted(set(save)), "nc": nc}


# --------------------------------------------------------------------------------------
# parameter construction (shapes + state_dict key layout only; values are seeded random)
# --------------------------------------------------------------------------------------
def _conv_keys(state, p, c1, c2, k, g=1, gen=None):
    kh, kw = (k, k) if isinstance(k, int) else k
    fan_in = (c1 // g) * kh * kw
    bound = 1.0 / math.sqrt(fan_in)
    state[p + ".conv.weight"] = (torch.rand(c2, c1 // g, kh, kw, generator=gen) * 2 - 1) * bound
    state[p + ".bn.weight"] = torch.ones(c2)
    state[p + ".bn.bias"] = torch.zeros(c2)
    state[p + ".bn.running_mean"] = torch.zeros(c2)
    state[p + ".bn.running_var"] = torch.ones(c2)
    state[p + ".bn.num_batches_tracked"] = torch.zeros((), dtype=torch.long)


def _plain_conv_keys(state, p, c1, c2, gen=None):
    bound = 1.0 / math.sqrt(c1)
    state[p + ".weight"] = (torch.rand(c2, c1, 1, 1, generator=gen) * 2 - 1) * bound
    state[p + ".bias"] = (torch.rand(c2, generator=gen) * 2 - 1) * bound


def init_state(spec: dict, seed: int = 0, randomize_bn: bool = True) -> Dict[str, torch.Tensor]:
    """Seeded random parameters with the reference's key layout.  With `randomize_bn` the BN
    affine parameters / running stats are perturbed so parity tests exercise them."""
    gen = torch.Generator().manual_seed(seed)
    st: Dict[str, torch.Tensor] = {}
    for L in spec["layers"]:
        p = f"model.{L['i']}"
        t = L["type"]
        if t == "Conv":
            _conv_keys(st, p, L["c1"], L["c2"], L["k"], gen=gen)
        elif t in ("C2f", "C2fCIB"):
            c = int(L["c2"] * 0.5)
            _conv_keys(st, p + ".cv1", L["c1"], 2 * c, 1, gen=gen)
            _conv_keys(st, p + ".cv2", (2 + L["n"]) * c, L["c2"], 1, gen=gen)
            for j in range(L["n"]):
                q = f"{p}.m.{j}"
                if t == "C2f":
                    _conv_keys(st, q + ".cv1", c, c, 3, gen=gen)
                    _conv_keys(st, q + ".cv2", c, c, 3, gen=gen)
                else:  # CIB block.py:745-752 (e=1.0 -> c_ = c)
                    _conv_keys(st, q + ".cv1.0", c, c, 3, g=c, gen=gen)
                    _conv_keys(st, q + ".cv1.1", c, 2 * c, 1, gen=gen)
                    if L["lk"]:
                        _conv_keys(st, q + ".cv1.2.conv", 2 * c, 2 * c, 7, g=2 * c, gen=gen)
                        _conv_keys(st, q + ".cv1.2.conv1", 2 * c, 2 * c, 3, g=2 * c, gen=gen)
                    else:
                        _conv_keys(st, q + ".cv1.2", 2 * c, 2 * c, 3, g=2 * c, gen=gen)
                    _conv_keys(st, q + ".cv1.3", 2 * c, c, 1, gen=gen)
                    _conv_keys(st, q + ".cv1.4", c, c, 3, g=c, gen=gen)
        elif t == "SCDown":
            _conv_keys(st, p + ".cv1", L["c1"], L["c2"], 1, gen=gen)
            _conv_keys(st, p + ".cv2", L["c2"], L["c2"], L["k"], g=L["c2"], gen=gen)
        elif t == "SPPF":
            c_ = L["c1"] // 2
            _conv_keys(st, p + ".cv1", L["c1"], c_, 1, gen=gen)
            _conv_keys(st, p + ".cv2", c_ * 4, L["c2"], 1, gen=gen)
        elif t == "PSA":
            c = int(L["c1"] * 0.5)
            nh = c // 64
            hd = c // nh
            kd = int(hd * 0.5)
            _conv_keys(st, p + ".cv1", L["c1"], 2 * c, 1, gen=gen)
            _conv_keys(st, p + ".cv2", 2 * c, L["c1"], 1, gen=gen)
            _conv_keys(st, p + ".attn.qkv", c, c + 2 * nh * kd, 1, gen=gen)
            _conv_keys(st, p + ".attn.proj", c, c, 1, gen=gen)
            _conv_keys(st, p + ".attn.pe", c, c, 3, g=c, gen=gen)
            _conv_keys(st, p + ".ffn.0", c, 2 * c, 1, gen=gen)
            _conv_keys(st, p + ".ffn.1", 2 * c, c, 1, gen=gen)
        elif t == "v10Detect3d":
            outs = head3d_out_channels(L["nc"])
            for hs in ("o2o_heads", "o2m_heads"):
                for j, name in enumerate(HEAD3D_BRANCHES):
                    mid = L["channels"][name + "_c"]
                    last = mid // 2 if L.get("half") else mid  # head.py:629-636
                    extra = sum(outs[HEAD3D_BRANCHES.index(q_)] for q_ in HEAD3D_PREDECESSORS[name]) if L.get("pred") else 0  # head.py:597-605
                    for i in range(L["nl"]):
                        q = f"{p}.{hs}.{j}.{i}"
                        cin = L["ch"][i] + extra
                        if L.get("dsconv"):  # head.py:645-650
                            _conv_keys(st, q + ".0.0", cin, cin, L["k1"], g=cin, gen=gen)
                            _conv_keys(st, q + ".0.1", cin, mid, 1, gen=gen)
                            _conv_keys(st, q + ".1.0", mid, mid, L["k2"], g=mid, gen=gen)
                            _conv_keys(st, q + ".1.1", mid, last, 1, gen=gen)
                        else:
                            _conv_keys(st, q + ".0", cin, mid, L["k1"], gen=gen)
                            _conv_keys(st, q + ".1", mid, last, L["k2"], gen=gen)
                        _plain_conv_keys(st, q + ".2", last, outs[j], gen=gen)
        elif t == "v10Detect":
            nc, ch = L["nc"], L["ch"]
            c2 = max(16, ch[0] // 4, 64)
            c3 = max(ch[0], min(nc, 100))
            for i, x in enumerate(ch):
                for pre in ("", "one2one_"):
                    q = f"{p}.{pre}cv2.{i}"
                    _conv_keys(st, q + ".0", x, c2, 3, gen=gen)
                    _conv_keys(st, q + ".1", c2, c2, 3, gen=gen)
                    _plain_conv_keys(st, q + ".2", c2, 64, gen=gen)
                    q = f"{p}.{pre}cv3.{i}"
                    _conv_keys(st, q + ".0.0", x, x, 3, g=x, gen=gen)
                    _conv_keys(st, q + ".0.1", x, c3, 1, gen=gen)
                    _conv_keys(st, q + ".1.0", c3, c3, 3, g=c3, gen=gen)
                    _conv_keys(st, q + ".1.1", c3, c3, 1, gen=gen)
                    _plain_conv_keys(st, q + ".2", c3, nc, gen=gen)
            st[p + ".dfl.conv.weight"] = torch.arange(16, dtype=torch.float).view(1, 16, 1, 1)
    if randomize_bn:
        for k in list(st.keys()):
            if k.endswith(".bn.weight"):
                st[k] = 1.0 + 0.2 * (torch.rand(st[k].shape, generator=gen) - 0.5)
            elif k.endswith(".bn.bias"):
                st[k] = 0.2 * (torch.rand(st[k].shape, generator=gen) - 0.5)
            elif k.endswith(".bn.running_mean"):
                st[k] = 0.1 * (torch.rand(st[k].shape, generator=gen) - 0.5)
            elif k.endswith(".bn.running_var"):
                st[k] = 1.0 + 0.5 * torch.rand(st[k].shape, generator=gen)
    return st


# --------------------------------------------------------------------------------------
# layer forwards
# --------------------------------------------------------------------------------------
class Ctx:
    """Execution context: the flat parameter dict and the train/eval switch."""

    def __init__(self, state: Dict[str, torch.Tensor], training: bool, fp8_conv: bool = False):
        self.st = state
        self.training = training
        # restate the fp8 MFMA convolution mode (BASELINE configs[4]; yolov10-3d_amd/csrc/conv3x3_fp8.hip, no reference counterpart): the
        # forward of every 3x3 stride-1 Conv the kernel serves sees its input MX block-quantised (mx_quantize_act); the state is
        # expected to hold the fp8-valued weights already (fp8w_state); gradients pass straight through (_Fp8ConvSTE)
        self.fp8_conv = fp8_conv


def fp8_conv_served(x, w, s, pad, g):
    """the geometries y3d_conv3x3_fp8_ok takes: 3x3 stride 1 'same', input channels per group a multiple of 64 and >= 128, output
    channels per group a multiple of 16"""
    cin, cout = x.shape[1], w.shape[0]
    return (w.shape[-1] == 3 and s == 1 and pad == 1 and (cin // g) % 64 == 0 and cin // g >= 128 and (cout // g) % 16 == 0
            and x.shape[3] >= 8 and x.shape[2] >= 4)


class _Fp8ConvSTE(torch.autograd.Function):
    """forward: conv(mx_quantize_act(x), w); backward: the gradients of conv(x, w) - the data gradient sees w, the weight gradient the
    UNquantised x (the HIP path keeps both on its bf16 kernels: straight-through)"""

    @staticmethod
    def forward(ctx, x, w, g):
        ctx.save_for_backward(x, w)
        ctx.g = g
        return F.conv2d(mx_quantize_act(x)[2].to(x.dtype), w, None, 1, 1, 1, g)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = torch.nn.grad.conv2d_input(x.shape, w, dy, 1, 1, 1, ctx.g) if ctx.needs_input_grad[0] else None
        dw = torch.nn.grad.conv2d_weight(x, w.shape, dy, 1, 1, 1, ctx.g) if ctx.needs_input_grad[1] else None
        return dx, dw, None


def conv_bn_act(ctx: Ctx, p: str, x, k=1, s=1, g=1, act=True, pad=None):
    """nn/modules/conv.py:103-122 Conv.forward = act(bn(conv(x))); autopad :28; BN eps/mom torch_utils.py:327."""
    st = ctx.st
    w = st[p + ".conv.weight"]
    if pad is None:
        pad = w.shape[-1] // 2
    if getattr(ctx, "fp8_conv", False) and fp8_conv_served(x, w, s, pad, g):
        y = _Fp8ConvSTE.apply(x, w, g)
    else:
        y = F.conv2d(x, w, None, s, pad, 1, g)
    y = F.batch_norm(y, st[p + ".bn.running_mean"], st[p + ".bn.running_var"], st[p + ".bn.weight"],
                     st[p + ".bn.bias"], ctx.training, BN_MOM, BN_EPS)
    if ctx.training:
        st[p + ".bn.num_batches_tracked"] += 1
    return F.silu(y) if act else y


def plain_conv(ctx: Ctx, p: str, x):
    return F.conv2d(x, ctx.st[p + ".weight"], ctx.st[p + ".bias"])


def bottleneck(ctx, p, x, shortcut):
    """block.py:327-342 (k=(3,3), e=1.0 inside C2f)."""
    y = conv_bn_act(ctx, p + ".cv2", conv_bn_act(ctx, p + ".cv1", x, 3), 3)
    return x + y if shortcut else y


def repvggdw(ctx, p, x):
    """block.py:702-711: SiLU(dw7x7+BN  +  dw3x3+BN)."""
    c = x.shape[1]
    return F.silu(conv_bn_act(ctx, p + ".conv", x, 7, 1, c, act=False) + conv_bn_act(ctx, p + ".conv1", x, 3, 1, c, act=False))


def cib(ctx, p, x, shortcut, lk):
    """block.py:737-758."""
    c = x.shape[1]
    y = conv_bn_act(ctx, p + ".cv1.0", x, 3, 1, c)
    y = conv_bn_act(ctx, p + ".cv1.1", y, 1)
    c2 = y.shape[1]
    y = repvggdw(ctx, p + ".cv1.2", y) if lk else conv_bn_act(ctx, p + ".cv1.2", y, 3, 1, c2)
    y = conv_bn_act(ctx, p + ".cv1.3", y, 1)
    y = conv_bn_act(ctx, p + ".cv1.4", y, 3, 1, y.shape[1])
    return x + y if shortcut else y


def c2f(ctx, p, x, n, shortcut, cib_lk=None):
    """block.py:216-233 C2f.forward / :760-768 C2fCIB."""
    y = list(conv_bn_act(ctx, p + ".cv1", x, 1).chunk(2, 1))
    for j in range(n):
        q = f"{p}.m.{j}"
        y.append(bottleneck(ctx, q, y[-1], shortcut) if cib_lk is None else cib(ctx, q, y[-1], shortcut, cib_lk))
    return conv_bn_act(ctx, p + ".cv2", torch.cat(y, 1), 1)


def scdown(ctx, p, x, k, s):
    """block.py:820-827."""
    y = conv_bn_act(ctx, p + ".cv1", x, 1)
    return conv_bn_act(ctx, p + ".cv2", y, k, s, y.shape[1], act=False)


def sppf(ctx, p, x, k=5):
    """block.py:158-178."""
    x = conv_bn_act(ctx, p + ".cv1", x, 1)
    y1 = F.max_pool2d(x, k, 1, k // 2)
    y2 = F.max_pool2d(y1, k, 1, k // 2)
    y3 = F.max_pool2d(y2, k, 1, k // 2)
    return conv_bn_act(ctx, p + ".cv2", torch.cat((x, y1, y2, y3), 1), 1)


def psa_attention(ctx, p, x):
    """block.py:771-797 Attention.forward."""
    B, C, H, W = x.shape
    N = H * W
    nh = C // 64
    hd = C // nh
    kd = int(hd * 0.5)
    qkv = conv_bn_act(ctx, p + ".qkv", x, 1, act=False).view(B, nh, 2 * kd + hd, N)
    q, k, v = qkv.split([kd, kd, hd], dim=2)
    attn = (q.transpose(-2, -1) @ k) * (kd ** -0.5)
    attn = attn.softmax(dim=-1)
    o = (v @ attn.transpose(-2, -1)).view(B, C, H, W)
    o = o + conv_bn_act(ctx, p + ".pe", v.reshape(B, C, H, W), 3, 1, C, act=False)
    return conv_bn_act(ctx, p + ".proj", o, 1, act=False)


def psa(ctx, p, x):
    """block.py:799-818."""
    c = x.shape[1] // 2
    a, b = conv_bn_act(ctx, p + ".cv1", x, 1).split((c, c), dim=1)
    b = b + psa_attention(ctx, p + ".attn", b)
    b = b + conv_bn_act(ctx, p + ".ffn.1", conv_bn_act(ctx, p + ".ffn.0", b, 1), 1, act=False)
    return conv_bn_act(ctx, p + ".cv2", torch.cat((a, b), 1), 1)


# ---- v10Detect3d -----------------------------------------------------------------------
HEAD3D_PREDECESSORS = {"cls": (), "o2d": (), "s2d": (), "o3d": ("cls",), "s3d": ("cls",), "hd": ("cls",), "dep": ("cls", "s3d"),
                       "dep_un": ("cls", "s3d", "dep")}  # head.py:585-594
HEAD3D_DEP_NORM = 65.0  # head.py:595


def head3d_conv(ctx, q, x, k, dsconv, pad=None):
    """head.py:645-650 build_conv: Conv(k), or with `dsconv` a depth-wise Conv(k) followed by a 1x1 Conv.  `pad` = 0 is the patch path's
    override, which reaches a plain Conv only (head.py:706-708 looks at the branch's top-level layers)."""
    if not dsconv:
        return conv_bn_act(ctx, q, x, k, pad=pad)
    y = conv_bn_act(ctx, q + ".0", x, k, g=x.shape[1])
    return conv_bn_act(ctx, q + ".1", y, 1)


def head3d_branch(ctx, q, x, k1, k2, pad=None, want_emb=False, dsconv=False):
    """one branch of build_head (head.py:629-636); `half_channels` only changes the weight shapes"""
    e = head3d_conv(ctx, q + ".0", x, k1, dsconv, pad)
    y = head3d_conv(ctx, q + ".1", e, k2, dsconv, pad)
    y = plain_conv(ctx, q + ".2", y)
    return (y, e) if want_emb else y


def head3d_dense(ctx, p, hs, xs, L):
    """head.py:718-743 forward_feat: per level cat of 8 branches; 'dep' embeddings :745-749.  L['pred'] (`use_predecessors`): a
    branch's input is the level's map concatenated with the detached outputs of HEAD3D_PREDECESSORS, depth / 65 (:727-737);
    L['dsconv'] as head3d_conv.  `common_head` is not restated: the reference's own forward_feat fails on it (:746 asserts three
    layers per branch, build_small_head makes two)."""
    ys, embs = [], []
    ds, pred = bool(L.get("dsconv")), bool(L.get("pred"))
    for i in range(L["nl"]):
        outs = {}
        emb = None
        for j, name in enumerate(HEAD3D_BRANCHES):
            q = f"{p}.{hs}.{j}.{i}"
            xin = xs[i]
            if pred and HEAD3D_PREDECESSORS[name]:
                xin = torch.cat([xs[i]] + [(outs[k] / HEAD3D_DEP_NORM if k == "dep" else outs[k]).detach() for k in HEAD3D_PREDECESSORS[name]], 1)
            if name == "dep":
                o, emb = head3d_branch(ctx, q, xin, L["k1"], L["k2"], want_emb=True, dsconv=ds)
            else:
                o = head3d_branch(ctx, q, xin, L["k1"], L["k2"], dsconv=ds)
            outs[name] = o
        ys.append(torch.cat(list(outs.values()), 1))
        embs.append(emb)
    return ys, embs


def head3d_select_candidates(cls_map, max_det):
    """head.py:686-692: per image top-`max_det` cells of the max-class logit -> (row, col)."""
    B, _, H, W = cls_map.shape
    m = cls_map.max(dim=1)[0].reshape(B, -1)
    idx = torch.topk(m, max_det, dim=1, largest=True)[1]
    return torch.stack((idx // W, idx % W), -1)  # (B, max_det, 2) = (row, col)


def head3d_sparse(ctx, p, hs, xs, L, max_det=50):
    """head.py:694-716 inference_forward_feat: dense cls, the other 7 branches only on the
    (k1+k2-1)^2 input patches around the top-50 cells, both convs with padding 0 (patch semantics,
    SURVEY §0.5), results scattered into zero maps."""
    ps = (L["k1"] - 1) + (L["k2"] - 1) + 1
    pad = ps // 2
    outs_ch = head3d_out_channels(L["nc"])
    ys = []
    ds = bool(L.get("dsconv"))
    if L.get("pred"):
        raise RuntimeError("use_predecessors: the reference's patch path feeds the branches bare feature patches (head.py:709) - channel mismatch")
    for i in range(L["nl"]):
        x = xs[i]
        B, C, H, W = x.shape
        cls = head3d_branch(ctx, f"{p}.{hs}.0.{i}", x, L["k1"], L["k2"], dsconv=ds)
        cand = head3d_select_candidates(cls, max_det)  # (B,K,2)
        xp = F.pad(x, (pad, pad, pad, pad))
        rows = cand[..., 0].reshape(-1)
        cols = cand[..., 1].reshape(-1)
        bidx = torch.arange(B).repeat_interleave(max_det)
        dr = torch.arange(ps)
        patches = xp[bidx[:, None, None], :, (rows[:, None] + dr)[:, :, None], (cols[:, None] + dr)[:, None, :]]
        patches = patches.permute(0, 3, 1, 2).contiguous()  # (B*K, C, ps, ps)
        outs = [cls]
        for j in range(1, 8):
            # patch cell (0, 0): the centre's value when both convs run unpadded (5x5 -> 1x1); with `dsconv` the convs keep their padding
            # and the 5x5 result is read at its corner, as the reference does (head3d_conv)
            o = head3d_branch(ctx, f"{p}.{hs}.{j}.{i}", patches, L["k1"], L["k2"], pad=0, dsconv=ds)[:, :, 0, 0]  # (B*K, c)
            full = torch.zeros(B, outs_ch[j], H, W)
            # later candidates overwrite earlier ones only if duplicated (top-k indices are unique)
            full[bidx, :, rows, cols] = o
            outs.append(full)
        ys.append(torch.cat(outs, 1))
    return ys


def make_anchors(shapes: Sequence[Tuple[int, int]], strides: Sequence[float], offset=0.5):
    """utils/tal.py:300-312."""
    pts, st = [], []
    for (h, w), s in zip(shapes, strides):
        sx = torch.arange(w, dtype=torch.float32) + offset
        sy = torch.arange(h, dtype=torch.float32) + offset
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        pts.append(torch.stack((xx, yy), -1).view(-1, 2))
        st.append(torch.full((h * w, 1), float(s)))
    return torch.cat(pts), torch.cat(st)


def head3d_decode(ys, strides, nc):
    """head.py:755-797 inference()+decode(): (B,38,A) with xyxy px boxes and centre-3d px."""
    B = ys[0].shape[0]
    no = ys[0].shape[1]
    cat = torch.cat([y.view(B, no, -1) for y in ys], 2)
    anc, st = make_anchors([y.shape[2:] for y in ys], strides)
    anc, st = anc.t(), st.t()
    cls, o2d, s2d, o3d, s3d, hd, dep, dep_un = cat.split((nc, 2, 2, 2, 3, 24, 1, 1), 1)
    s2 = s2d * st
    c2 = (o2d + anc) * st
    bbox = torch.cat((c2 - s2 / 2, c2 + s2 / 2), 1)
    c3 = (o3d + anc) * st
    return torch.cat((cls, bbox, c3, s3d, hd, dep, dep_un), 1)


def head3d(ctx, p, xs, L, strides):
    """head.py:814-833 v10Detect3d.forward."""
    xs = xs[: L["nl"]]
    if ctx.training:
        o2o, o2o_e = head3d_dense(ctx, p, "o2o_heads", [x.detach() for x in xs], L)
        o2m, o2m_e = head3d_dense(ctx, p, "o2m_heads", xs, L)
        return {"one2many": o2m, "one2one": o2o, "o2m_embs": o2m_e, "o2o_embs": o2o_e}
    maps = head3d_sparse(ctx, p, "o2o_heads", [x.detach() for x in xs], L)
    return {"one2one": (head3d_decode(maps, strides, L["nc"]), maps), "o2o_embs": None}


# ---- v10Detect (2D) ----------------------------------------------------------------------
def head2d_feat(ctx, p, pre, xs):
    """head.py:85-89 forward_feat with v10Detect's cv3 (:511-515)."""
    ys = []
    for i, x in enumerate(xs):
        q = f"{p}.{pre}cv2.{i}"
        box = plain_conv(ctx, q + ".2", conv_bn_act(ctx, q + ".1", conv_bn_act(ctx, q + ".0", x, 3), 3))
        q = f"{p}.{pre}cv3.{i}"
        c = conv_bn_act(ctx, q + ".0.1", conv_bn_act(ctx, q + ".0.0", x, 3, 1, x.shape[1]), 1)
        c = conv_bn_act(ctx, q + ".1.1", conv_bn_act(ctx, q + ".1.0", c, 3, 1, c.shape[1]), 1)
        ys.append(torch.cat((box, plain_conv(ctx, q + ".2", c)), 1))
    return ys


def dfl_expect(box, reg_max=16):
    """block.py:59-62 DFL.forward: softmax over 16 bins, expectation."""
    b, _, a = box.shape
    return (box.view(b, 4, reg_max, a).softmax(2) * torch.arange(reg_max, dtype=box.dtype).view(1, 1, -1, 1)).sum(2)


def head2d_decode(ys, strides, nc):
    """head.py:53-79 Detect.inference: xywh boxes px + sigmoid scores -> (B, 4+nc, A)."""
    B, no = ys[0].shape[:2]
    cat = torch.cat([y.view(B, no, -1) for y in ys], 2)
    anc, st = make_anchors([y.shape[2:] for y in ys], strides)
    anc, st = anc.t(), st.t()
    box, cls = cat.split((64, nc), 1)
    d = dfl_expect(box)
    lt, rb = d.split([2, 2], 1)
    x1y1, x2y2 = anc.unsqueeze(0) - lt, anc.unsqueeze(0) + rb
    dbox = torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), 1) * st
    return torch.cat((dbox, cls.sigmoid()), 1)


def head2d(ctx, p, xs, L, strides):
    """head.py:519-533 v10Detect.forward."""
    o2o = head2d_feat(ctx, p, "one2one_", [x.detach() for x in xs])
    o2m = head2d_feat(ctx, p, "", xs)
    if ctx.training:
        return {"one2many": o2m, "one2one": o2o}
    return {"one2many": (head2d_decode(o2m, strides, L["nc"]), o2m), "one2one": (head2d_decode(o2o, strides, L["nc"]), o2o)}


# --------------------------------------------------------------------------------------
# whole-model forward (nn/tasks.py:115-135 _predict_once)
# --------------------------------------------------------------------------------------
def model_strides(spec):
    """Strides implied by the table (the reference probes them with a 256x256 forward, nn/tasks.py:301-310)."""
    down = {}
    cur = 1
    for L in spec["layers"]:
        f = L["f"]
        src = down[f if f >= 0 else L["i"] + f] if isinstance(f, int) and L["i"] > 0 else (1 if isinstance(f, int) else None)
        t = L["type"]
        if t == "Conv":
            cur = src * L["s"]
        elif t == "SCDown":
            cur = src * L["s"]
        elif t == "nn.Upsample":
            cur = src // L["scale"]
        elif t == "Concat":
            cur = down[f[0] if f[0] >= 0 else L["i"] + f[0]]
        elif t in ("v10Detect", "v10Detect3d"):
            nl = L.get("nl", len(f))
            return [float(down[x]) for x in f][:nl]
        else:
            cur = src
        down[L["i"]] = cur
    raise ValueError("no detect layer")


def forward(spec, state, img, training: bool, fp8_conv: bool = False):
    ctx = Ctx(state, training, fp8_conv)
    strides = model_strides(spec)
    saved = {}
    x = img
    for L in spec["layers"]:
        i, f, t = L["i"], L["f"], L["type"]
        p = f"model.{i}"
        if f != -1:
            x = saved[f] if isinstance(f, int) else [x if j == -1 else saved[j] for j in f]
        if t == "Conv":
            x = conv_bn_act(ctx, p, x, L["k"], L["s"])
        elif t == "C2f":
            x = c2f(ctx, p, x, L["n"], L["shortcut"])
        elif t == "C2fCIB":
            x = c2f(ctx, p, x, L["n"], L["shortcut"], cib_lk=L["lk"])
        elif t == "SCDown":
            x = scdown(ctx, p, x, L["k"], L["s"])
        elif t == "SPPF":
            x = sppf(ctx, p, x, L["k"])
        elif t == "PSA":
            x = psa(ctx, p, x)
        elif t == "nn.Upsample":
            x = F.interpolate(x, scale_factor=float(L["scale"]), mode=L["mode"])
        elif t == "Concat":
            x = torch.cat(x, 1)
        elif t == "v10Detect3d":
            x = head3d(ctx, p, x, L, strides)
        elif t == "v10Detect":
            x = head2d(ctx, p, x, L, strides)
        if i in spec["save"]:
            saved[i] = x
    return x


# --------------------------------------------------------------------------------------
# geometry helpers
# --------------------------------------------------------------------------------------
def ciou(b1, b2, eps=1e-7):
    """utils/metrics.py:78-134 bbox_iou(xywh=False, CIoU=True); boxes (...,4) broadcastable; returns (...)."""
    x11, y11, x12, y12 = b1.unbind(-1)
    x21, y21, x22, y22 = b2.unbind(-1)
    w1, h1 = x12 - x11, y12 - y11 + eps
    w2, h2 = x22 - x21, y22 - y21 + eps
    inter = (torch.minimum(x12, x22) - torch.maximum(x11, x21)).clamp(min=0) * \
            (torch.minimum(y12, y22) - torch.maximum(y11, y21)).clamp(min=0)
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    cw = torch.maximum(x12, x22) - torch.minimum(x11, x21)
    chh = torch.maximum(y12, y22) - torch.minimum(y11, y21)
    c2 = cw.pow(2) + chh.pow(2) + eps
    rho2 = ((x21 + x22 - x11 - x12).pow(2) + (y21 + y22 - y11 - y12).pow(2)) / 4
    v = (4 / math.pi ** 2) * ((w2 / h2).atan() - (w1 / h1).atan()).pow(2)
    with torch.no_grad():
        alpha = v / (v - iou + (1 + eps))
    return iou - (rho2 / c2 + v * alpha)


def keypoints_3d(center, dep, size3d, hbin, hres, calib):
    """utils/keypoint_utils.py:11-118 get_3d_keypoints.
    center (B,N,2) px, dep (B,N,1), size3d (B,N,3)=(h,w,l), hbin (B,N,12 logits | 1 index), hres (B,N,12 | 1), calib (B,6).
    Returns (B,N,8,3)."""
    cu, cv, fu, fv, tx, ty = [calib[:, None, k:k + 1] for k in range(6)]  # (B,1,1)
    X = (center[..., 0:1] - cu) * dep / fu + tx
    Y = (center[..., 1:2] - cv) * dep / fv + ty
    loc = torch.cat((X, Y, dep), -1)  # :113-118
    hl, hw, hh = size3d[..., 2:3] / 2, size3d[..., 1:2] / 2, size3d[..., 0:1] / 2
    cx = torch.cat((hl, hl, -hl, -hl, hl, hl, -hl, -hl), -1)
    cy = torch.cat((hw, -hw, hw, -hw, hw, -hw, hw, -hw), -1)
    cz = torch.cat((-hh, -hh, -hh, -hh, hh, hh, hh, hh), -1)
    corners = torch.stack((cx, cy, cz), -1)  # (B,N,8,3)  :20-26
    if hbin.shape[-1] > 1:
        bi = hbin.argmax(-1)
    else:
        bi = hbin[..., 0].long()
    res = hres.gather(-1, bi.unsqueeze(-1))[..., 0] if hres.shape[-1] > 1 else hres[..., 0]
    ang = bi.to(res.dtype) * (2 * math.pi / 12.0) + res  # :42-48
    ang = torch.where(ang > math.pi, ang - 2 * math.pi, ang)
    ry = ang.unsqueeze(-1) + torch.arctan2(center[..., 0:1] - cu, fu)  # :94-101
    ry = torch.where(ry > math.pi, ry - 2 * math.pi, ry)
    ry = torch.where(ry < -math.pi, ry + 2 * math.pi, ry)
    # R = Rx(pi/2) @ Ry(-ry) @ Rz(0)  (:87-91), applied as  out_i = sum_j R[j,i] * p_j  (:104-110)
    a = -ry[..., 0]
    ca, sa = torch.cos(a), torch.sin(a)
    cxr, sxr = math.cos(math.pi / 2), math.sin(math.pi / 2)
    one, zero = torch.ones_like(ca), torch.zeros_like(ca)
    Rx = torch.tensor([[1.0, 0.0, 0.0], [0.0, cxr, -sxr], [0.0, sxr, cxr]], dtype=ca.dtype)
    Ry = torch.stack((ca, zero, sa, zero, one, zero, -sa, zero, ca), -1).reshape(ca.shape + (3, 3))
    R = torch.matmul(Rx, Ry)
    out = torch.einsum("bnji,bnkj->bnki", R, corners) + loc.unsqueeze(-2)
    return out


# --------------------------------------------------------------------------------------
# task-aligned assigner (2D: utils/tal.py:19-264 ; 3D: utils/tal.py:355-753)
# --------------------------------------------------------------------------------------
def stable_topk_mask(metric, k, valid_gt):
    """Membership mask of the k largest entries per row with LOWEST-INDEX-FIRST tie rule.

    Restates tal.py:615-649 select_topk_candidates (topk -> masked_fill(~mask_gt, 0) -> scatter_add
    -> count>1 := 0).  The reference's tie order among exactly-equal metrics is whatever
    libstdc++ partial_sort/nth_element (CPU) or radix-select (CUDA) yields and differs between
    its own devices; we pin it to a stable order ("ties")."""
    B, n, A = metric.shape
    order = torch.sort(metric, dim=-1, descending=True, stable=True)[1][..., :k]  # (B,n,k)
    order = torch.where(valid_gt.expand(-1, -1, k).bool(), order, torch.zeros_like(order))
    cnt = torch.zeros(B, n, A, dtype=torch.int32)
    cnt.scatter_add_(-1, order, torch.ones_like(order, dtype=torch.int32))
    cnt = torch.where(cnt > 1, torch.zeros_like(cnt), cnt)
    return cnt.to(metric.dtype)


def _resolve(mask_pos, overlaps):
    """tal.py:728-753 select_highest_overlaps."""
    n = mask_pos.shape[1]
    fg = mask_pos.sum(-2)
    if fg.max() > 1:
        multi = (fg.unsqueeze(1) > 1).expand(-1, n, -1)
        best = overlaps.argmax(1)
        onehot = torch.zeros_like(mask_pos)
        onehot.scatter_(1, best.unsqueeze(1), 1)
        mask_pos = torch.where(multi, onehot, mask_pos).float()
        fg = mask_pos.sum(-2)
    return mask_pos.argmax(-2), fg, mask_pos


def _in_gts(anc, gt_bboxes, eps=1e-9):
    """tal.py:709-726 select_candidates_in_gts."""
    lt, rb = gt_bboxes[..., None, :2], gt_bboxes[..., None, 2:]
    d = torch.cat((anc[None, None] - lt, rb - anc[None, None]), -1)
    return (d.amin(-1) > eps).to(gt_bboxes.dtype)


def tal2d(pd_scores, pd_bboxes, anc, gt_labels, gt_bboxes, mask_gt, topk, nc, alpha=0.5, beta=6.0, eps=1e-9):
    """utils/tal.py:45-94 TaskAlignedAssigner.forward.  Returns (labels, bboxes, scores, fg_mask bool, gt_idx int64)."""
    B, A = pd_scores.shape[:2]
    n = gt_bboxes.shape[1]
    if n == 0:
        return (torch.full((B, A), float(nc)), torch.zeros_like(pd_bboxes), torch.zeros_like(pd_scores),
                torch.zeros(B, A, dtype=torch.bool), torch.zeros(B, A, dtype=torch.long))
    in_g = _in_gts(anc, gt_bboxes)
    m = (in_g * mask_gt).bool()
    lab = gt_labels.squeeze(-1).long()
    sc = pd_scores.gather(2, lab.clamp(min=0)[:, None, :].expand(-1, A, -1)).permute(0, 2, 1)  # (B,n,A)
    sc = torch.where(m, sc, torch.zeros_like(sc))
    ov = ciou(gt_bboxes[:, :, None, :], pd_bboxes[:, None, :, :]).clamp(min=0)
    ov = torch.where(m, ov, torch.zeros_like(ov))
    align = sc.pow(alpha) * ov.pow(beta)
    mask_pos = stable_topk_mask(align, topk, mask_gt) * in_g * mask_gt
    gt_idx, fg, mask_pos = _resolve(mask_pos, ov)
    flat = gt_idx + torch.arange(B)[:, None] * n
    t_lab = lab.flatten()[flat].clamp(min=0)
    t_box = gt_bboxes.reshape(-1, 4)[flat]
    t_sc = F.one_hot(t_lab, nc).to(pd_scores.dtype) * (fg > 0).unsqueeze(-1)
    align = align * mask_pos
    pa = align.amax(-1, keepdim=True)
    po = (ov * mask_pos).amax(-1, keepdim=True)
    norm = (align * po / (pa + eps)).amax(-2).unsqueeze(-1)
    return t_lab, t_box, t_sc * norm, fg.bool(), gt_idx


def tal3d(pd_scores, pd_bboxes, pd_3d, anc, gts, mask_gt, stride_tensor, calibs, mean_sizes, topk, nc,
          alpha=0.5, beta=1.0, gamma=1.0, eps=1e-9, use_2d=True, use_3d=True, kps_dist="l1", constrain=True):
    """utils/tal.py:392-452 TaskAlignedAssigner3d.forward.  Defaults = cfg/default.yaml:112-119 (use_2d = use_3d = True, kps 'l1',
    constrain_anchors); the other modes follow tal.py:465-497: box-only metric (get_box_metrics, "overlaps" = CIoU), keypoint-only
    (get_keypoint_metrics, "overlaps" = similarities), 'l2' keypoint distance 1 / exp(0.5 * sum(d^2) / 24), and candidates not
    restricted to the anchors inside the box (`constrain_anchors: False`: the metric mask and mask_pos use mask_gt alone).
    `gts` = (labels(B,n,1), bboxes xyxy px(B,n,4), center_2d, size_2d, center_3d, size_3d residual, depth, heading_bin, heading_res).
    Returns (targets[9], fg_mask bool (B,A), target_gt_idx int64 (B,A))."""
    if not (use_2d or use_3d):
        raise RuntimeError("Either 2D or 3D assignment or both has to be selected!")  # tal.py:484
    gl, gb, gc2, gs2, gc3, gs3, gd, ghb, ghr = gts
    B, A = pd_scores.shape[:2]
    n = gb.shape[1]
    o3d, s3d, hd, dep, _ = pd_3d.split((2, 3, 24, 1, 1), -1)
    pc3 = anc + o3d * stride_tensor  # :454-456
    ps3 = mean_sizes[pd_scores.argmax(-1)] + s3d  # :458-462
    lab = gl.squeeze(-1).long()
    gs3_full = mean_sizes[lab.clamp(min=0)] + gs3  # :605-609
    g_kps = keypoints_3d(gc3, gd, gs3_full, ghb, ghr, calibs)  # (B,n,8,3)
    p_kps = keypoints_3d(pc3, dep, ps3, hd[..., :12], hd[..., 12:], calibs)  # (B,A,8,3)
    in_g = _in_gts(anc, gb)
    gmask = (in_g * mask_gt) if constrain else mask_gt.expand(-1, -1, A)  # :476, 480, 483
    m = gmask.bool()
    sc = pd_scores.gather(2, lab.clamp(min=0)[:, None, :].expand(-1, A, -1)).permute(0, 2, 1)
    sc = torch.where(m, sc, torch.zeros_like(sc))
    diff = p_kps[:, None] - g_kps[:, :, None]
    if kps_dist == "l1":
        sim_all = 1 / torch.exp(diff.abs().sum((-1, -2)) / 24)  # :465-467
    else:
        sim_all = 1 / torch.exp(0.5 * (diff * diff).sum((-1, -2)) / 24)  # :468-470
    sim = torch.where(m, sim_all, torch.zeros_like(sim_all))
    ov = torch.where(m, ciou(gb[:, :, None, :], pd_bboxes[:, None, :, :]).clamp(min=0), torch.zeros_like(sim_all))
    if use_2d and use_3d:
        align, second = sc.pow(alpha) * ov.pow(beta) * sim.pow(gamma), sim  # :602-603 (returns `similarities` as "overlaps")
    elif use_3d:
        align, second = sc.pow(alpha) * sim.pow(gamma), sim  # :575-576
    else:
        align, second = sc.pow(alpha) * ov.pow(beta), ov  # :553-554
    mask_pos = stable_topk_mask(align, topk, mask_gt) * gmask  # :488-497
    sim = second
    gt_idx, fg, mask_pos = _resolve(mask_pos, sim)
    flat = gt_idx + torch.arange(B)[:, None] * n
    t_lab = lab.flatten()[flat].clamp(min=0)

    def take(t):
        return t.reshape(-1, t.shape[-1])[flat]

    t_sc = F.one_hot(t_lab, nc).to(pd_scores.dtype) * (fg > 0).unsqueeze(-1)
    align = align * mask_pos
    pa = align.amax(-1, keepdim=True)
    po = (sim * mask_pos).amax(-1, keepdim=True)
    norm = (align * po / (pa + eps)).amax(-2).unsqueeze(-1)
    targets = [t_lab, t_sc * norm, take(gc2), take(gs2), take(gc3), take(gs3), take(gd), take(ghb), take(ghr)]
    return targets, fg.bool(), gt_idx


# --------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------
HYP = dict(box=5.0, cls=1.0, dfl=1.5, loss2d=2.0, depth=1.0, offset3d=10.0, size3d=1.0, heading=1.0,
           tal_topk=8, tal_alpha=0.5, tal_beta=1.0, tal_gamma=1.0)  # cfg/default.yaml:102-115,141


def pad_targets(rows, B, width, scale):
    """loss.py:795-810 preprocess: (n, 1+width) rows [batch_idx | ...] -> (B, n_max, width), boxes (cols 1:5) -> xyxy px."""
    if rows.shape[0] == 0:
        return torch.zeros(B, 0, width)
    bi = rows[:, 0]
    counts = torch.stack([(bi == j).sum() for j in range(B)])
    out = torch.zeros(B, int(counts.max()), width)
    for j in range(B):
        sel = rows[bi == j, 1:]
        out[j, : sel.shape[0]] = sel
    xywh = out[..., 1:5] * scale
    xy, wh = xywh[..., :2], xywh[..., 2:]
    out[..., 1:5] = torch.cat((xy - wh / 2, xy + wh / 2), -1)
    return out


def loss3d_one(feats, batch, strides, nc, topk, hyp=HYP):
    """utils/loss.py:821-900 DDDetectionLoss.__call__ (distillation off).  feats: list of (B,38,H,W).
    Returns (loss.sum()*B, 6-vector, aux dict with assignment)."""
    B = feats[0].shape[0]
    no = feats[0].shape[1]
    cat = torch.cat([f.view(B, no, -1) for f in feats], 2).permute(0, 2, 1)  # (B,A,38)
    sc, o2d, s2d, o3d, s3d, hd, dep, dun = cat.split((nc, 2, 2, 2, 3, 24, 1, 1), -1)
    pred_2d = torch.cat((o2d, s2d), -1)
    pred_3d = torch.cat((o3d, s3d, hd, dep, dun), -1)
    H, W = feats[0].shape[2:]
    imgsz = torch.tensor([H, W], dtype=torch.float32) * strides[0]
    anc, st = make_anchors([f.shape[2:] for f in feats], strides)
    rows = torch.cat((batch["batch_idx"].view(-1, 1), batch["cls"].view(-1, 1), batch["bboxes"], batch["center_2d"],
                      batch["size_2d"], batch["center_3d"], batch["size_3d"], batch["depth"].view(-1, 1),
                      batch["heading_bin"].view(-1, 1), batch["heading_res"].view(-1, 1)), 1).float()
    g = pad_targets(rows, B, 17, imgsz[[1, 0, 1, 0]])
    gts = g.split((1, 4, 2, 2, 2, 3, 1, 1, 1), 2)
    mask_gt = (gts[1].sum(2, keepdim=True) > 0).float()
    centers = anc + pred_2d[..., :2]
    pb = torch.cat((centers - pred_2d[..., 2:] / 2, centers + pred_2d[..., 2:] / 2), -1) * st  # :812-819
    loss = torch.zeros(6)
    if g.shape[1] == 0:  # reference returns graph-less zeros here
        return loss.sum() * B, loss, {}
    targets, fg, gt_idx = tal3d(sc.detach().sigmoid(), pb.detach(), pred_3d.detach(), anc * st, gts, mask_gt, st,
                                batch["calib"].float(), batch["mean_sizes"].float(), topk, nc,
                                hyp["tal_alpha"], hyp["tal_beta"], hyp["tal_gamma"])
    _, t_sc, t_c2, t_s2, t_c3, t_s3, t_d, t_hb, t_hr = targets
    tss = max(t_sc.sum(), 1)
    anc_px = anc * st
    # 2D box: L1 "mean" over fg elements, then / target_scores_sum   (:913-926)
    p2 = (pred_2d * st)[fg]
    off_l = F.l1_loss(p2[..., :2], (t_c2 - anc_px)[fg], reduction="mean")
    siz_l = F.l1_loss(p2[..., 2:], t_s2[fg], reduction="mean")
    loss[0] = (siz_l + off_l) / tss * hyp["loss2d"]
    loss[1] = F.binary_cross_entropy_with_logits(sc, t_sc, reduction="none").sum() / tss * hyp["cls"]
    # 3D (:928-963)
    p3 = pred_3d[fg]
    pd, pu = p3[..., -2], p3[..., -1]
    td = t_d[fg].squeeze(-1)
    loss[2] = (1.4142 * torch.exp(-0.5 * pu) * (pd - td).abs() + 0.5 * pu).sum() / tss * hyp["depth"]  # :1112-1119
    loss[3] = F.l1_loss((pred_3d[..., :2] * st)[fg], (t_c3 - anc_px)[fg], reduction="mean") / tss * hyp["offset3d"]
    loss[4] = F.l1_loss(p3[..., 2:5], t_s3[fg], reduction="sum") / tss * hyp["size3d"]
    ph = p3[..., 5:29]
    tb = t_hb[fg].view(-1).long()
    ce = F.cross_entropy(ph[..., :12], tb, reduction="sum")  # :1122-1136
    reg = F.l1_loss(ph[..., 12:].gather(1, tb.view(-1, 1)).squeeze(1), t_hr[fg].view(-1), reduction="sum")
    loss[5] = (ce + reg) / tss * hyp["heading"]
    return loss.sum() * B, loss, {"fg_mask": fg, "target_gt_idx": gt_idx, "target_scores": t_sc}


def loss3d(preds, batch, strides, nc, hyp=HYP):
    """utils/loss.py:740-771 DetectLoss3d.__call__: o2o (topk=1) + o2m (topk=tal_topk); items = cat(o2m, o2o)."""
    l1, i1, a1 = loss3d_one(preds["one2one"], batch, strides, nc, 1, hyp)
    lm, im, am = loss3d_one(preds["one2many"], batch, strides, nc, hyp["tal_topk"], hyp)
    return lm + l1, torch.cat((im, i1)), {"one2many": am, "one2one": a1}


def loss2d_one(feats, batch, strides, nc, topk, hyp=HYP):
    """utils/loss.py:206-257 v8DetectionLoss.__call__ (+BboxLoss :82-113)."""
    B, no = feats[0].shape[:2]
    cat = torch.cat([f.view(B, no, -1) for f in feats], 2).permute(0, 2, 1)
    dist, sc = cat.split((64, nc), -1)
    H, W = feats[0].shape[2:]
    imgsz = torch.tensor([H, W], dtype=torch.float32) * strides[0]
    anc, st = make_anchors([f.shape[2:] for f in feats], strides)
    rows = torch.cat((batch["batch_idx"].view(-1, 1), batch["cls"].view(-1, 1), batch["bboxes"]), 1).float()
    g = pad_targets(rows, B, 5, imgsz[[1, 0, 1, 0]])
    gl, gb = g.split((1, 4), 2)
    mask_gt = (gb.sum(2, keepdim=True) > 0).float()
    A = dist.shape[1]
    d = dist.view(B, A, 4, 16).softmax(3).matmul(torch.arange(16, dtype=torch.float32))  # :197-204
    pb = torch.cat((anc - d[..., :2], anc + d[..., 2:]), -1)
    _, t_box, t_sc, fg, gt_idx = tal2d(sc.detach().sigmoid(), pb.detach() * st, anc * st, gl, gb, mask_gt, topk, nc)
    tss = max(t_sc.sum(), 1)
    loss = torch.zeros(3)
    loss[1] = F.binary_cross_entropy_with_logits(sc, t_sc, reduction="none").sum() / tss
    if fg.sum():
        t_box = t_box / st
        w = t_sc.sum(-1)[fg].unsqueeze(-1)
        iou = ciou(pb[fg], t_box[fg]).unsqueeze(-1)
        loss[0] = ((1.0 - iou) * w).sum() / tss
        ltrb = torch.cat((anc - t_box[..., :2], t_box[..., 2:] - anc), -1).clamp(0, 15 - 0.01)[fg]  # bbox2dist :328-331
        pdist = dist[fg].view(-1, 16)
        tl = ltrb.long()
        tr = tl + 1
        wl = tr - ltrb
        wr = 1 - wl
        dfl = (F.cross_entropy(pdist, tl.view(-1), reduction="none").view(tl.shape) * wl +
               F.cross_entropy(pdist, tr.view(-1), reduction="none").view(tl.shape) * wr).mean(-1, keepdim=True)
        loss[2] = (dfl * w).sum() / tss
    loss = loss * torch.tensor([hyp["box"], hyp["cls"], hyp["dfl"]])
    return loss.sum() * B, loss.detach(), {"fg_mask": fg, "target_gt_idx": gt_idx}


def loss2d(preds, batch, strides, nc, hyp=HYP):
    """utils/loss.py:727-737 v10DetectLoss: o2m topk=10, o2o topk=1."""
    lm, im, am = loss2d_one(preds["one2many"], batch, strides, nc, 10, hyp)
    l1, i1, a1 = loss2d_one(preds["one2one"], batch, strides, nc, 1, hyp)
    return lm + l1, torch.cat((im, i1)), {"one2many": am, "one2one": a1}


# --------------------------------------------------------------------------------------
# postprocess + BN folding
# --------------------------------------------------------------------------------------
def postprocess3d(preds, max_det=50, nc=3):
    """utils/ops.py:867-880."""
    scores, reg = preds.split([nc, preds.shape[-1] - nc], -1)
    ms, idx = torch.topk(scores.amax(-1), max_det, dim=-1)
    reg = reg.gather(1, idx.unsqueeze(-1).expand(-1, -1, reg.shape[-1]))
    scores = scores.gather(1, idx.unsqueeze(-1).expand(-1, -1, nc))
    s2, idx2 = torch.topk(scores.flatten(1), max_det, dim=-1)
    return reg.gather(1, (idx2 // nc).unsqueeze(-1).expand(-1, -1, reg.shape[-1])), s2, idx2 % nc


def postprocess2d(preds, max_det=300, nc=80):
    """utils/ops.py:852-865."""
    boxes, scores = preds.split([4, nc], -1)
    ms, idx = torch.topk(scores.amax(-1), max_det, dim=-1)
    boxes = boxes.gather(1, idx.unsqueeze(-1).expand(-1, -1, 4))
    scores = scores.gather(1, idx.unsqueeze(-1).expand(-1, -1, nc))
    s2, idx2 = torch.topk(scores.flatten(1), max_det, dim=-1)
    return boxes.gather(1, (idx2 // nc).unsqueeze(-1).expand(-1, -1, 4)), s2, idx2 % nc


def fold_conv_bn(w, gamma, beta, mean, var, eps=BN_EPS):
    """utils/torch_utils.py:171-198 fuse_conv_and_bn: W' = diag(g/sqrt(v+eps)) W ; b' = beta - g*mean/sqrt(v+eps)."""
    s = gamma / torch.sqrt(var + eps)
    return w * s.view(-1, 1, 1, 1), beta - mean * s


def fold_repvggdw(w7, b7, w3, b3):
    """block.py:716-735 RepVGGDW.fuse: pad the folded 3x3 into the folded 7x7."""
    return w7 + F.pad(w3, [2, 2, 2, 2]), b7 + b3


# --------------------------------------------------------------------------------------
# KITTI decode of the post-processed predictions  (data/datasets/kitti.py:519-576 decode_preds; helpers:
# data/datasets/decode_helper.py:12-18 bin2angle, data/datasets/kitti_utils.py:241-251 img_to_rect, :286-299
# camera_dis_to_rect, :311-325 alpha2ry, :467-470 affine_transform)
# --------------------------------------------------------------------------------------
KITTI_MEAN_SIZE = ((1.52563191462, 1.62856739989, 3.88311640418), (1.76255119, 0.66068622, 0.84422524),
                   (1.73698127, 0.59706367, 1.76282397))  # kitti.py:38-41, (h, w, l) per class


def kitti_decode(preds, calib, ratio, inv_trans, mean_size=KITTI_MEAN_SIZE, threshold=0.001, use_camera_dis=False):
    """preds (B, K, 37) fp32 rows [xyxy 4 | center3d 2 | size3d 3 | heading 24 | depth | depth log-variance | score logit | label]
    (`v10_3Dpostprocess` output, validator layout); calib (B, 6) = (cu, cv, fu, fv, tx, ty) of the ORIGINAL image (the reference
    reads them from its float32 P2 matrix); ratio (B, 2) = ratio_pad[i][0]; inv_trans (B, 2, 3) or None (undo_augment=False).
    -> rows (B, K, 14) float64 [cls, alpha, x1, y1, x2, y2, h, w, l, x, y, z, ry, score], keep (B, K) bool (score >= threshold).
    The reference computes these per detection in numpy with float32 inputs promoted to float64 by the calibration constants;
    the dtype of every intermediate below follows that code."""
    import numpy as np
    P = preds.detach().cpu().float().numpy()
    B, K, _ = P.shape
    calib = np.asarray(calib, dtype=np.float64).reshape(B, 6)
    ratio = np.asarray(ratio, dtype=np.float64).reshape(B, 2)
    ms = np.asarray(mean_size, dtype=np.float64)
    out = np.zeros((B, K, 14), dtype=np.float64)
    keep = np.zeros((B, K), dtype=bool)
    bins = P[..., 9:21].argmax(-1)                                            # first maximum, as torch.argmax
    res = np.take_along_axis(P[..., 21:33], bins[..., None], -1)[..., 0]      # float32
    ang = (bins.astype(np.float32) * np.float32(2 * math.pi / 12.0)) + res    # torch: int64 * python float -> float32
    ang = np.where(ang > np.float32(math.pi), ang - np.float32(2 * math.pi), ang).astype(np.float32)
    sig = 1.0 / (1.0 + np.exp(-P[..., 35].astype(np.float32)))               # torch.sigmoid in float32
    for i in range(B):
        cu, cv, fu, fv, tx, ty = calib[i]
        for j in range(K):
            r = P[i, j]
            cid = int(r[36])
            box = r[0:4].astype(np.float64) / ratio[i][[0, 1, 0, 1]]
            x = (box[0] + box[2]) / 2
            dim = (r[6:9] + ms[cid]).astype(np.float32)                       # in-place add on the float32 view
            depth = np.float64(r[33])
            sigma = float(np.exp(np.float32(-r[34])))                         # torch.exp in float32, then .item()
            if inv_trans is not None:
                t = np.asarray(inv_trans[i], dtype=np.float64).reshape(2, 3)
                c3 = t @ np.array([r[4], r[5], 1.0], dtype=np.float32).astype(np.float64)
                u, v = c3[0], c3[1]
            else:
                u = np.float64(np.float32(r[4] * np.float32(1242)) / np.float32(1280.0))
                v = np.float64(np.float32(r[5] * np.float32(375)) / np.float32(384.0))
            if use_camera_dis:
                fd = np.sqrt((u - cu) ** 2 + (v - cv) ** 2 + fu ** 2)
                lx = ((u - cu) * depth) / fd + tx
                ly = ((v - cv) * depth) / fd + ty
                lz = np.sqrt(depth ** 2 - lx ** 2 - ly ** 2)
            else:
                lx = ((u - cu) * depth) / fu + tx
                ly = ((v - cv) * depth) / fv + ty
                lz = depth
            ly = ly + np.float64(dim[0]) / 2
            alpha = float(ang[i, j])
            ry = alpha + np.arctan2(x - cu, fu)
            if ry > np.pi:
                ry -= 2 * np.pi
            if ry < -np.pi:
                ry += 2 * np.pi
            score = float(sig[i, j]) * sigma
            out[i, j] = [cid, alpha, *box, *dim.astype(np.float64), lx, ly, lz, ry, score]
            keep[i, j] = not (score < threshold)
    return out, keep


# ---------------------------------------------------------------------------------------------------------
# f2: one-to-many depth fusion of the validator (models/yolov10_3D/val.py:78-102)
# ---------------------------------------------------------------------------------------------------------
def box_iou_xyxy(b1, b2, eps=1e-7):
    """utils/metrics.py box_iou: (N, 4) x (M, 4) -> (N, M), float32 in the reference's operation order"""
    (a1, a2), (c1, c2) = b1.unsqueeze(1).chunk(2, 2), b2.unsqueeze(0).chunk(2, 2)
    inter = (torch.min(a2, c2) - torch.max(a1, c1)).clamp_(0).prod(2)
    return inter / ((a2 - a1).prod(2) + (c2 - c1).prod(2) - inter + eps)


def kde_fuse_depth(predsO, predsM, thres=0.1, iou_thres=0.9, nprop=500):
    """val.py:78-102 `aggregate_o2m_preds`: predsO (B, K, 37), predsM (B, KM, 37) post-processed rows [xyxy | ... | depth(-4) |
    depth log-variance(-3) | score(-2) | label(-1)].  For every one-to-one detection: the one-to-many detections with IoU > 0.9 whose
    label matches and whose depth score exp(-log-variance) exceeds `thres` vote with it on the depth: a gaussian kernel density
    (sklearn KernelDensity(bandwidth="silverman"): h = (n (d + 2) / 4)^(-1 / (d + 4)) with d = 1 - NOT scaled by the spread of the
    samples), weights = normalised depth scores, evaluated at `nprop` evenly spaced proposals between the smallest and the largest
    vote (np.linspace on float32 end points: float32 arithmetic); the most likely proposal (first maximum of the log-density)
    replaces the depth.  -> fused predsO (copy)"""
    import numpy as np
    out = predsO.clone()
    for i in range(predsO.shape[0]):
        iou = box_iou_xyxy(predsO[i, :, :4], predsM[i, :, :4])
        for j in range(predsO.shape[1]):
            m = iou[j] > iou_thres
            d = torch.cat((predsO[i, j, -4:-3], predsM[i, m, -4]))
            u = torch.cat((predsO[i, j, -3:-2], predsM[i, m, -3]))
            c = torch.cat((predsO[i, j, -1:], predsM[i, m, -1]))
            s = torch.exp(-u)
            mask = (s > thres) & (c == predsO[i, j, -1])
            if int(mask.sum()) > 1:
                s, d = s[mask], d[mask]
                w = (s / s.sum()).numpy().astype(np.float64)
                x = d.numpy().astype(np.float64)
                h = (len(x) * 3 / 4.0) ** (-1 / 5.0)
                lo, hi = np.float32(d.min()), np.float32(d.max())
                step = np.float32((hi - lo) / np.float32(nprop - 1))
                prop = (np.arange(nprop, dtype=np.float32) * step + lo).astype(np.float32)
                prop[-1] = hi
                pp = prop.astype(np.float64)
                dens = (w[None, :] * np.exp(-0.5 * ((pp[:, None] - x[None, :]) / h) ** 2)).sum(1)
                out[i, j, -4] = float(prop[int(np.argmax(np.log(dens)))])
    return out


# ---------------------------------------------------------------------------------------------------------
# f3: image side of KITTIDataset.__getitem__ (data/datasets/kitti.py:132-206): mirror, mixup blend, affine crop, /255
# ---------------------------------------------------------------------------------------------------------
def kitti_image_aug(img, img2, flip, trans_inv, out_wh):
    """img (H, W, 3) uint8 RGB, img2 the mixup partner or None, flip: mirror both left-right (kitti.py:149, 184), blend
    `Image.blend(img, img2, 0.5)` (:188; Pillow: in1 + alpha * (in2 - in1) in float, truncated to uint8), then
    `img.transform(resolution, AFFINE, trans_inv, BILINEAR)` (:192-196; Pillow Geometry.c: the output pixel centre (x + .5, y + .5)
    is mapped through the 2x3 matrix in double precision, a source position outside [0, W) x [0, H) gives 0, otherwise bilinear
    interpolation around (xin - .5, yin - .5) with clamped neighbours, result truncated to uint8).
    -> (outH, outW, 3) uint8; the reference's tensor is this / 255 as float32, CHW (:204-205)."""
    import numpy as np
    a = np.asarray(img).astype(np.float64)
    if flip:
        a = a[:, ::-1]
    if img2 is not None:
        b = np.asarray(img2).astype(np.float64)
        if flip:
            b = b[:, ::-1]
        a = np.floor(a + 0.5 * (b - a))
    h, w, _ = a.shape
    W, H = int(out_wh[0]), int(out_wh[1])
    t = np.asarray(trans_inv, np.float64).reshape(2, 3)
    ys, xs = np.mgrid[0:H, 0:W]
    xin = t[0, 0] * (xs + 0.5) + t[0, 1] * (ys + 0.5) + t[0, 2]
    yin = t[1, 0] * (xs + 0.5) + t[1, 1] * (ys + 0.5) + t[1, 2]
    inside = (xin >= 0) & (xin < w) & (yin >= 0) & (yin < h)
    xf, yf = xin - 0.5, yin - 0.5
    x, y = np.floor(xf).astype(np.int64), np.floor(yf).astype(np.int64)
    dx, dy = (xf - x)[..., None], (yf - y)[..., None]
    yc, x0, x1 = np.clip(y, 0, h - 1), np.clip(x, 0, w - 1), np.clip(x + 1, 0, w - 1)
    v1 = a[yc, x0] + (a[yc, x1] - a[yc, x0]) * dx
    y1 = np.clip(y + 1, 0, h - 1)
    v2 = a[y1, x0] + (a[y1, x1] - a[y1, x0]) * dx
    v2 = np.where(((y + 1 >= 0) & (y + 1 < h))[..., None], v2, v1)
    out = np.floor(v1 + (v2 - v1) * dy).astype(np.uint8)
    out[~inside] = 0
    return out


def get_affine_transform(center, scale, output_size, inv=False):
    """kitti_utils.py:423-464 with rot = 0, shift = 0 (the only form the dataset uses, kitti.py:192): the affine map through the
    three point pairs (crop centre, the point half a crop WIDTH above it, their right-angle companion) -> (output centre, ...).
    float32 points as the reference builds them, solved in float64 (cv2.getAffineTransform).  -> trans (2, 3) [, trans_inv]"""
    import numpy as np
    center = np.asarray(center, np.float64)
    scale = np.asarray(scale, np.float64) if np.ndim(scale) else np.array([scale, scale], np.float64)
    src_w, dst_w, dst_h = scale[0], output_size[0], output_size[1]
    src = np.zeros((3, 2), np.float32)
    dst = np.zeros((3, 2), np.float32)
    src[0] = center
    src[1] = center + np.array([0, src_w * -0.5])
    dst[0] = [dst_w * 0.5, dst_h * 0.5]
    dst[1] = np.array([dst_w * 0.5, dst_h * 0.5], np.float32) + np.array([0, dst_w * -0.5], np.float32)

    def third(a, b):
        d = a - b
        return b + np.array([-d[1], d[0]], np.float32)

    src[2], dst[2] = third(src[0], src[1]), third(dst[0], dst[1])

    def solve(p, q):
        A = np.hstack((p.astype(np.float64), np.ones((3, 1))))
        return np.linalg.solve(A, q.astype(np.float64)).T.copy()

    trans = solve(src, dst)
    return (trans, solve(dst, src)) if inv else trans


# ---------------------------------------------------------------------------------------------------------
# fp8 weights (BASELINE configs[4]; no reference counterpart: the build's own format, csrc/fp8w.hip)
# ---------------------------------------------------------------------------------------------------------
def fp8w_quantize(w):
    """w (Cout, ...) fp32 -> (codes uint8 same shape, scale (Cout,), w_eff fp32): OCP e4m3fn codes (torch.float8_e4m3fn: round to
    nearest even) of w / scale with scale = 2^ceil(log2(absmax / 448)) per output channel (1 for an all-zero row)."""
    rows = w.shape[0]
    flat = w.detach().float().reshape(rows, -1)
    amax = flat.abs().amax(1)
    scale = torch.where(amax > 0, torch.exp2(torch.ceil(torch.log2(amax.double() / 448.0))).float(), torch.ones_like(amax))
    q = (flat / scale[:, None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    w_eff = (q.float() * scale[:, None]).reshape(w.shape)
    return q.view(torch.uint8).reshape(w.shape), scale, w_eff


def fp8w_state(spec, state):
    """a copy of the oracle state with every dense / grouped conv weight that is followed by BatchNorm replaced by its fp8 value
    (the weights `y3d.set_weight_quant("fp8")` multiplies with); depth-wise filters and the heads' final projections untouched"""
    out = {}
    for k, v in state.items():
        if k.endswith(".conv.weight") and v.dim() == 4 and k[: -len("conv.weight")] + "bn.weight" in state and not (v.shape[1] == 1 and v.shape[0] > 1):
            out[k] = fp8w_quantize(v)[2]
        else:
            out[k] = v
    return out


def mx_quantize_act(x, block=32):
    """OCP-MX block quantisation of an activation tensor along its channels - the operand format of the fp8 MFMA convolution
    (yolov10-3d_amd/csrc/conv3x3_fp8.hip: y3d_fp8_quantize_act; no reference counterpart - the reference has no 8-bit path).
    x (B, C, H, W), C % 32 == 0 -> (codes uint8 (B, C, H, W), scale bytes uint8 (B, C / 32, H, W), x_eff fp32 = value(code) * 2^(byte - 127)).
    Per (pixel, 32-channel block): scale = the smallest power of two with amax / scale <= 448 (so nothing saturates; 1 for an all-zero
    block, exponent clamped at -127), code = torch.float8_e4m3fn(x / scale): round to nearest even."""
    B, C, H, W = x.shape
    assert C % block == 0
    xb = x.detach().float().reshape(B, C // block, block, H, W)
    amax = xb.abs().amax(2, keepdim=True)
    m, E = torch.frexp(amax)  # amax = m * 2^E, m in [0.5, 1)
    e = (E - 9 + (m > 0.875).to(E.dtype)).clamp(min=-127)
    e = torch.where(amax > 0, e, torch.zeros_like(e))
    inv = torch.exp2(-e.double()).float()
    q = (xb * inv).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    x_eff = (q.float().double() * torch.exp2(e.double())).float().reshape(B, C, H, W)
    return q.view(torch.uint8).reshape(B, C, H, W), (e + 127).to(torch.uint8).reshape(B, C // block, H, W), x_eff


def conv3x3_fp8(x, w, groups=1):
    """3x3 stride-1 'same' convolution as the fp8 MFMA path computes it: fp8w-quantised weights (per-output-channel power-of-two
    scales, fp8w_quantize) x MX block-quantised activations (mx_quantize_act), products exact, accumulation in high precision
    (float64 here; the kernel: fp32 in its own order).  -> y fp32"""
    w_eff = fp8w_quantize(w)[2]
    x_eff = mx_quantize_act(x)[2]
    return F.conv2d(x_eff.double(), w_eff.double(), None, 1, 1, 1, groups).float()
