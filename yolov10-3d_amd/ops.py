"""Autograd glue between torch tensors and the y3d C ABI (include/y3d.h).

PyTorch is plumbing here (device memory, streams, the autograd tape); every forward/backward body
below is a sequence of liby3d_hip.so launches on the current HIP stream.  Tensors keep the logical
NCHW shape of the reference's modules but live in NHWC ("channels last") memory in the compute
dtype (bf16 by default, fp32 for the parity mode).
"""
from __future__ import annotations

import os

import torch

from ._lib import BF16, F32, Y3DError, lib

_COMPUTE_DTYPE = torch.bfloat16


def set_compute_dtype(dt: torch.dtype):
    """torch.bfloat16 (performance mode) or torch.float32 (parity mode, exact-f32 MFMA)."""
    global _COMPUTE_DTYPE
    if dt not in (torch.bfloat16, torch.float32):
        raise ValueError("compute dtype must be torch.bfloat16 or torch.float32")
    _COMPUTE_DTYPE = dt


def compute_dtype() -> torch.dtype:
    return _COMPUTE_DTYPE


def code(dt: torch.dtype) -> int:
    if dt == torch.bfloat16:
        return BF16
    if dt == torch.float32:
        return F32
    raise Y3DError(f"unsupported tensor dtype {dt}")


def ce(dt: torch.dtype) -> int:
    return 8 if dt == torch.bfloat16 else 4


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _require_gpu(x: torch.Tensor):
    if not x.is_cuda:
        raise Y3DError("the y3d HIP path needs tensors on a HIP device (there is no CPU fallback)")


def nhwc_empty(B, C, H, W, dtype, device):
    """logical (B,C,H,W), memory (B,H,W,C)"""
    return torch.empty((B, H, W, C), dtype=dtype, device=device).permute(0, 3, 1, 2)


# ---- output placement: a producer writes its result straight into a channel slice of its consumer's concat buffer ----------------
# (C2f / SPPF / Upsample -> Concat: the reference's torch.cat, block.py:236 / :176, conv.py:404, becomes a no-op instead of one copy per
# input; every kernel on this path takes a pixel stride, so a channel slice of a wider NHWC buffer is an ordinary operand.)
PLACEMENT = True  # tests flip it for A/B comparisons against copying
EVAL_LEVEL_STREAMS = True  # captured eval forward: one hipGraph branch per detection level (modules.v10Detect3d.inference_forward_feat)
_LEVEL_STREAMS: list = []


def level_streams(n: int):
    while len(_LEVEL_STREAMS) < n:
        _LEVEL_STREAMS.append(torch.cuda.Stream())
    return _LEVEL_STREAMS[:n]
_PLACE = None


class place:
    """`with place(buf, off): y = producer(x)` - the next output tensor of matching batch / spatial shape and dtype allocated by a
    kernel wrapper inside the block is `buf[:, off:off + C]` instead of a fresh tensor (consumed by the first match)."""

    def __init__(self, buf, off):
        self.t = (buf, off) if buf is not None else None

    def __enter__(self):
        global _PLACE
        self.prev, _PLACE = _PLACE, self.t

    def __exit__(self, *a):
        global _PLACE
        _PLACE = self.prev


# cross-layer placement (round 3): a Concat row of the yaml also takes the output of an EARLIER row (the backbone / neck skips: layers 4,
# 6, 10, 13 of the v10 tables).  tasks._predict_once opens `place_final(buf, off)` around that earlier row; the row's module hands the
# slot to its LAST kernel with `final_place()` (C2f / C2fCIB / PSA: the closing 1x1 conv), so the skip is written where the concat will
# read it - 13 -> 8 member copies per forward, 157 -> 33 MB.  A module that never calls final_place() leaves the slot unused (the concat
# copies that member as before).
_PLACE_FINAL = None


class place_final:
    def __init__(self, buf, off):
        self.t = (buf, off) if buf is not None else None

    def __enter__(self):
        global _PLACE_FINAL
        self.prev, _PLACE_FINAL = _PLACE_FINAL, self.t

    def __exit__(self, *a):
        global _PLACE_FINAL
        _PLACE_FINAL = self.prev


def final_place():
    """`with ops.final_place(): return last_op(...)` inside a module: the pending cross-layer slot (if any) becomes the placement of that op"""
    global _PLACE_FINAL
    t, _PLACE_FINAL = _PLACE_FINAL, None
    return place(*(t or (None, 0)))


def concat_buffer(x, C, H=None, W=None):
    """the buffer a group of producers will fill for a channel concat of C channels at x's batch / spatial size, or None"""
    dtype = _COMPUTE_DTYPE
    if not PLACEMENT or not x.is_cuda or C % ce(dtype) != 0:
        return None
    buf = nhwc_empty(x.shape[0], C, x.shape[2] if H is None else H, x.shape[3] if W is None else W, dtype, x.device)
    _CONCAT_BASE[buf.data_ptr()] = (C, buf)
    return buf


# addresses of the live concat buffers: ConcatFn only trusts "already in place" for slices of a buffer made by concat_buffer (a PSA
# block concatenates a slice of its cv1 output with a NEW tensor: that slice has the right strides too, and filling "its" buffer would
# overwrite the other half, which the backward still needs).  An entry dies with its ConcatFn; a model forward starts from none.
# The entry HOLDS its buffer: while an address is in the table the memory behind it cannot be freed and handed to another tensor, so
# "address + channel count match" cannot be a coincidence (a buffer nobody consumed lives until the next reset_placement).
_CONCAT_BASE = {}


# gradient placement (the mirror image of `place` for the backward pass): SplitChannelsFn hands channel slices of one map to several
# consumers; a consumer whose backward ends in a data-gradient kernel writes that gradient straight into ITS slice of one shared
# gradient buffer, and SplitChannelsFn.backward returns the buffer instead of concatenating the pieces (0.8 ms per step on the M head).
# key: (address, channels) of a slice as the consumer receives it -> (holder dict, channel offset)
_GRAD_SLOT = {}


def grad_slot(x):
    """the slice of the shared gradient buffer that belongs to input `x` of a consumer (a view made by SplitChannelsFn), or None"""
    ent = _GRAD_SLOT.get((x.data_ptr(), x.shape[1]))
    if ent is None:
        return None
    holder, off = ent
    if holder["shape"] != (x.shape[0], x.shape[2], x.shape[3]) or holder["dtype"] != x.dtype:
        return None
    return holder, off, x.shape[1]


def grad_slot_tensor(slot):
    """materialise (once) the shared buffer of a slot and return this consumer's slice of it"""
    holder, off, w = slot
    if holder["buf"] is None:
        B, H, W = holder["shape"]
        holder["buf"] = nhwc_empty(B, holder["C"], H, W, holder["dtype"], holder["device"])
    return holder["buf"][:, off:off + w]


def reset_placement():
    global _PLACE, _PLACE_FINAL
    _PLACE = None
    _PLACE_FINAL = None
    _CONCAT_BASE.clear()
    _GRAD_SLOT.clear()
    _FP8_ACT.clear()


def out_tensor(B, C, H, W, dtype, device):
    """nhwc_empty, or the placement slice if one is pending and fits"""
    global _PLACE
    if _PLACE is not None:
        buf, off = _PLACE
        if (buf.dtype == dtype and buf.shape[0] == B and buf.shape[2] == H and buf.shape[3] == W and off + C <= buf.shape[1]
                and off % ce(dtype) == 0 and C % ce(dtype) == 0 and buf.device == device):
            _PLACE = None
            return buf[:, off:off + C]
    return nhwc_empty(B, C, H, W, dtype, device)


def is_nhwc(x: torch.Tensor) -> bool:
    return x.dim() == 4 and (x.stride(1) == 1 or x.shape[1] == 1)


def px_dense(x: torch.Tensor) -> bool:
    """element (b,h,w,c) at ((b*H + h)*W + w) * stride(3) + c  (size-1 dims may carry arbitrary strides)"""
    B, C, H, W = x.shape
    sw = x.stride(3)
    return is_nhwc(x) and sw >= C and (H == 1 or x.stride(2) == W * sw) and (B == 1 or x.stride(0) == H * W * sw)


def to_nhwc(x: torch.Tensor, dtype=None, dense=False) -> torch.Tensor:
    """Return x as an NHWC tensor of the compute dtype, 16-byte aligned; copies only when needed."""
    dtype = dtype or _COMPUTE_DTYPE
    c = ce(dtype)
    ok = (x.dtype == dtype and is_nhwc(x) and x.data_ptr() % 16 == 0 and x.stride(0) % c == 0 and x.stride(2) % c == 0
          and x.stride(3) % c == 0 and (not dense or px_dense(x)))
    if ok:
        return x
    B, C, H, W = x.shape
    out = nhwc_empty(B, C, H, W, dtype, x.device)
    out.copy_(x)
    return out


def s3(x):
    return x.stride(0), x.stride(2), x.stride(3)


def _f32(n, device):
    return torch.empty(n, dtype=torch.float32, device=device)


class KernelTimer:
    """Live per-launch timing of selected kernels with HIP events on the launch stream (bench.py's roofline leg).
    `want(key)` decides which launches are bracketed; results: {key: [ms, ...]}."""

    def __init__(self, want):
        self.want = want
        self.pending = []

    def bracket(self, key, fn):
        if not self.want(key):
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = fn()
        b.record()
        self.pending.append((key, a, b))
        return r

    def results(self):
        torch.cuda.synchronize()
        out = {}
        for key, a, b in self.pending:
            out.setdefault(key, []).append(a.elapsed_time(b))
        return out


TIMER = None  # set by bench.py

# Parameters, BatchNorm running statistics and EMA weights are written by HIP kernels through raw pointers, which torch's tensor
# version counters do not see.  Every such writer bumps this epoch, and the eval-path cache of packed / folded weights keys on it.
PARAM_EPOCH = 0


def bump_param_epoch():
    global PARAM_EPOCH
    PARAM_EPOCH += 1


WEIGHT_EPOCH = 0  # bumped by the raw-pointer writers of PARAMETERS only (optimizer step): the packed training weights key on it


def bump_weight_epoch():
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1
    bump_param_epoch()


class _PackRegistry:
    """Packed compute-dtype copies of the conv weights for the training step, refreshed by ONE multi-tensor launch per step
    (`y3d_mt_pack_weights`) instead of one tiny launch per conv and direction (S-3D: 113 launches of ~5 us + their gaps).

    A weight is identified by (address, geometry); its packed buffer is persistent.  The first lookup after the weights changed
    (WEIGHT_EPOCH moved: the fused optimizer wrote them) repacks every registered weight from whatever its address holds now;
    a weight changed through torch (in-place op: its `_version` moved) is repacked on its own; an unknown weight is packed on its
    own and registered.  Entries that nobody looked up for a few epochs are dropped (re-pointed / freed parameters)."""

    CHUNK = 16384
    KEEP = 4

    def __init__(self, mode):
        self.mode, self.entries, self.epoch, self.tables = mode, {}, None, None
        # A hipGraph capture (graph.GraphedTrainStep) bakes in the ADDRESSES of the descriptor tables and of the packed buffers it looked
        # up: both must outlive the graph.  Tables used while capturing are kept here for good (a later `_pack_all` with more entries -
        # a second model in the process - builds new tables and would otherwise free the ones the graph's launch still reads: it then
        # follows whatever "pointers" the reused memory holds), entries looked up while capturing are never evicted.
        self.captured_tables = []

    def _pack_one(self, e):
        L, st = lib(), stream()
        a, b, c, taps, kpad, dt = e["geo"]
        if self.mode == 0:
            k = int(round(taps ** 0.5))
            L.pack_weight_fwd(dt, e["src"], e["dst"].data_ptr(), a, b, c, k, k, st)
        else:
            k = int(round(taps ** 0.5))
            L.pack_weight_dgrad(dt, e["src"], e["dst"].data_ptr(), a * b, c, a, k, k, st)

    def _pack_all(self, dt):
        ents = [e for e in self.entries.values() if e["geo"][5] == dt]
        if not ents:
            return
        dev = ents[0]["dst"].device
        key = tuple(id(e) for e in ents)
        if self.tables is None or self.tables[0] != (key, dt):
            desc, ct, co = [], [], []
            for t, e in enumerate(ents):
                a, b, c, taps, kpad, _ = e["geo"]
                n = a * kpad if self.mode == 0 else a * c * kpad
                desc += [e["src"], e["dst"].data_ptr(), a, b, c, taps, kpad, self.mode]
                for j in range((n + self.CHUNK - 1) // self.CHUNK):
                    ct.append(t)
                    co.append(j)
            self.tables = ((key, dt), torch.tensor(desc, dtype=torch.int64, device=dev), torch.tensor(ct, dtype=torch.int32, device=dev),
                           torch.tensor(co, dtype=torch.int32, device=dev), len(ct))
        _, desc, ct, co, n = self.tables
        if torch.cuda.is_current_stream_capturing() and not any(t is self.tables for t in self.captured_tables):
            self.captured_tables.append(self.tables)
        lib().mt_pack_weights(dt, desc.data_ptr(), ct.data_ptr(), co.data_ptr(), n, self.CHUNK, stream())

    def lookup(self, w32, src_ptr, geo, dtype, ver=None):
        """geo = (a, b, c, taps, Kpad, dt) as y3d_mt_pack_weights; src_ptr: address of the (sub)tensor to pack -> packed tensor.
        `ver`: the torch-side version token of the weights (default: w32's own counter).  A stacked weight passes the counters of
        its per-branch Parameters: they are re-pointed at slices of the flat tensor with `.data =`, which shares storage but NOT the
        version counter, so torch-side writers (torch.optim, load_state_dict, init, broadcast) move only the Parameters' counters."""
        key = (src_ptr, geo)
        ver = w32._version if ver is None else ver
        e = self.entries.get(key)
        now = (WEIGHT_EPOCH, geo[5])
        if e is None:
            a, b, c, taps, kpad, dt = geo
            n = a * kpad if self.mode == 0 else a * c * kpad
            # "keep": the registry re-reads `src` at every epoch, so the storage behind it must outlive the entry
            e = {"src": src_ptr, "geo": geo, "dst": torch.empty(n, dtype=dtype, device=w32.device), "ver": ver, "seen": WEIGHT_EPOCH,
                 "keep": w32}
            self.entries[key] = e
            self._pack_one(e)
            return e["dst"]
        e["seen"] = WEIGHT_EPOCH
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            e["pinned"] = True
        if self.epoch != now:
            self.epoch = now
            # (not while a hipGraph is being captured: dropping entries changes the table key, and rebuilding the descriptor tables is a
            # host-to-device copy, which a capture does not allow - entries of a model that is gone wait for the next eager step)
            stale = [] if capturing else [k for k, v in self.entries.items() if WEIGHT_EPOCH - v["seen"] > self.KEEP and not v.get("pinned")]
            for k in stale:
                del self.entries[k]
            self._pack_all(geo[5])
            for v in self.entries.values():
                if v["geo"][5] == geo[5]:
                    v["ver"] = None  # refreshed from memory: whatever version the tensor has now is the packed one
        if e["ver"] is None:
            e["ver"] = ver
        elif e["ver"] != ver:  # changed through torch since it was packed (load_state_dict, init, broadcast, torch.optim)
            e["ver"] = ver
            self._pack_one(e)
        return e["dst"]


class _QuantRegistry:
    """fp8 (e4m3fn, per-output-channel power-of-two scale) shadows of the conv weights — the `fp8w` weight mode (csrc/fp8w.hip).

    For every registered fp32 master weight (identified by address + shape) it keeps `w_eff` (fp32, what the kernels pack and
    multiply: exactly representable in bf16), the 1-byte codes and the scales.  All shadows are refreshed by ONE multi-tensor
    launch the first time a weight is looked up after the fused optimizer moved the weight epoch; a weight changed through torch
    (its version token moved) is refreshed on its own.  `lookup` returns (w_eff, token): the token stands in for the master's
    version in the pack registries' and the eval caches' keys (the shadow's own counter never moves)."""

    def __init__(self):
        self.entries, self.epoch, self.tables = {}, None, None
        self.captured_tables = []  # as _PackRegistry: tables a hipGraph capture has baked in stay alive

    def _run(self, ents):
        dev = ents[0]["weff"].device
        key = tuple(id(e) for e in ents)
        if self.tables is None or self.tables[0] != key:
            desc, rb, r = [], [], 0
            for e in ents:
                rows, K = e["shape"]
                desc += [e["src"], e["weff"].data_ptr(), e["codes"].data_ptr(), e["scale"].data_ptr(), rows, K]
                rb.append(r)
                r += rows
            self.tables = (key, torch.tensor(desc, dtype=torch.int64, device=dev), torch.tensor(rb, dtype=torch.int32, device=dev), len(ents), r)
        _, desc, rb, nt, nrows = self.tables
        if torch.cuda.is_current_stream_capturing() and not any(t is self.tables for t in self.captured_tables):
            self.captured_tables.append(self.tables)
        lib().mt_fp8w_quantize(desc.data_ptr(), rb.data_ptr(), nt, nrows, stream())

    def lookup(self, w32, ver=None):
        ver = w32._version if ver is None else ver
        rows, K = w32.shape[0], w32[0].numel()
        key = (w32.data_ptr(), rows, K)
        e = self.entries.get(key)
        self.last = e  # (the entry of the weight just looked up: the fp8 MFMA path packs its codes, fp8_weight)
        if e is None:
            dev = w32.device
            e = {"src": w32.data_ptr(), "shape": (rows, K), "keep": w32, "weff": torch.empty(w32.shape, dtype=torch.float32, device=dev),
                 "codes": torch.empty((rows, K), dtype=torch.uint8, device=dev), "scale": torch.empty(rows, dtype=torch.float32, device=dev),
                 "ver": ver, "count": 0, "seen": WEIGHT_EPOCH}
            self.entries[key] = e
            self.last = e
            tb, self.tables = self.tables, None
            self._run([e])
            self.tables = tb
            return e["weff"], (WEIGHT_EPOCH, e["count"])
        e["seen"] = WEIGHT_EPOCH
        if torch.cuda.is_current_stream_capturing():
            e["pinned"] = True
        if self.epoch != WEIGHT_EPOCH:
            self.epoch = WEIGHT_EPOCH
            if not torch.cuda.is_current_stream_capturing():  # as _PackRegistry.lookup
                for k in [k for k, v in self.entries.items() if WEIGHT_EPOCH - v["seen"] > _PackRegistry.KEEP and not v.get("pinned")]:
                    del self.entries[k]
            self._run(list(self.entries.values()))
            for v in self.entries.values():
                v["ver"] = None
        if e["ver"] is None:
            e["ver"] = ver
        elif e["ver"] != ver:  # the master was written through torch since the shadow was made
            e["ver"], e["count"] = ver, e["count"] + 1
            tb, self.tables = self.tables, None
            self._run([e])
            self.tables = tb
        return e["weff"], (WEIGHT_EPOCH, e["count"])


PACK_FWD, PACK_DGRAD, QUANT = _PackRegistry(0), _PackRegistry(1), _QuantRegistry()
WEIGHT_QUANT = None  # None | "fp8": conv (+BatchNorm) weights as e4m3fn codes with per-output-channel power-of-two scales


def set_weight_quant(mode):
    """None (default) or "fp8": BASELINE configs[4].  Applies to the weights of Conv modules (dense / grouped convs followed by
    BatchNorm: 99 % of the parameters); depth-wise filters and the heads' final 1x1 projections (with bias) stay in full precision."""
    global WEIGHT_QUANT
    if mode not in (None, "fp8"):
        raise ValueError("weight quantisation mode must be None or 'fp8'")
    WEIGHT_QUANT = mode
    bump_param_epoch()


def weight_quant():
    return WEIGHT_QUANT


# ---- fp8 MFMA convolutions (csrc/conv3x3_fp8.hip; BASELINE configs[4]) ------------------------------------------------------------
# With `set_weight_quant("fp8")` AND `set_fp8_conv(True)` the forward of every bf16 3x3 stride-1 Conv whose geometry the kernel serves
# (y3d_conv3x3_fp8_ok: the head's two 3x3 layers at the S / B / L / X widths) runs on v_mfma_scale_f32_16x16x128_f8f6f4: e4m3 weight
# codes (the fp8w quantiser's) x e4m3 activations with one E8M0 scale per (pixel, 32 channels).  The activation's fp8 copy comes from its
# producer when that is a BatchNorm + SiLU pass told to write one (`want_fp8_copy`: y3d_bn_act_fwd_q), else from y3d_fp8_quantize_act.
# Data and weight gradients stay on the bf16 kernels (x and w_eff in bf16: straight-through).
FP8_CONV = False
_FP8_WANT = False
_FP8_ACT = {}  # address of a bf16 activation -> (q, s, shape, z): the fp8 copy its producer wrote; dropped at the next forward


def set_fp8_conv(on: bool):
    global FP8_CONV
    if on and WEIGHT_QUANT != "fp8":
        raise ValueError('set_fp8_conv(True) needs set_weight_quant("fp8"): the kernel multiplies the fp8w quantiser\'s codes')
    FP8_CONV = bool(on)
    bump_param_epoch()


def fp8_conv():
    return FP8_CONV


class want_fp8_copy:
    """`with want_fp8_copy(): z = conv_bn_act(...)`: the BatchNorm + SiLU pass inside also writes the fp8 copy of z (when fp8 convolutions
    are on and the tensor qualifies); the next fp8 convolution that takes z as its input picks it up"""

    def __enter__(self):
        global _FP8_WANT
        self.prev, _FP8_WANT = _FP8_WANT, FP8_CONV
        return self

    def __exit__(self, *a):
        global _FP8_WANT
        _FP8_WANT = self.prev


def fp8_input(xin):
    """(q (B, H, W, C) e4m3 codes, s (B, H, W, C / 32) E8M0 bytes) of the bf16 NHWC activation `xin`: the producer's copy, or a quantising pass"""
    B, C, H, W = xin.shape
    ent = _FP8_ACT.pop(xin.data_ptr(), None)
    if ent is not None and ent[2] == (B, C, H, W):
        return ent[0], ent[1]
    if not px_dense(xin):
        xin = to_nhwc(xin, torch.bfloat16, dense=True).contiguous(memory_format=torch.channels_last)
    q = torch.empty(B, H, W, C, dtype=torch.uint8, device=xin.device)
    s = torch.empty(B, H, W, lib().fp8_scale_pitch(C), dtype=torch.uint8, device=xin.device)
    lib().fp8_quantize_act(xin.data_ptr(), xin.stride(3), B * H * W, C, q.data_ptr(), s.data_ptr(), stream())
    return q, s


def fp8_weight(ent, Cg):
    """(wq (rows, 9, Cg) bytes, ws (rows,) E8M0 bytes) of a quantised 3x3 weight (a _QuantRegistry entry), repacked when its codes changed"""
    tok = (WEIGHT_EPOCH, ent["count"])
    hit = ent.get("wq")
    if hit is None or hit[0] != tok:
        rows = ent["shape"][0]
        dev = ent["codes"].device
        wq = hit[1] if hit is not None else torch.empty(rows, 9, Cg, dtype=torch.uint8, device=dev)
        ws = hit[2] if hit is not None else torch.empty(rows, dtype=torch.uint8, device=dev)
        lib().fp8_pack_weight_fwd(ent["codes"].data_ptr(), ent["scale"].data_ptr(), rows, Cg, wq.data_ptr(), ws.data_ptr(), stream())
        hit = ent["wq"] = (tok, wq, ws)
    return hit[1], hit[2]
PROJ_BN_MFMA = True  # BatchNorm backward of the second head layer recomputing dz on MFMA (tests flip it: materialised path)
STEM_FUSED = True  # the stem in one pass (tests flip it: im2col + dense conv)
PACK_CACHE = True  # weight packs through the registry (one multi-tensor launch per step)
DW_EVAL_FUSED = True  # eval depth-wise conv: folded BatchNorm + SiLU + residual in the conv's epilogue (tests flip it: conv + bn_act_fwd)


# ------------------------------------------------------------------------------------------------------
# Conv (dense / grouped / depth-wise) + BatchNorm + SiLU + residual
# ------------------------------------------------------------------------------------------------------
def conv_bn_act_eval(x, w, gamma, beta, rm, rv, k, s, p, g, act, eps, cache=None, ver=None):
    """eval-mode act(bn(conv(x))) on explicit (possibly stacked / sliced) tensors; no autograd.  `cache`: a dict owned by the
    caller that keeps the packed weights + folded BatchNorm scale/shift until a parameter changes; `ver`: version token of the
    tensors behind a stacked view (StackedConvs.ver)"""
    z, _, _ = _cba_forward(x, w, gamma, beta, rm, rv, k, s, p, g, act, None, 0, False, eps, 0.0, cache, ver=ver)
    return z


_IDENT = {}


def conv_bias_act_eval(x, w, bias, k, s, p, g, act, res, res_mode, cache):
    """act(conv(x) + bias) (+res): a Conv after the reference's BaseModel.fuse() (nn/tasks.py:187-192, conv.py:124-126), run as the eval
    sequence with an IDENTITY BatchNorm - gamma 1, running mean 0, running variance 1, eps 0: scale exactly 1, shift exactly the bias -
    so the folded model goes through the same kernels (affine conv epilogue, residual forms) as the unfolded eval path."""
    _require_gpu(x)
    C = w.shape[0]
    ident = _IDENT.get((str(w.device), C))
    if ident is None:
        ident = _IDENT[(str(w.device), C)] = (torch.ones(C, dtype=torch.float32, device=w.device), torch.zeros(C, dtype=torch.float32, device=w.device))
    ones, zeros = ident
    with torch.no_grad():
        z, _, _ = _cba_forward(x, _w32(w), ones, bias.detach().float(), zeros, ones, k, s, p, g, act, res, res_mode, False, 0.0, 0.0, cache)
    return z


def _eval_consts(cache, kind, w32, g32, b32, rm, rv, ver, dtype, k, g, Cin_g, Cg_pad, Cout, eps):
    """eval-mode constants of one conv, cached in `cache` (a dict owned by the module / stack) until a parameter or buffer changes:
    the packed weights (kind "dense": K-contiguous compute-dtype rows; "dw": tap-major fp32 rows) and the folded BatchNorm
    (scale, shift) pair.  A steady-state eval forward launches neither packing nor bn_eval_scale kernels."""
    L, st, dev = lib(), stream(), w32.device
    vkey = ver[1] if ver is not None else (w32._version, g32._version, b32._version, rm._version, rv._version)
    key = (kind, PARAM_EPOCH, w32.data_ptr(), g32.data_ptr(), rm.data_ptr(), vkey, dtype, k, g, Cg_pad, Cout, float(eps))
    hit = cache.get(key) if cache is not None else None
    if hit is None:
        if kind == "dw":
            wp = _f32(k * k * Cout, dev)
            L.dw_pack_weight(w32.data_ptr(), wp.data_ptr(), Cout, k, k, st)
        else:
            wp = torch.empty(Cout * k * k * Cg_pad, dtype=dtype, device=dev)
            L.pack_weight_fwd(code(dtype), w32.data_ptr(), wp.data_ptr(), Cout, Cin_g, Cg_pad, k, k, st)
        ss = _f32(2 * Cout, dev).view(2, Cout)
        L.bn_eval_scale(Cout, g32.data_ptr(), b32.data_ptr(), rm.data_ptr(), rv.data_ptr(), eps, ss[0].data_ptr(), ss[1].data_ptr(), st)
        hit = (wp, ss, w32)  # w32 kept alive: its address is part of the key
        if cache is not None:  # owned by the module / stack, so it dies with the tensors it describes
            for old in [q for q in cache if q[1] != PARAM_EPOCH]:
                del cache[old]
            cache[key] = hit
    return hit[0], hit[1]


def _timed(key, launch):
    if TIMER is not None:
        TIMER.bracket(key, launch)
    else:
        launch()


def _cba_forward(x, w32, g32, b32, rm, rv, k, s, p, g, act, res, res_mode, training, eps, momentum, cache=None, bn_apply=True, pack_cache=True,
                 ver=None, quant=True, pre_conv=None):
    """conv -> BN statistics -> BN apply + SiLU (+res).  Returns (z, saved) with everything the backward needs.
    ver = (weights token, all-tensors token) of a stacked view (StackedConvs), None for tensors that carry their own counters."""
    L = lib()
    _require_gpu(x)
    dtype = _COMPUTE_DTYPE
    dt, c = code(dtype), ce(dtype)
    st = stream()
    dev = x.device
    B, Cin, H, W = x.shape
    qent = None
    if WEIGHT_QUANT and quant and not (g > 1 and g == Cin and g == w32.shape[0]):
        # fp8w mode: the kernels pack and multiply the fp8-valued shadow of the weight; the master keeps receiving the gradient
        w32, qtok = QUANT.lookup(w32, ver[0] if ver is not None else None)
        qent = QUANT.last
        ver = ((qtok,), (qtok,) + (ver[1] if ver is not None else (g32._version, b32._version, rm._version, rv._version)))
    Cout, Cg_w, kh, kw = w32.shape
    assert kh == k and kw == k and Cin // g == Cg_w, "Conv: weight shape does not match input"
    Ho = (H + 2 * p - k) // s + 1
    Wo = (W + 2 * p - k) // s + 1
    M = B * Ho * Wo
    if Cin == 3 and k == 3 and s == 2 and p == 1 and g == 1 and not x.requires_grad:
        # the stem: im2col the image once (27 window values + 5 zeros per output pixel) and run a dense 1x1 conv with K = 32;
        # as a 9-tap conv over an 8-channel-padded image the MFMA tiles were 86 % padding.  dW is mapped back in _cba_backward.
        wkey = ("stemcol", PARAM_EPOCH, w32.data_ptr(), ver[0] if ver is not None else w32._version)
        wcol = cache.get(wkey) if (cache is not None and not training) else None
        if wcol is None:
            wcol = torch.zeros(Cout, 32, dtype=torch.float32, device=dev)
            wcol[:, :27] = w32.detach().permute(0, 2, 3, 1).reshape(Cout, 27)  # column (r*3+q)*3+ci
            if cache is not None and not training:  # eval: the im2col form of the stem weight is a constant of the forward
                for old in [q for q in cache if q[1] != PARAM_EPOCH]:
                    del cache[old]
                cache[wkey] = wcol
        if (not training and not res_mode and dtype == torch.bfloat16 and Cout % 16 == 0 and 16 <= Cout <= 80 and STEM_FUSED
                and x.dtype in (torch.uint8, torch.float32)):
            # eval: gather + MFMA + folded BatchNorm + SiLU in one pass over the image (csrc/stem_fused.hip); no column tensor
            _, ss = _eval_consts(cache, "dense", wcol.view(Cout, 32, 1, 1), g32, b32, rm, rv, ver, dtype, 1, 1, 32, 32, Cout, eps)
            if x.dtype == torch.uint8:
                hwc = int(not x.is_contiguous() and x.permute(0, 2, 3, 1).is_contiguous())
                xs, mode = (x if (hwc or x.is_contiguous()) else x.contiguous()), 1 + hwc
            else:
                xs, mode = x.contiguous(), 0
            ye = out_tensor(B, Cout, Ho, Wo, dtype, dev)
            _timed(("conv_eval", dt, B, H, W, 3, Cout, 3, 2, 1, 1),
                   lambda: L.stem_conv_eval(xs.data_ptr(), mode, wcol.data_ptr(), ss[0].data_ptr(), ss[1].data_ptr(), int(act), ye.data_ptr(), ye.stride(3),
                                            B, H, W, Cout, st))
            return ye, None, None
        xcol = nhwc_empty(B, 32, Ho, Wo, dtype, dev)
        if training and dtype == torch.bfloat16 and Cout % 16 == 0 and 16 <= Cout <= 80 and STEM_FUSED and x.dtype in (torch.uint8, torch.float32):
            # training: the same one-pass kernel writes the raw conv output, the BatchNorm partials and - its own B operand - the column
            # tensor the weight gradient reads; the two-step form wrote the column tensor and read it back (im2col 167 us + conv 146 us)
            if x.dtype == torch.uint8:
                hwc = int(not x.is_contiguous() and x.permute(0, 2, 3, 1).is_contiguous())
                xs, mode = (x if (hwc or x.is_contiguous()) else x.contiguous()), 1 + hwc
            else:
                xs, mode = x.contiguous(), 0
            rows = L.stem_conv_train_rows(B, H, W)
            ypre = nhwc_empty(B, Cout, Ho, Wo, dtype, dev)
            part = _f32(rows * Cout * 2, dev)
            _timed(("conv_fwd", dt, B, H, W, 3, Cout, 3, 2, 1),
                   lambda: L.stem_conv_train(xs.data_ptr(), mode, wcol.data_ptr(), ypre.data_ptr(), ypre.stride(3), xcol.data_ptr(), part.data_ptr(), B, H, W, Cout, st))
            z, cfg, saved = _cba_forward(xcol, wcol.view(Cout, 32, 1, 1), g32, b32, rm, rv, 1, 1, 0, 1, act, res, res_mode, training, eps, momentum, cache,
                                         pack_cache=False, quant=False, pre_conv=(ypre, part, rows))
            return z, (cfg + ("stem",) if cfg is not None else None), saved
        if x.dtype == torch.uint8:  # the dataset's bytes: /255 happens in the kernel (NCHW, or NHWC when the tensor is channels-last)
            hwc = int(not x.is_contiguous() and x.permute(0, 2, 3, 1).is_contiguous())
            xs = x if (hwc or x.is_contiguous()) else x.contiguous()
            L.stem_im2col_u8(dt, xs.data_ptr(), hwc, xcol.data_ptr(), B, H, W, Ho, Wo, st)
        else:
            L.stem_im2col(dt, x.float().contiguous().data_ptr(), xcol.data_ptr(), B, H, W, Ho, Wo, st)
        z, cfg, saved = _cba_forward(xcol, wcol.view(Cout, 32, 1, 1), g32, b32, rm, rv, 1, 1, 0, 1, act, res, res_mode, training, eps, momentum, cache,
                                     pack_cache=False, quant=False)  # wcol is built from the already quantised stem weight
        return z, (cfg + ("stem",) if cfg is not None else None), saved
    dw = g > 1 and g == Cin and g == Cout
    # ---- input: NHWC compute dtype; the stem (Cin=3) is channel-padded while converting from NCHW fp32
    Cin_k = Cin
    if Cin % c != 0:
        if g != 1:
            raise Y3DError(f"grouped conv with {Cin} channels is not 16-byte chunkable")
        Cin_k = (Cin + c - 1) // c * c
        xin = nhwc_empty(B, Cin_k, H, W, dtype, dev)
        L.nchw_to_nhwc(dt, x.float().contiguous().data_ptr(), xin.data_ptr(), B, Cin, H, W, Cin_k, st)
    else:
        xin = to_nhwc(x, dtype)
    sb, sh, sw = s3(xin)
    dw_fused = dw and not training and bn_apply and DW_EVAL_FUSED
    y = nhwc_empty(B, Cout, Ho, Wo, dtype, dev) if ((dw or training or res_mode) and not dw_fused) else None  # pre-BatchNorm tensor (the eval fast paths have none)
    part = None
    if dw:
        if Cout % c != 0:
            raise Y3DError(f"depth-wise conv with {Cout} channels is not 16-byte chunkable")
        nblk = L.dw_blocks(M)
        if training:
            part = _f32(nblk * Cout * 2, dev)
        ss_eval = None
        if training:
            wp = _f32(k * k * Cout, dev)
            L.dw_pack_weight(w32.data_ptr(), wp.data_ptr(), Cout, k, k, st)
        else:
            wp, ss_eval = _eval_consts(cache, "dw", w32, g32, b32, rm, rv, ver, dtype, k, g, 1, 1, Cout, eps)
            if dw_fused:
                # eval: folded BatchNorm + SiLU (+ the RepVGGDW / shortcut residual) in the depth-wise kernel's epilogue - one launch, no
                # pre-BN tensor (12 depth-wise layers of S-3D: 12 bn_act_fwd launches and their read + write less per forward)
                ze = out_tensor(B, Cout, Ho, Wo, dtype, dev)
                rr = to_nhwc(res, dtype, dense=True) if res_mode else None
                if rr is not None:
                    assert rr.shape == ze.shape, "residual shape mismatch"
                L.dwconv2d_fwd_affine(dt, xin.data_ptr(), sb, sh, sw, B, H, W, Cin, wp.data_ptr(), ss_eval[0].data_ptr(), ss_eval[1].data_ptr(), int(act),
                                      res_mode, rr.data_ptr() if rr is not None else None, rr.stride(3) if rr is not None else 0,
                                      ze.data_ptr(), ze.stride(3), Ho, Wo, k, k, s, p, st)
                return ze, None, None
        L.dwconv2d_fwd(dt, xin.data_ptr(), sb, sh, sw, B, H, W, Cin, wp.data_ptr(), y.data_ptr(), Cout, Ho, Wo, k, k, s, p,
                       part.data_ptr() if training else None, st)
    elif pre_conv is not None:  # the caller has run the conv (fused training stem): pre-BN tensor + BatchNorm partial rows
        y, part, nblk = pre_conv
    else:
        # fp8 MFMA forward (set_fp8_conv): e4m3 codes of the weight x the MX-quantised activation; everything after the conv - BatchNorm
        # statistics from the partial rows, the apply pass, the bf16 backward over (xin, w_eff) - is the bf16 path's
        f8 = (FP8_CONV and qent is not None and dtype == torch.bfloat16 and k == 3 and s == 1 and p == 1 and Cin_k == Cin
              and bool(L.conv3x3_fp8_ok(B, H, W, Cin, Cout, g)))
        nblk = L.conv3x3_fp8_stat_rows(B, H, W) if f8 else L.conv2d_stat_rows(dt, B, H, W, Cin_k, Cout, g, k, k, s, p)
        if training:
            part = _f32(nblk * Cout * 2, dev)
        Cg_pad = Cin_k // g
        ss_eval = None
        if not training:
            wp, ss_eval = _eval_consts(cache, "dense", w32, g32, b32, rm, rv, ver, dtype, k, g, Cin // g, Cg_pad, Cout, eps)
        if f8:
            xq, xs = fp8_input(xin)
            wq, ws = fp8_weight(qent, Cin // g)
            if not training and not res_mode:
                ye = out_tensor(B, Cout, Ho, Wo, dtype, dev)
                _timed(("conv_eval_fp8", dt, B, H, W, Cin_k, Cout, k, s, g, p),
                       lambda: L.conv3x3_fp8_fwd(xq.data_ptr(), xs.data_ptr(), B, H, W, Cin, wq.data_ptr(), ws.data_ptr(), ye.data_ptr(), ye.stride(3), Cout, g, None,
                                                 ss_eval[0].data_ptr(), ss_eval[1].data_ptr(), int(act), st))
                return ye, None, None
            _timed(("conv_fwd_fp8", dt, B, H, W, Cin_k, Cout, k, s, g),
                   lambda: L.conv3x3_fp8_fwd(xq.data_ptr(), xs.data_ptr(), B, H, W, Cin, wq.data_ptr(), ws.data_ptr(), y.data_ptr(), y.stride(3), Cout, g,
                                             part.data_ptr() if training else None, None, None, 0, st))
        elif not training and not res_mode:
            # eval: BatchNorm (running statistics) + SiLU folded into the conv epilogue - one launch, no pre-BN tensor; the packed
            # weights and the scale/shift pair are cached until a parameter / buffer is modified in place or re-pointed
            ye = out_tensor(B, Cout, Ho, Wo, dtype, dev)  # a pending placement (C2f / SPPF / Concat slot) is honoured in eval too
            _timed(("conv_eval", dt, B, H, W, Cin_k, Cout, k, s, g, p),
                   lambda: L.conv2d_fwd_affine(dt, xin.data_ptr(), sb, sh, sw, B, H, W, Cin_k, wp.data_ptr(), ss_eval[0].data_ptr(), ss_eval[1].data_ptr(),
                                               int(act), ye.data_ptr(), ye.stride(3), Ho, Wo, Cout, g, k, k, s, p, st))
            return ye, None, None
        if f8:
            pass
        elif (not training and res_mode == 1 and dtype == torch.bfloat16
                and L.conv2d_fwd_affine_res_ok(dt, B, H, W, Cin_k, Cout, g, k, k, s, p)):
            # eval Bottleneck shortcut: conv + folded BatchNorm + SiLU + residual in ONE launch (the narrow resident-weight kernel)
            rr = to_nhwc(res, dtype, dense=True)
            ye = out_tensor(B, Cout, Ho, Wo, dtype, dev)
            L.conv2d_fwd_affine_res(dt, xin.data_ptr(), sb, sh, sw, B, H, W, Cin_k, wp.data_ptr(), ss_eval[0].data_ptr(), ss_eval[1].data_ptr(), int(act),
                                    rr.data_ptr(), rr.stride(3), ye.data_ptr(), ye.stride(3), Ho, Wo, Cout, g, k, k, s, p, st)
            return ye, None, None
        if f8:
            pass  # the conv ran above
        elif not training:
            pass  # eval with a residual: conv (cached packed weights) + one BatchNorm / SiLU / residual pass below
        elif pack_cache and training and PACK_CACHE:
            wp = PACK_FWD.lookup(w32, w32.data_ptr(), (Cout, Cin // g, Cg_pad, k * k, k * k * Cg_pad, dt), dtype, ver[0] if ver is not None else None)
        else:
            wp = torch.empty(Cout * k * k * Cg_pad, dtype=dtype, device=dev)
            L.pack_weight_fwd(dt, w32.data_ptr(), wp.data_ptr(), Cout, Cin // g, Cg_pad, k, k, st)
        if not f8:
            _timed(("conv_fwd", dt, B, H, W, Cin_k, Cout, k, s, g),
                   lambda: L.conv2d_fwd(dt, xin.data_ptr(), sb, sh, sw, B, H, W, Cin_k, wp.data_ptr(), None, y.data_ptr(), Cout, Ho, Wo, Cout, g,
                                        k, k, s, p, part.data_ptr() if training else None, st))
    if training:
        stats = _f32(6 * Cout, dev).view(6, Cout)  # mean, invstd, scale, shift, mean_g, mean_gx
        bump_param_epoch()  # bn_finalize updates the running statistics in place
        L.bn_finalize(part.data_ptr(), nblk, Cout, M, g32.data_ptr(), b32.data_ptr(), eps, momentum, rm.data_ptr(), rv.data_ptr(),
                      stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(), st)
    else:
        stats = None  # eval: the cached (scale, shift) pair
        if not bn_apply:  # the fused consumer reads rows 2 / 3 of a statistics tensor
            stats = torch.zeros(6, Cout, dtype=torch.float32, device=dev)
            stats[2:4].copy_(ss_eval)
    if not bn_apply:  # the consumer applies BatchNorm + activation itself (FusedConvBNProjFn): hand back the pre-BN tensor
        cfg = (B, Cin, Cin_k, H, W, Cout, Ho, Wo, k, s, p, g, dw, res_mode, training, dtype, int(act), ver[0] if ver is not None else None)
        return y, cfg, (xin, w32, y, stats, None)
    z = out_tensor(B, Cout, Ho, Wo, dtype, dev)
    rr = None
    if res_mode:
        rr = to_nhwc(res, dtype, dense=True)
        assert rr.shape == z.shape, "residual shape mismatch"
    sc_t, sh_t = (stats[2], stats[3]) if training else (ss_eval[0], ss_eval[1])
    if _FP8_WANT and dtype == torch.bfloat16 and not res_mode and Cout % 64 == 0 and not dw:
        # the consumer is an fp8 MFMA convolution: this pass also writes z's fp8 copy (e4m3 codes + E8M0 block scales)
        zq = torch.empty(B, Ho, Wo, Cout, dtype=torch.uint8, device=dev)
        zs = torch.empty(B, Ho, Wo, L.fp8_scale_pitch(Cout), dtype=torch.uint8, device=dev)
        L.bn_act_fwd_q(y.data_ptr(), Cout, sc_t.data_ptr(), sh_t.data_ptr(), int(act), z.data_ptr(), z.stride(3), zq.data_ptr(), zs.data_ptr(), M, Cout, st)
        _FP8_ACT[z.data_ptr()] = (zq, zs, (B, Cout, Ho, Wo), z)
    else:
        L.bn_act_fwd(dt, y.data_ptr(), Cout, sc_t.data_ptr(), sh_t.data_ptr(), int(act), res_mode,
                     rr.data_ptr() if rr is not None else None, rr.stride(3) if rr is not None else 0, z.data_ptr(), z.stride(3), M, Cout, st)
    # [17]: version token of the weights as packed (stacked views / fp8 shadows carry their identity outside the tensor's own counter)
    cfg = (B, Cin, Cin_k, H, W, Cout, Ho, Wo, k, s, p, g, dw, res_mode, training, dtype, int(act), ver[0] if ver is not None else None)
    return z, cfg, (xin, w32, y, stats, rr if res_mode == 2 else None)


def _cba_backward(cfg, saved, dz, need_dx, need_dres, dx_range=None, pre=None, dx_out=None):
    """-> dx, dW (fp32 OIHW), dgamma, dbeta, dres.  dx_range=(lo, hi): only output channels lo..hi feed dx
    (the one-to-one head sees a detached input, reference head.py:820)."""
    L = lib()
    xin, w32, y, stats, rr = saved
    stem = cfg[-1] == "stem"
    B, Cin, Cin_k, H, W, Cout, Ho, Wo, k, s, p, g, dw, res_mode, training, dtype, act = cfg[:17]
    if not training:
        raise Y3DError("backward through an eval-mode (running-statistics) Conv is not supported")
    if pre is not None:  # (dy, dgb): the BatchNorm part was done by the caller (FusedConvBNProjFn)
        return _conv_backward(cfg, saved, pre[0], pre[1], None, need_dx, dx_range, dx_out)
    dt = code(dtype)
    st = stream()
    dev = dz.device
    esz = 2 if dtype == torch.bfloat16 else 4
    M = B * Ho * Wo
    dz = to_nhwc(dz, dtype, dense=True)
    nb = L.bn_bwd_blocks(M, Cout)
    part = _f32(nb * Cout * 2, dev)
    rptr = rr.data_ptr() if rr is not None else None
    rsw = rr.stride(3) if rr is not None else 0
    L.bn_act_bwd_reduce(dt, y.data_ptr(), Cout, dz.data_ptr(), dz.stride(3), rptr, rsw, stats[2].data_ptr(), stats[3].data_ptr(),
                        stats[0].data_ptr(), stats[1].data_ptr(), act, res_mode, part.data_ptr(), M, Cout, st)
    dgb = _f32(2 * Cout, dev).view(2, Cout)
    L.bn_bwd_finalize(part.data_ptr(), nb, Cout, M, dgb[0].data_ptr(), dgb[1].data_ptr(), 0, stats[4].data_ptr(), stats[5].data_ptr(), st)
    dy = nhwc_empty(B, Cout, Ho, Wo, dtype, dev)
    dres = None
    if res_mode == 2 and need_dres:
        dres = nhwc_empty(B, Cout, Ho, Wo, dtype, dev)
    L.bn_act_bwd_apply(dt, y.data_ptr(), Cout, dz.data_ptr(), dz.stride(3), rptr, rsw, stats[2].data_ptr(), stats[3].data_ptr(),
                       stats[0].data_ptr(), stats[1].data_ptr(), stats[4].data_ptr(), stats[5].data_ptr(), act, res_mode, 1,
                       dy.data_ptr(), Cout, dres.data_ptr() if dres is not None else None, Cout, M, Cout, st)
    if res_mode == 1:
        dres = dz
    return _conv_backward(cfg, saved, dy, dgb, dres, need_dx, dx_range, dx_out)


def _conv_backward(cfg, saved, dy, dgb, dres, need_dx, dx_range, dx_out=None):
    """data and weight gradients of the conv given dy (gradient wrt its pre-BatchNorm output); cfg[17] = version token of the weights
    as the forward packed them"""
    L = lib()
    xin, w32, y, stats, rr = saved
    stem = cfg[-1] == "stem"
    B, Cin, Cin_k, H, W, Cout, Ho, Wo, k, s, p, g, dw, res_mode, training, dtype, act = cfg[:17]
    wver = cfg[17]
    dt = code(dtype)
    st = stream()
    dev = dy.device
    esz = 2 if dtype == torch.bfloat16 else 4
    M = B * Ho * Wo
    sb, sh, sw = s3(xin)
    dx = None
    dW = torch.empty_like(w32)
    if dw:
        wp = _f32(k * k * Cout, dev)
        L.dw_pack_weight(w32.data_ptr(), wp.data_ptr(), Cout, k, k, st)
        if need_dx:
            dx = nhwc_empty(B, Cin, H, W, dtype, dev)
            dsb, dsh, dsw = s3(dy)
            L.dwconv2d_bwd_data(dt, dy.data_ptr(), dsb, dsh, dsw, B, Ho, Wo, Cout, wp.data_ptr(), dx.data_ptr(), Cin, H, W, k, k, s, p, st)
        slab = _f32(L.dw_wgrad_blocks(M) * k * k * Cout, dev)
        L.dwconv2d_bwd_weight(dt, xin.data_ptr(), sb, sh, sw, B, H, W, Cin, dy.data_ptr(), Cout, Ho, Wo, k, k, s, p, slab.data_ptr(),
                              dW.data_ptr(), 0, st)
    else:
        if need_dx and Cin_k == Cin:
            lo, hi = dx_range if dx_range is not None else (0, Cout)
            assert g == 1 or (lo, hi) == (0, Cout)
            co = hi - lo
            kp = L.conv_kpad(dt, k * k * (co // g))
            src = w32.data_ptr() + lo * (Cin // g) * k * k * 4
            if stem or not PACK_CACHE:
                wpd = torch.empty(Cin * kp, dtype=dtype, device=dev)
                L.pack_weight_dgrad(dt, src, wpd.data_ptr(), co, Cin // g, g, k, k, st)
            else:
                wpd = PACK_DGRAD.lookup(w32, src, (g, co // g, Cin // g, k * k, kp, dt), dtype, wver)
            dx = dx_out if dx_out is not None else nhwc_empty(B, Cin, H, W, dtype, dev)  # dx_out: a slice of a shared gradient buffer (grad_slot)
            dsb, dsh, dsw = s3(dy)
            _timed(("conv_dgrad", dt, B, H, W, Cin, co, k, s, g),
                   lambda: L.conv2d_bwd_data(dt, dy.data_ptr() + lo * esz, dsb, dsh, dsw, B, Ho, Wo, co, wpd.data_ptr(), dx.data_ptr(), dx.stride(3), H, W,
                                             Cin, g, k, k, s, p, st))
        ns = L.conv2d_wgrad_plan(dt, B, H, W, Cin_k, Cout, g, k, k, s, p)
        slab = _f32(ns * Cout * k * k * (Cin_k // g), dev)
        _timed(("conv_wgrad", dt, B, H, W, Cin_k, Cout, k, s, g),
               lambda: L.conv2d_bwd_weight(dt, xin.data_ptr(), sb, sh, sw, B, H, W, Cin_k, Cin, dy.data_ptr(), Cout, Ho, Wo, Cout, g, k, k,
                                           s, p, slab.data_ptr(), ns, dW.data_ptr(), 0, st))
    if stem:  # [Cout][(r*3+q)*3+ci | 5 zeros] -> OIHW (Cout, 3, 3, 3); the image itself gets no gradient
        dW = dW.reshape(Cout, 32)[:, :27].reshape(Cout, 3, 3, 3).permute(0, 3, 1, 2).contiguous()
        dx = None
    return dx, dW, dgb[0], dgb[1], dres


def _w32(weight):
    w = weight.detach()
    if w.dtype != torch.float32 or not w.is_contiguous():
        w = w.float().contiguous()
    return w


class ConvBNActFn(torch.autograd.Function):
    """act(bn(conv(x))) (+res)   — reference nn/modules/conv.py:120-122 plus the residual adds of
    block.py:342,711,758,816-817 fused into the BN-apply kernel.

    `m` is the owning Conv module (non-tensor): k, s, p, g, act, BN buffers/eps/momentum, training flag."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, res, res_mode, m):
        z, cfg, saved = _cba_forward(x, _w32(weight), gamma.detach().float(), beta.detach().float(), m.bn.running_mean, m.bn.running_var,
                                     m.k, m.s, m.p, m.g, m.has_act, res, res_mode, m.training, m.eps, m.momentum,
                                     m.__dict__.setdefault("_eval_cache", {}))
        if m.training:
            m._nbt_pending += 1
        ctx.cfg = cfg
        if saved is not None:
            ctx.save_for_backward(*saved)
        return z

    @staticmethod
    def backward(ctx, dz):
        if ctx.cfg is None:
            raise Y3DError("backward through an eval-mode (running-statistics) Conv is not supported")
        dx, dW, dg, db, dres = _cba_backward(ctx.cfg, ctx.saved_tensors, dz, ctx.needs_input_grad[0], ctx.needs_input_grad[4])
        return dx, dW, dg, db, dres, None, None


class StackedConvs:
    """N sibling Conv modules (same input, kernel, stride) run as ONE convolution with their output channels stacked
    (the 8 branches x 2 head sets of v10Detect3d share their input).  The per-branch Parameters / BN buffers
    keep their identity and state_dict keys; their storage is re-pointed at slices of one flat tensor, so stacking costs
    nothing per step and optimizer / DDP / load_state_dict keep working on the per-branch tensors."""

    def __init__(self, convs, groups=1):
        self.convs = list(convs)
        self.groups = groups
        self.couts = [c.conv.out_channels for c in self.convs]
        self.flat = {}

    def _ensure(self, name, tensors, setter):
        flat = self.flat.get(name)
        ok = flat is not None and flat.device == tensors[0].device
        if ok:
            base, off = flat.data_ptr(), 0
            for t in tensors:
                if t.data_ptr() != base + off * 4:
                    ok = False
                    break
                off += t.numel()
        if not ok:
            flat = torch.cat([t.detach().reshape(-1).float() for t in tensors])
            off = 0
            for i, t in enumerate(tensors):
                n = t.numel()
                setter(i, flat[off:off + n].view(t.shape))
                off += n
            self.flat[name] = flat
        return flat

    def tensors(self):
        cs = self.convs
        w = self._ensure("w", [c.conv.weight for c in cs], lambda i, v: setattr(cs[i].conv.weight, "data", v))
        gm = self._ensure("g", [c.bn.weight for c in cs], lambda i, v: setattr(cs[i].bn.weight, "data", v))
        bt = self._ensure("b", [c.bn.bias for c in cs], lambda i, v: setattr(cs[i].bn.bias, "data", v))
        rm = self._ensure("rm", [c.bn.running_mean for c in cs], lambda i, v: cs[i].bn._buffers.__setitem__("running_mean", v))
        rv = self._ensure("rv", [c.bn.running_var for c in cs], lambda i, v: cs[i].bn._buffers.__setitem__("running_var", v))
        c0 = cs[0].conv
        # identity of what the flat tensors hold, as torch sees it: the per-branch tensors' own version counters (the flat views'
        # counters never move: `.data =` shares storage, not the counter)
        wv = tuple(c.conv.weight._version for c in cs)
        self.ver = (wv, wv + tuple(t._version for c in cs for t in (c.bn.weight, c.bn.bias, c.bn.running_mean, c.bn.running_var)))
        return w.view(sum(self.couts), c0.in_channels // c0.groups, *c0.kernel_size), gm, bt, rm, rv

    def params(self):
        return [c.conv.weight for c in self.convs] + [c.bn.weight for c in self.convs] + [c.bn.bias for c in self.convs]


class FusedConvBNActFn(torch.autograd.Function):
    """One conv+BN+SiLU over the stacked output channels of `stack.convs`.  args: x, stack, groups, dx_range, *params
    (params only tie the per-branch Parameters into the autograd graph; the math runs on the stacked storage)."""

    @staticmethod
    def forward(ctx, x, stack, groups, dx_range, *params):
        w, gm, bt, rm, rv = stack.tensors()
        m = stack.convs[0]
        z, cfg, saved = _cba_forward(x, w, gm, bt, rm, rv, m.k, m.s, m.p, groups, m.has_act, None, 0, m.training, m.eps, m.momentum, ver=stack.ver)
        if m.training:
            for c in stack.convs:
                c._nbt_pending += 1
        ctx.cfg, ctx.couts, ctx.dx_range = cfg, stack.couts, dx_range
        ctx.gslot = grad_slot(x) if (m.training and torch.is_tensor(x) and x.is_cuda) else None
        if saved is not None:
            ctx.save_for_backward(*saved)
        return z

    @staticmethod
    def backward(ctx, dz):
        dx_out = grad_slot_tensor(ctx.gslot) if (ctx.gslot is not None and ctx.needs_input_grad[0]) else None
        dx, dW, dg, db, _ = _cba_backward(ctx.cfg, ctx.saved_tensors, dz, ctx.needs_input_grad[0], False, ctx.dx_range, dx_out=dx_out)
        dWs, dgs, dbs, off = [], [], [], 0
        for co in ctx.couts:
            dWs.append(dW[off:off + co])
            dgs.append(dg[off:off + co])
            dbs.append(db[off:off + co])
            off += co
        return (dx, None, None, None, *dWs, *dgs, *dbs)


# ------------------------------------------------------------------------------------------------------
# channel slices of one map for several consumers
# ------------------------------------------------------------------------------------------------------
class SplitChannelsFn(torch.autograd.Function):
    """x[:, o_j : o_j + m_j] for every j as strided views (no copy); the backward writes the consumers' gradients side by side into
    ONE tensor.  Plain slicing makes autograd zero-fill a full-size tensor per slice and add them up pairwise: with the 16 branches
    of a non-uniform head (M-3D: 128-channel cls next to 64-channel regression branches) that was 15 full-map adds per level.
    Round 3: consumers that end their backward in a data-gradient kernel (FusedConvBNActFn / FusedConvBNProjFn) write it straight
    into their slice of one shared buffer (`grad_slot`), so a dense split returns that buffer without the concatenation either."""

    @staticmethod
    def forward(ctx, x, offs, widths):
        ctx.offs, ctx.widths, ctx.C = tuple(offs), tuple(widths), x.shape[1]
        covered = sorted(zip(offs, widths))
        ctx.dense = covered[0][0] == 0 and all(a + w == b for (a, w), (b, _) in zip(covered, covered[1:])) and covered[-1][0] + covered[-1][1] == x.shape[1]
        ctx.order = [j for _, j in sorted((o, j) for j, o in enumerate(offs))]
        outs = tuple(x[:, o:o + w] for o, w in zip(offs, widths))
        ctx.holder = None
        if ctx.dense and x.is_cuda and len(set(offs)) == len(offs) and all(o % ce(x.dtype) == 0 and w % ce(x.dtype) == 0 for o, w in zip(offs, widths)):
            ctx.holder = {"buf": None, "C": x.shape[1], "shape": (x.shape[0], x.shape[2], x.shape[3]), "dtype": x.dtype, "device": x.device}
            ctx.keys = []
            for v, o in zip(outs, offs):
                k = (v.data_ptr(), v.shape[1])
                _GRAD_SLOT[k] = (ctx.holder, o)
                ctx.keys.append(k)
        return outs

    @staticmethod
    def backward(ctx, *grads):
        ref = next(g for g in grads if g is not None)
        B, _, H, W = ref.shape
        if ctx.holder is not None:
            for k in ctx.keys:
                _GRAD_SLOT.pop(k, None)
            buf = ctx.holder["buf"]
            esz = ref.element_size()
            if buf is not None and all(g is not None and g.dtype == buf.dtype and g.data_ptr() == buf.data_ptr() + o * esz and g.shape[1] == w
                                       and g.stride() == buf[:, o:o + w].stride() for g, o, w in zip(grads, ctx.offs, ctx.widths)):
                return buf, None, None  # every consumer wrote its slice in place
        if ctx.dense and all(g is not None for g in grads):
            parts = [to_nhwc(grads[j], ref.dtype, dense=True).permute(0, 2, 3, 1) for j in ctx.order]  # physical (B, H, W, c_j)
            return torch.cat(parts, 3).permute(0, 3, 1, 2), None, None
        dx = nhwc_empty(B, ctx.C, H, W, ref.dtype, ref.device)
        dx.zero_()
        for g, o, w in zip(grads, ctx.offs, ctx.widths):
            if g is not None:
                dx[:, o:o + w].add_(g)
        return dx, None, None


class C2fSplitFn(torch.autograd.Function):
    """`y = list(cv1(x).chunk(2, 1))` of C2f (reference block.py:233) with the second half handed out TWICE (it feeds the first block and
    the concat): -> (y0, y1 for the concat, y1 for the block), all views.  Plain chunk + reuse makes autograd (a) add the two
    gradients of y1 with a strided-operand elementwise kernel and (b) assemble the gradient of the chunked tensor from zero-filled
    full-size pieces.  Here the backward adds the block's gradient INTO the concat gradient's slice (`y3d_add2d`, in place: that slice
    of the concat gradient has no other reader) and returns the first two slices of that buffer as one strided view - no copy at all
    when the two concat gradients are neighbours in one buffer, one gather otherwise."""

    @staticmethod
    def forward(ctx, t):
        c = t.shape[1] // 2
        ctx.c = c
        return t[:, :c], t[:, c:], t[:, c:]

    @staticmethod
    def backward(ctx, d0, d1a, d1b):
        L, st, c = lib(), stream(), ctx.c
        ref = next(g for g in (d0, d1a, d1b) if g is not None)
        dtype, dev = ref.dtype, ref.device
        B, _, H, W = ref.shape
        dt = code(dtype)
        esz = ref.element_size()
        P = B * H * W
        ok = lambda g: g is not None and g.dtype == dtype and is_nhwc(g) and g.data_ptr() % 16 == 0 and g.stride(3) % ce(dtype) == 0 and px_dense(g)
        if (d0 is not None and d1a is not None and ok(d0) and ok(d1a) and d0.stride() == d1a.stride()
                and d1a.data_ptr() == d0.data_ptr() + c * esz and d0.stride(3) >= 2 * c):
            if d1b is not None:
                d1b = to_nhwc(d1b, dtype, dense=True)
                L.add2d(dt, d1a.data_ptr(), d1a.stride(3), d1b.data_ptr(), d1b.stride(3), d1a.data_ptr(), d1a.stride(3), P, c, st)
            return torch.as_strided(d0, (B, 2 * c, H, W), d0.stride(), d0.storage_offset())
        out = nhwc_empty(B, 2 * c, H, W, dtype, dev)
        if d0 is None:
            out[:, :c].zero_()
        else:
            d0 = to_nhwc(d0, dtype, dense=True)
            L.copy2d(dt, d0.data_ptr(), d0.stride(3), out.data_ptr(), 2 * c, P, c, st)
        parts = [to_nhwc(g, dtype, dense=True) for g in (d1a, d1b) if g is not None]
        if not parts:
            out[:, c:].zero_()
        elif len(parts) == 1:
            L.copy2d(dt, parts[0].data_ptr(), parts[0].stride(3), out.data_ptr() + c * esz, 2 * c, P, c, st)
        else:
            L.add2d(dt, parts[0].data_ptr(), parts[0].stride(3), parts[1].data_ptr(), parts[1].stride(3), out.data_ptr() + c * esz, 2 * c, P, c, st)
        return out


# ------------------------------------------------------------------------------------------------------
# plain nn.Conv2d(c, out, 1) with bias (head projections)
# ------------------------------------------------------------------------------------------------------
class HeadProjFn(torch.autograd.Function):
    """cat_j( conv1x1_j(x_j) + b_j ) written straight into one (B, sum(out_j), H, W) NHWC tensor
    (reference head.py:637 + the torch.cat of head.py:742).  args: n, x_0..x_{n-1}, w_0.., b_0.."""

    @staticmethod
    def forward(ctx, n, *args):
        L = lib()
        xs, ws, bs = args[:n], args[n:2 * n], args[2 * n:3 * n]
        dtype = _COMPUTE_DTYPE
        dt = code(dtype)
        st = stream()
        B, _, H, W = xs[0].shape
        P = B * H * W
        couts = [w.shape[0] for w in ws]
        tot = sum(couts)
        out = nhwc_empty(B, tot, H, W, dtype, xs[0].device)
        esz = out.element_size()
        xs = [to_nhwc(x, dtype, dense=True) for x in xs]
        w32 = [w.detach().float().contiguous() for w in ws]
        b32 = [b.detach().float().contiguous() for b in bs]
        off = 0
        wide = []
        for x, w, b, co in zip(xs, w32, b32, couts):
            Cin = x.shape[1]
            # wide projections (the 2D head's 64 box-distribution and nc class outputs): a 1x1 conv with bias on the MFMA kernels,
            # written into its channel slice of `out`; the VALU kernel below is for the 3D head's 1..24 outputs per branch
            mf = co >= 32 and co % 8 == 0 and off % 8 == 0 and tot % 8 == 0 and Cin % 8 == 0
            wide.append(mf)
            if mf:
                wp = torch.empty(co * Cin, dtype=dtype, device=x.device)
                L.pack_weight_fwd(dt, w.data_ptr(), wp.data_ptr(), co, Cin, Cin, 1, 1, st)
                sb, sh, sw = s3(x)
                L.conv2d_fwd(dt, x.data_ptr(), sb, sh, sw, B, H, W, Cin, wp.data_ptr(), b.data_ptr(), out.data_ptr() + off * esz, tot, H, W, co,
                             1, 1, 1, 1, 0, None, st)
            else:
                for o0 in range(0, co, 24):
                    oc = min(24, co - o0)
                    L.proj_fwd(dt, x.data_ptr(), x.stride(3), w.data_ptr() + o0 * w.shape[1] * 4, b.data_ptr() + o0 * 4,
                               out.data_ptr() + (off + o0) * esz, tot, P, x.shape[1], oc, st)
            off += co
        ctx.n, ctx.couts, ctx.dtype, ctx.wide = n, couts, dtype, wide
        ctx.save_for_backward(*xs, *w32)
        return out

    @staticmethod
    def backward(ctx, dout):
        L = lib()
        n, couts, dtype = ctx.n, ctx.couts, ctx.dtype
        xs, ws = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        dt = code(dtype)
        st = stream()
        if not (dout.dtype == dtype and px_dense(dout)):
            dout = to_nhwc(dout, dtype, dense=True) if dout.shape[1] % ce(dtype) == 0 else _dense_any(dout, dtype)
        B, tot, H, W = dout.shape
        P = B * H * W
        dev = dout.device
        esz = dout.element_size()
        dsw = dout.stride(3)
        nb = L.proj_blocks(P)
        dxs, dws, dbs = [], [], []
        off = 0
        for j, (x, w, co) in enumerate(zip(xs, ws, couts)):
            Cin = x.shape[1]
            dx = None
            if ctx.wide[j]:
                dptr = dout.data_ptr() + off * esz
                dsb, dsh, _ = s3(dout)
                if ctx.needs_input_grad[1 + j]:
                    kp = L.conv_kpad(dt, co)
                    wpd = torch.empty(Cin * kp, dtype=dtype, device=dev)
                    L.pack_weight_dgrad(dt, w.data_ptr(), wpd.data_ptr(), co, Cin, 1, 1, 1, st)
                    dx = nhwc_empty(B, Cin, H, W, dtype, dev)
                    L.conv2d_bwd_data(dt, dptr, dsb, dsh, dsw, B, H, W, co, wpd.data_ptr(), dx.data_ptr(), Cin, H, W, Cin, 1, 1, 1, 1, 0, st)
                sb, sh, sw = s3(x)
                ns = L.conv2d_wgrad_plan(dt, B, H, W, Cin, co, 1, 1, 1, 1, 0)
                slab = _f32(ns * co * Cin, dev)
                dW = torch.empty_like(w)
                L.conv2d_bwd_weight(dt, x.data_ptr(), sb, sh, sw, B, H, W, Cin, Cin, dptr, dsw, H, W, co, 1, 1, 1, 1, 0, slab.data_ptr(), ns,
                                    dW.data_ptr(), 0, st)
                nbb = L.bn_bwd_blocks(P, co)  # bias gradient: column sums of the dout slice
                part = _f32(nbb * co * 2, dev)
                L.colsum_partials(dt, dptr, dsw, part.data_ptr(), P, co, st)
                db, scratch = _f32(co, dev), _f32(2 * co, dev)
                L.bn_bwd_finalize(part.data_ptr(), nbb, co, P, None, db.data_ptr(), 0, scratch.data_ptr(), scratch.data_ptr() + 4 * co, st)
                dxs.append(dx)
                dws.append(dW)
                dbs.append(db)
                off += co
                continue
            if ctx.needs_input_grad[1 + j]:
                dx = nhwc_empty(B, Cin, H, W, dtype, dev)
                first = True
                for o0 in range(0, co, 24):
                    oc = min(24, co - o0)
                    if first:
                        L.proj_bwd_data(dt, dout.data_ptr() + (off + o0) * esz, dsw, w.data_ptr() + o0 * Cin * 4, dx.data_ptr(), Cin, P, Cin, oc, st)
                        first = False
                    else:
                        tmp = nhwc_empty(B, Cin, H, W, dtype, dev)
                        L.proj_bwd_data(dt, dout.data_ptr() + (off + o0) * esz, dsw, w.data_ptr() + o0 * Cin * 4, tmp.data_ptr(), Cin, P, Cin, oc, st)
                        L.add2d(dt, dx.data_ptr(), Cin, tmp.data_ptr(), Cin, dx.data_ptr(), Cin, P, Cin, st)
            dW = torch.empty_like(w)
            db = _f32(co, dev)
            slab = _f32(nb * co * Cin, dev)
            bslab = _f32(nb * co, dev)
            L.proj_bwd_weight(dt, x.data_ptr(), x.stride(3), dout.data_ptr() + off * esz, dsw, slab.data_ptr(), bslab.data_ptr(),
                              dW.data_ptr(), db.data_ptr(), 0, P, Cin, co, st)
            dxs.append(dx)
            dws.append(dW)
            dbs.append(db)
            off += co
        return (None, *dxs, *dws, *dbs)


_IDENT = {}  # (channels, device) -> (ones, zeros): the identity BatchNorm of the eval projection


def proj_slices_eval(x, offsets, cin, ws, bs):
    """eval form of HeadProjSlicesFn (no graph): branch j projects channels [offsets[j], offsets[j] + cin) of the ACTIVATED features x with
    its 1x1 conv + bias (head.py:637), all branches in one launch.  bf16 with 64 / 128 channels per branch: the matrix-core kernel of
    the training head (proj_bn_mfma.hip) with an identity BatchNorm (scale 1, shift 0, no activation) - the VALU form
    (y3d_proj_group_fwd) took 42 us per level for the 1 600 candidate pixels of a batch of 32 (128-long serial dot products)."""
    import ctypes
    n = len(ws)
    dtype = _COMPUTE_DTYPE
    if not (dtype == torch.bfloat16 and cin in (64, 128) and PROJ_BN_MFMA and n <= 16):
        return HeadProjSlicesFn.apply(x, offsets, [cin] * n, n, *ws, *bs)
    L, st = lib(), stream()
    x = to_nhwc(x, dtype, dense=True)
    B, Ct, H, W = x.shape
    couts = [w.shape[0] for w in ws]
    tot = sum(couts)
    out = nhwc_empty(B, tot, H, W, dtype, x.device)
    w32 = [w.detach().float().contiguous() for w in ws]
    b32 = [b.detach().float().contiguous() for b in bs]
    ident = _IDENT.get((Ct, x.device))
    if ident is None:
        ident = _IDENT[(Ct, x.device)] = (torch.ones(Ct, dtype=torch.float32, device=x.device), torch.zeros(Ct, dtype=torch.float32, device=x.device))
    PV, IA = ctypes.c_void_p * n, ctypes.c_int * n
    L.proj_group_fwd_bn_mfma(n, cin, x.data_ptr(), x.stride(3), IA(*offsets), PV(*[t.data_ptr() for t in w32]), PV(*[t.data_ptr() for t in b32]), IA(*couts),
                             ident[0].data_ptr(), ident[1].data_ptr(), 0, out.data_ptr(), tot, B * H * W, st)
    return out


class HeadProjSlicesFn(torch.autograd.Function):
    """Same projections as HeadProjFn, but every branch reads a channel slice [off_j, off_j+cin) of ONE stacked feature
    tensor, all branches of a level run as ONE launch per direction (proj_group.hip) and the backward writes each branch's
    input gradient straight into its slice of one gradient tensor.  args: x_full, offsets, cins, n, w_0.., b_0.."""

    @staticmethod
    def forward(ctx, x, offsets, cins, n, *args):
        import ctypes
        L = lib()
        ws, bs = args[:n], args[n:2 * n]
        dtype = _COMPUTE_DTYPE
        dt = code(dtype)
        st = stream()
        x = to_nhwc(x, dtype, dense=True)
        B, Ct, H, W = x.shape
        P = B * H * W
        couts = [w.shape[0] for w in ws]
        cin = cins[0]
        assert all(c == cin for c in cins) and n <= 16, "HeadProjSlicesFn: uniform branch width, at most 16 branches"
        tot = sum(couts)
        out = nhwc_empty(B, tot, H, W, dtype, x.device)
        w32 = [w.detach().float().contiguous() for w in ws]
        b32 = [b.detach().float().contiguous() for b in bs]
        PV = ctypes.c_void_p * n
        IA = ctypes.c_int * n
        c_w, c_b = PV(*[w.data_ptr() for w in w32]), PV(*[b.data_ptr() for b in b32])
        c_off, c_co = IA(*offsets), IA(*couts)
        L.proj_group_fwd(dt, n, cin, x.data_ptr(), x.stride(3), c_off, c_w, c_b, c_co, out.data_ptr(), tot, P, st)
        ctx.meta = (list(offsets), cin, n, couts, dtype)
        ctx.save_for_backward(x, *w32)
        return out

    @staticmethod
    def backward(ctx, dout):
        import ctypes
        L = lib()
        offsets, cin, n, couts, dtype = ctx.meta
        x, ws = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        dt = code(dtype)
        st = stream()
        if not (dout.dtype == dtype and px_dense(dout)):
            dout = _dense_any(dout, dtype)
        B, Ct, H, W = x.shape
        P = B * H * W
        dev = x.device
        tot = sum(couts)
        PV = ctypes.c_void_p * n
        IA = ctypes.c_int * n
        c_w, c_off, c_co = PV(*[w.data_ptr() for w in ws]), IA(*offsets), IA(*couts)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = nhwc_empty(B, Ct, H, W, dtype, dev)
            if n * cin != Ct:
                dx.zero_()
            L.proj_group_bwd_data(dt, n, cin, dout.data_ptr(), dout.stride(3), c_off, c_w, c_co, dx.data_ptr(), Ct, P, st)
        dws = [torch.empty_like(w) for w in ws]
        dbs = [_f32(co, dev) for co in couts]
        nb = L.proj_group_blocks(P)
        slab = _f32(nb * tot * cin, dev)
        bslab = _f32(nb * tot, dev)
        L.proj_group_bwd_weight(dt, n, cin, x.data_ptr(), x.stride(3), c_off, dout.data_ptr(), dout.stride(3), c_co, slab.data_ptr(),
                                bslab.data_ptr(), PV(*[t.data_ptr() for t in dws]), PV(*[t.data_ptr() for t in dbs]), P, st)
        return (dx, None, None, None, *dws, *dbs)


class FusedConvBNProjFn(torch.autograd.Function):
    """Second stacked head layer + the 16 projections of a level as ONE node: grouped conv -> BatchNorm statistics -> projections
    that apply BatchNorm + SiLU on the fly (proj_group.hip).  The activation tensor is never materialised: the forward reads the
    pre-BN tensor once instead of bn_act_fwd's read + write + the projections' read, and the projection weight gradient rebuilds
    the activation from the pre-BN tensor.  (A fully fused backward - BatchNorm reduce / apply recomputing W^T dout per pixel on the
    VALU - was measured slower than the three separate kernels and is not used.)  args: x, stack, groups, offsets, cins, n, *stack.params(), *proj_w, *proj_b"""

    @staticmethod
    def forward(ctx, x, stack, groups, offsets, cins, n, *params):
        import ctypes
        L = lib()
        w, gm, bt, rm, rv = stack.tensors()
        m = stack.convs[0]
        npar = len(params) - 2 * n
        ws, bs = params[npar:npar + n], params[npar + n:]
        ctx.gslot = grad_slot(x) if (m.training and x.is_cuda) else None
        y, cfg, saved = _cba_forward(x, w, gm, bt, rm, rv, m.k, m.s, m.p, groups, m.has_act, None, 0, m.training, m.eps, m.momentum, bn_apply=False,
                                     ver=stack.ver)
        if m.training:
            for c in stack.convs:
                c._nbt_pending += 1
        dtype = _COMPUTE_DTYPE
        dt, st = code(dtype), stream()
        B, Ct, H, W = y.shape
        P = B * H * W
        couts = [t.shape[0] for t in ws]
        cin = cins[0]
        assert all(c == cin for c in cins) and n <= 16 and n * cin == Ct and cin % 64 == 0
        tot = sum(couts)
        out = nhwc_empty(B, tot, H, W, dtype, y.device)
        w32 = [t.detach().float().contiguous() for t in ws]
        b32 = [t.detach().float().contiguous() for t in bs]
        PV, IA = ctypes.c_void_p * n, ctypes.c_int * n
        stats = saved[3]
        if dtype == torch.bfloat16 and cin in (64, 128) and PROJ_BN_MFMA:
            L.proj_group_fwd_bn_mfma(n, cin, y.data_ptr(), y.stride(3), IA(*offsets), PV(*[t.data_ptr() for t in w32]), PV(*[t.data_ptr() for t in b32]),
                                     IA(*couts), stats[2].data_ptr(), stats[3].data_ptr(), int(m.has_act), out.data_ptr(), tot, P, st)
        else:
            L.proj_group_fwd_bn(dt, n, cin, y.data_ptr(), y.stride(3), IA(*offsets), PV(*[t.data_ptr() for t in w32]), PV(*[t.data_ptr() for t in b32]),
                                IA(*couts), stats[2].data_ptr(), stats[3].data_ptr(), int(m.has_act), out.data_ptr(), tot, P, st)
        ctx.cfg, ctx.couts_stack, ctx.meta = cfg, stack.couts, (list(offsets), cin, n, couts, dtype, int(m.has_act))
        ctx.nsaved = len(saved)
        ctx.save_for_backward(*[t for t in saved if t is not None], *w32)
        ctx.none_mask = [t is None for t in saved]
        return out

    @staticmethod
    def backward(ctx, dout):
        import ctypes
        L = lib()
        offsets, cin, n, couts, dtype, act = ctx.meta
        it = iter(ctx.saved_tensors)
        saved = tuple(None if isnone else next(it) for isnone in ctx.none_mask)
        ws = list(it)
        xin, w32c, y, stats, _ = saved
        dt, st = code(dtype), stream()
        if not (dout.dtype == dtype and px_dense(dout)):
            dout = _dense_any(dout, dtype)
        B, Ct, H, W = y.shape
        P = B * H * W
        dev = y.device
        tot = sum(couts)
        PV, IA = ctypes.c_void_p * n, ctypes.c_int * n
        c_w, c_off, c_co = PV(*[t.data_ptr() for t in ws]), IA(*offsets), IA(*couts)
        dws = [torch.empty_like(t) for t in ws]
        dbs = [_f32(co, dev) for co in couts]
        mfma = dtype == torch.bfloat16 and cin in (64, 128) and PROJ_BN_MFMA
        dx_out = grad_slot_tensor(ctx.gslot) if (ctx.gslot is not None and ctx.needs_input_grad[0]) else None
        nb = L.proj_group_bwd_weight_bn_mfma_blocks(P) if mfma else L.proj_group_blocks(P)
        slab, bslab = _f32(nb * tot * cin, dev), _f32(nb * tot, dev)
        if mfma:
            L.proj_group_bwd_weight_bn_mfma(n, cin, y.data_ptr(), y.stride(3), c_off, dout.data_ptr(), dout.stride(3), c_co, stats[2].data_ptr(),
                                            stats[3].data_ptr(), act, slab.data_ptr(), bslab.data_ptr(), PV(*[t.data_ptr() for t in dws]),
                                            PV(*[t.data_ptr() for t in dbs]), P, st)
        else:
            L.proj_group_bwd_weight_bn(dt, n, cin, y.data_ptr(), y.stride(3), c_off, dout.data_ptr(), dout.stride(3), c_co, stats[2].data_ptr(),
                                       stats[3].data_ptr(), act, slab.data_ptr(), bslab.data_ptr(), PV(*[t.data_ptr() for t in dws]),
                                       PV(*[t.data_ptr() for t in dbs]), P, st)
        if mfma and ctx.cfg[14]:
            # BatchNorm backward straight from `dout`: dz = dout . W is recomputed on the matrix cores by the reduce and the apply pass
            # (proj_bn_mfma.hip) instead of being written once and read twice (839 MB at the stride-8 level)
            nblk = L.proj_group_bn_bwd_blocks(P)
            part = _f32(nblk * Ct * 2, dev)
            args = (n, cin, y.data_ptr(), y.stride(3), c_off, dout.data_ptr(), dout.stride(3), c_w, c_co, stats[2].data_ptr(), stats[3].data_ptr(),
                    stats[0].data_ptr(), stats[1].data_ptr())
            L.proj_group_bn_bwd(0, *args, None, None, act, part.data_ptr(), nblk, None, 0, P, Ct, st)
            dgb = _f32(2 * Ct, dev).view(2, Ct)
            L.bn_bwd_finalize(part.data_ptr(), nblk, Ct, P, dgb[0].data_ptr(), dgb[1].data_ptr(), 0, stats[4].data_ptr(), stats[5].data_ptr(), st)
            dy = nhwc_empty(B, Ct, H, W, dtype, dev)
            L.proj_group_bn_bwd(1, *args, stats[4].data_ptr(), stats[5].data_ptr(), act, None, 0, dy.data_ptr(), Ct, P, Ct, st)
            dx, dW, dg, db, _ = _cba_backward(ctx.cfg, saved, None, ctx.needs_input_grad[0], False, None, pre=(dy, dgb), dx_out=dx_out)
        else:
            dz = nhwc_empty(B, Ct, H, W, dtype, dev)
            L.proj_group_bwd_data(dt, n, cin, dout.data_ptr(), dout.stride(3), c_off, c_w, c_co, dz.data_ptr(), Ct, P, st)
            dx, dW, dg, db, _ = _cba_backward(ctx.cfg, saved, dz, ctx.needs_input_grad[0], False, None, dx_out=dx_out)
        dWs, dgs, dbs_s, off = [], [], [], 0
        for co in ctx.couts_stack:
            dWs.append(dW[off:off + co])
            dgs.append(dg[off:off + co])
            dbs_s.append(db[off:off + co])
            off += co
        return (dx, None, None, None, None, None, *dWs, *dgs, *dbs_s, *dws, *dbs)


def _dense_any(x, dtype):
    """pixel-dense NHWC copy without the 16-byte channel constraint (head maps with 38 / 144 channels)"""
    B, C, H, W = x.shape
    out = nhwc_empty(B, C, H, W, dtype, x.device)
    out.copy_(x)
    return out


# ------------------------------------------------------------------------------------------------------
# graph glue
# ------------------------------------------------------------------------------------------------------
class ConcatFn(torch.autograd.Function):
    """torch.cat(xs, 1) on NHWC tensors (Concat conv.py:404; C2f/SPPF/PSA cats). Backward returns channel-slice views."""

    @staticmethod
    def forward(ctx, *xs):
        L = lib()
        dtype = _COMPUTE_DTYPE
        dt = code(dtype)
        st = stream()
        B, _, H, W = xs[0].shape
        cs = [x.shape[1] for x in xs]
        tot = sum(cs)
        ctx.cs = cs
        # inputs that already sit in their slice of one tot-channel NHWC buffer (ops.place) are not copied
        esz = 2 if dtype == torch.bfloat16 else 4
        out, inplace, off = None, [False] * len(xs), 0
        for i, (x, c) in enumerate(zip(xs, cs)):
            if (PLACEMENT and x.dtype == dtype and x.is_cuda and is_nhwc(x) and x.stride(3) == tot and x.stride(2) == W * tot
                    and x.stride(0) == H * W * tot):
                base = x.data_ptr() - off * esz
                if out is None and _CONCAT_BASE.get(base, (None, None))[0] == tot:
                    del _CONCAT_BASE[base]
                    out = torch.as_strided(x, (B, tot, H, W), x.stride(), x.storage_offset() - off)
                    assert out.data_ptr() == base
                    inplace[i] = True
                elif out is not None and out.data_ptr() == base:
                    inplace[i] = True
            off += c
        if out is None:
            out = nhwc_empty(B, tot, H, W, dtype, xs[0].device)
        off = 0
        for i, (x, c) in enumerate(zip(xs, cs)):
            if not inplace[i]:
                x = to_nhwc(x, dtype, dense=True)
                L.copy2d(dt, x.data_ptr(), x.stride(3), out.data_ptr() + off * esz, tot, B * H * W, c, st)
            off += c
        return out

    @staticmethod
    def backward(ctx, dout):
        outs, off = [], 0
        for c in ctx.cs:
            outs.append(dout[:, off:off + c])
            off += c
        return tuple(outs)


class MaxPoolFn(torch.autograd.Function):
    """nn.MaxPool2d(k, 1, k//2) (SPPF block.py:171)"""

    @staticmethod
    def forward(ctx, x, k):
        L = lib()
        dtype = _COMPUTE_DTYPE
        dt = code(dtype)
        x = to_nhwc(x, dtype)
        B, C, H, W = x.shape
        y = out_tensor(B, C, H, W, dtype, x.device)
        need = x.requires_grad
        arg = torch.empty(B * H * W * C, dtype=torch.uint8, device=x.device) if need else None
        sb, sh, sw = s3(x)
        L.maxpool_fwd(dt, x.data_ptr(), sb, sh, sw, y.data_ptr(), y.stride(3), arg.data_ptr() if need else None, B, H, W, C, k, stream())
        ctx.k = k
        ctx.dtype = dtype
        if need:
            ctx.save_for_backward(arg)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = lib()
        (arg,) = ctx.saved_tensors
        dtype = ctx.dtype
        dy = to_nhwc(dy, dtype, dense=True)
        B, C, H, W = dy.shape
        dx = nhwc_empty(B, C, H, W, dtype, dy.device)
        L.maxpool_bwd(code(dtype), dy.data_ptr(), dy.stride(3), arg.data_ptr(), dx.data_ptr(), C, B, H, W, C, ctx.k, stream())
        return dx, None


class Upsample2xFn(torch.autograd.Function):
    """nn.Upsample(None, 2, 'nearest')"""

    @staticmethod
    def forward(ctx, x):
        L = lib()
        dtype = _COMPUTE_DTYPE
        x = to_nhwc(x, dtype)
        B, C, H, W = x.shape
        y = out_tensor(B, C, 2 * H, 2 * W, dtype, x.device)
        sb, sh, sw = s3(x)
        L.upsample2x_fwd(code(dtype), x.data_ptr(), sb, sh, sw, y.data_ptr(), y.stride(3), B, H, W, C, stream())
        ctx.dtype = dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        L = lib()
        dtype = ctx.dtype
        dy = to_nhwc(dy, dtype)
        B, C, H2, W2 = dy.shape
        dx = nhwc_empty(B, C, H2 // 2, W2 // 2, dtype, dy.device)
        sb, sh, sw = s3(dy)
        L.upsample2x_bwd(code(dtype), dy.data_ptr(), sb, sh, sw, dx.data_ptr(), C, B, H2 // 2, W2 // 2, C, stream())
        return dx


class AttentionFn(torch.autograd.Function):
    """softmax(q^T k * scale) applied to v, per head, on the NHWC qkv tensor; also returns v re-laid-out as
    (B, nh*hd, H, W) for the `pe` branch (reference block.py:785-797)."""

    @staticmethod
    def forward(ctx, qkv, nh, kd, hd, scale):
        L = lib()
        dtype = _COMPUTE_DTYPE
        dt = code(dtype)
        st = stream()
        qkv = to_nhwc(qkv, dtype, dense=True)
        B, Ct, H, W = qkv.shape
        N = H * W
        C = nh * hd
        dev = qkv.device
        out = nhwc_empty(B, C, H, W, dtype, dev)
        v = nhwc_empty(B, C, H, W, dtype, dev)
        lse = _f32(B * nh * N, dev)
        L.attn_fwd(dt, qkv.data_ptr(), qkv.stride(3), out.data_ptr(), C, lse.data_ptr(), B, N, nh, kd, hd, scale, st)
        esz = qkv.element_size()
        for h in range(nh):
            L.copy2d(dt, qkv.data_ptr() + (h * (2 * kd + hd) + 2 * kd) * esz, qkv.stride(3), v.data_ptr() + h * hd * esz, C, B * N, hd, st)
        ctx.cfg = (nh, kd, hd, scale, dtype)
        ctx.save_for_backward(qkv, out, lse)
        return out, v

    @staticmethod
    def backward(ctx, dout, dv):
        L = lib()
        qkv, out, lse = ctx.saved_tensors
        nh, kd, hd, scale, dtype = ctx.cfg
        dt = code(dtype)
        B, Ct, H, W = qkv.shape
        N = H * W
        C = nh * hd
        dev = qkv.device
        if dout is None:
            dout = torch.zeros_like(out)
        dout = to_nhwc(dout, dtype, dense=True)
        dvp, dvs = None, 0
        if dv is not None:
            dv = to_nhwc(dv, dtype, dense=True)
            dvp, dvs = dv.data_ptr(), dv.stride(3)
        dqkv = nhwc_empty(B, Ct, H, W, dtype, dev)
        delta = _f32(B * nh * N, dev)
        L.attn_bwd(dt, qkv.data_ptr(), qkv.stride(3), out.data_ptr(), C, dout.data_ptr(), dout.stride(3), dvp, dvs, lse.data_ptr(),
                   delta.data_ptr(), dqkv.data_ptr(), Ct, B, N, nh, kd, hd, scale, stream())
        return dqkv, None, None, None, None
