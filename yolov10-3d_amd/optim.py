"""Fused optimizer step: gradient clipping + SGD(Nesterov) or AdamW over all parameters in three HIP launches, + ModelEMA in one.

Reference recipe (engine/trainer.py:567-575): unscale -> clip_grad_norm_(max_norm=10) -> optimizer.step -> zero_grad -> ema.update,
with the three parameter groups of build_optimizer (:734-790: weights with decay, norm weights without, biases without; SGD with
nesterov for long schedules, AdamW(betas=(momentum, 0.999)) for short ones).  Semantics are torch.optim.SGD's (dampening 0),
torch.optim.AdamW's (amsgrad off), torch.nn.utils.clip_grad_norm_'s and utils/torch_utils.py:416-443's ModelEMA.
"""
from __future__ import annotations

import copy
import math

import torch

from . import ops
from ._lib import Y3DError, lib

CHUNK = 16384


class PtrUploader:
    """Pointer tables that change every step (the step's gradient tensors) go to the device through rotating PINNED host buffers
    with a non-blocking copy: `torch.tensor(list, device=...)` stages through pageable memory and holds the host until the stream
    has drained, which leaves the GPU idle while the rest of the optimizer's host code runs (measured ~1 ms per step)."""

    def __init__(self, n, device, depth=4):
        import numpy as np
        self.host = [torch.empty(n, dtype=torch.int64).pin_memory() for _ in range(depth)]
        self.np = [h.numpy() for h in self.host]
        self.dev = [torch.empty(n, dtype=torch.int64, device=device) for _ in range(depth)]
        self.i, self.n = 0, n
        self._np = np

    def upload(self, ptrs):
        assert len(ptrs) == self.n
        k = self.i
        self.i = (self.i + 1) % len(self.host)
        self.np[k][:] = self._np.asarray(ptrs, dtype=self._np.int64)
        self.dev[k].copy_(self.host[k], non_blocking=True)
        return self.dev[k]


class FusedSGD:
    NSTATE = 1  # flat zero-initialised state buffers per parameter (SGD: momentum)

    def __init__(self, param_groups, lr=0.01, momentum=0.937, nesterov=True, weight_decay=0.0):
        if isinstance(param_groups, (list, tuple)) and param_groups and not isinstance(param_groups[0], dict):
            param_groups = [{"params": list(param_groups)}]
        self.param_groups = []
        for g in param_groups:
            g = dict(g)
            g.setdefault("lr", lr)
            g.setdefault("weight_decay", weight_decay)
            g["params"] = [p for p in g["params"] if p.requires_grad]
            self.param_groups.append(g)
        self.momentum, self.nesterov = float(momentum), bool(nesterov)
        self.params = [p for g in self.param_groups for p in g["params"]]
        if not self.params:
            raise ValueError("FusedSGD: no parameters")
        self._state = None
        self._steps = 0
        self.last_norm = None
        self._pending_load = None

    def add_param_group(self, g):
        g = dict(g)
        g.setdefault("lr", self.param_groups[0]["lr"])
        g.setdefault("weight_decay", 0.0)
        g["params"] = [p for p in g["params"] if p.requires_grad]
        self.param_groups.append(g)
        self.params = [p for gg in self.param_groups for p in gg["params"]]
        self._state = None

    def _build(self, active):
        """tables over the parameters that carry a gradient (torch.optim.SGD and clip_grad_norm_ skip the others, e.g. the neck
        blocks behind a detect level that `num_scales: 2` leaves unused); momentum buffers are zero-initialised, so the first
        update `buf = momentum*0 + g` equals torch's `buf = clone(g)` exactly."""
        dev = self.params[0].device
        if dev.type != "cuda":
            raise Y3DError("FusedSGD needs parameters on a HIP device")
        for p in self.params:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise Y3DError("FusedSGD: parameters must be contiguous fp32 tensors")
        old = self._state
        if old is None:
            sizes_all = [p.numel() for p in self.params]
            flat = torch.zeros(self.NSTATE, sum(sizes_all), dtype=torch.float32, device=dev)  # state buffers, one allocation
            bufs, off = [], 0
            for n in sizes_all:
                bufs.append(flat[:, off:off + n])
                off += n
        else:
            flat, bufs = old["flat"], old["bufs"]
        sizes = [self.params[i].numel() for i in active]
        ct, co = [], []
        for t, n in enumerate(sizes):
            for c in range((n + CHUNK - 1) // CHUNK):
                ct.append(t)
                co.append(c)
        i64 = lambda v: torch.tensor(v, dtype=torch.int64, device=dev)
        lr_all = [g["lr"] for g in self.param_groups for _ in g["params"]]
        wd_all = [g["weight_decay"] for g in self.param_groups for _ in g["params"]]
        st = {
            "dev": dev, "flat": flat, "bufs": bufs, "nchunks": len(ct), "active": list(active),
            "sizes": i64(sizes), "ctensor": torch.tensor(ct, dtype=torch.int32, device=dev), "coff": torch.tensor(co, dtype=torch.int32, device=dev),
            "bptr": i64([bufs[i][0].data_ptr() for i in active]),
            "bptr2": i64([bufs[i][1].data_ptr() for i in active]) if self.NSTATE > 1 else None, "partials": torch.empty(len(ct), dtype=torch.float32, device=dev),
            "norm_clip": old["norm_clip"] if old is not None else torch.zeros(5, dtype=torch.float32, device=dev), "pkey": None, "gkey": None,
            "lr": torch.tensor([lr_all[i] for i in active], dtype=torch.float32, device=dev),
            "wd": torch.tensor([wd_all[i] for i in active], dtype=torch.float32, device=dev),
            "hkey": [(g["lr"], g["weight_decay"], len(g["params"])) for g in self.param_groups],
        }
        self._state = st
        if self._pending_load is not None:  # load_state_dict() before the first step
            sd, self._pending_load = self._pending_load, None
            self.load_state_dict(sd)
        return st

    def state_dict(self):
        """optimizer state for checkpoint / resume (the reference saves `optimizer.state_dict()`, engine/trainer.py:470-486): the flat
        state buffers in parameter order (SGD: momentum; AdamW: exp_avg, exp_avg_sq) and the device-side counters [norm, clip
        coefficient, finite flag, steps APPLIED, steps skipped] - `steps applied` is AdamW's bias-correction step count"""
        st = self._state
        return {"steps": self._steps, "flat": None if st is None else st["flat"].detach().clone(),
                "norm_clip": None if st is None else st["norm_clip"].detach().clone()}

    def load_state_dict(self, sd):
        """inverse of state_dict(); before the first step the state is kept and applied when the tables are built"""
        if self._state is None:
            self._pending_load = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in sd.items()}
            self._steps = int(sd["steps"])
            return
        st = self._state
        self._steps = int(sd["steps"])
        if sd.get("flat") is not None:
            if sd["flat"].shape != st["flat"].shape:
                raise Y3DError(f"optimizer state of shape {tuple(sd['flat'].shape)} does not fit these parameters {tuple(st['flat'].shape)}")
            st["flat"].copy_(sd["flat"])
            st["norm_clip"].copy_(sd["norm_clip"])

    def set_hyper(self):
        """write the groups' current lr / weight_decay into the EXISTING device tables (in place: a captured hipGraph of the step keeps
        reading them, graph.GraphedTrainStep) - what a scheduler's `param_group["lr"] = ...` needs before the next replay"""
        st = self._state
        if st is None:
            return
        lr_all = [g["lr"] for g in self.param_groups for _ in g["params"]]
        wd_all = [g["weight_decay"] for g in self.param_groups for _ in g["params"]]
        st["lr"].copy_(torch.tensor([lr_all[i] for i in st["active"]], dtype=torch.float32), non_blocking=False)
        st["wd"].copy_(torch.tensor([wd_all[i] for i in st["active"]], dtype=torch.float32), non_blocking=False)
        st["hkey"] = [(g["lr"], g["weight_decay"], len(g["params"])) for g in self.param_groups]

    def _tables(self):
        active = [i for i, p in enumerate(self.params) if p.grad is not None]
        if not active:
            raise Y3DError("FusedSGD.step: no parameter has a gradient")
        st = self._state
        hkey = [(g["lr"], g["weight_decay"], len(g["params"])) for g in self.param_groups]
        if st is not None and st["active"] == active and st["hkey"] != hkey and [h[2] for h in st["hkey"]] == [h[2] for h in hkey]:
            self.set_hyper()  # only the values moved (a scheduler): same tables, new contents
        if st is None or st["active"] != active or st["hkey"] != hkey:
            st = self._build(active)
        pkey = [self.params[i].data_ptr() for i in active]
        if pkey != st["pkey"]:  # parameters re-pointed (e.g. sibling-branch stacking, .to()): refresh
            st["pptr"] = torch.tensor(pkey, dtype=torch.int64, device=st["dev"])
            st["pkey"] = pkey
        gkey = []
        for i in active:
            p = self.params[i]
            g = p.grad
            if g.dtype != torch.float32 or not g.is_contiguous():
                g = p.grad = g.float().contiguous()
            gkey.append(g.data_ptr())
        if gkey != st["gkey"]:
            if st.get("gup") is None or st["gup"].n != len(gkey):
                st["gup"] = PtrUploader(len(gkey), st["dev"])
            st["gptr"] = st["gup"].upload(gkey)
            st["gkey"] = gkey
        return st

    @torch.no_grad()
    def step(self, max_norm: float | None = 10.0):
        """clip (when max_norm is given) + update; self.last_norm holds the device tensor [total_norm, clip_coef, finite flag, steps
        applied, steps skipped].  A step whose gradient norm is inf / NaN is skipped ON THE DEVICE (no host synchronisation): parameters,
        momentum / Adam state and AdamW's bias-correction step count stay - torch.cuda.amp.GradScaler.step's rule, engine/trainer.py:
        567-572.  Under ddp.FlatGradReducer the norm is that of the all-reduced buffer, so every rank takes the same decision."""
        st = self._tables()
        L, s = lib(), ops.stream()
        clip = None
        if max_norm is not None:
            L.mt_sqnorm(st["gptr"].data_ptr(), st["sizes"].data_ptr(), st["ctensor"].data_ptr(), st["coff"].data_ptr(), st["nchunks"], CHUNK,
                        st["partials"].data_ptr(), s)
            L.mt_clip_coef(st["partials"].data_ptr(), st["nchunks"], float(max_norm), st["norm_clip"].data_ptr(), s)
            clip = st["norm_clip"].data_ptr()
            self.last_norm = st["norm_clip"]
        self._steps += 1
        self._update(L, st, clip, s)
        ops.bump_weight_epoch()  # the parameters changed behind torch's version counters

    def _update(self, L, st, clip, s):
        L.mt_sgd(st["pptr"].data_ptr(), st["gptr"].data_ptr(), st["bptr"].data_ptr(), st["sizes"].data_ptr(), st["lr"].data_ptr(),
                 st["wd"].data_ptr(), st["ctensor"].data_ptr(), st["coff"].data_ptr(), st["nchunks"], CHUNK, self.momentum, int(self.nesterov),
                 0, clip, s)

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()


class FusedAdamW(FusedSGD):
    """torch.optim.AdamW(betas, eps, weight_decay per group), amsgrad off; same tables and clipping as FusedSGD"""
    NSTATE = 2  # exp_avg, exp_avg_sq

    def __init__(self, param_groups, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(param_groups, lr=lr, momentum=betas[0], nesterov=False, weight_decay=weight_decay)
        self.betas, self.eps = (float(betas[0]), float(betas[1])), float(eps)

    def step(self, max_norm: float | None = 10.0):
        """as FusedSGD.step; without a `max_norm` the clipping launch still runs (with an infinite bound: coefficient 1), because the
        bias corrections read the number of APPLIED steps from its device-side counter - ONE source for the step count, whether or not
        the step is clipped, eager or replayed from a hipGraph, fresh or resumed (round-3 advisor finding).  Side effect: a step with a
        non-finite gradient norm is skipped in that case too."""
        return super().step(max_norm=float("inf") if max_norm is None else max_norm)

    def _update(self, L, st, clip, s):
        b1, b2 = self.betas
        bc1 = 1.0 - b1 ** max(self._steps, 1)  # (placeholders: the kernel recomputes both from the device-side step count)
        bc2s = math.sqrt(1.0 - b2 ** max(self._steps, 1))
        L.mt_adamw(st["pptr"].data_ptr(), st["gptr"].data_ptr(), st["bptr"].data_ptr(), st["bptr2"].data_ptr(), st["sizes"].data_ptr(),
                   st["lr"].data_ptr(), st["wd"].data_ptr(), st["ctensor"].data_ptr(), st["coff"].data_ptr(), st["nchunks"], CHUNK, b1, b2,
                   self.eps, bc1, bc2s, clip, s)


class ModelEMA:
    """utils/torch_utils.py:416-443: exponential moving average of every floating-point state_dict tensor (parameters AND
    BatchNorm running statistics), decay ramp `decay * (1 - exp(-updates / tau))`, one multi-tensor launch per update.

    The reference walks the EMA model's state_dict KEYS; the one-to-one head branches are registered under two names each
    (`o2o_heads.*` and `cls/o2d/...`, nn/modules/head.py:627-629), so those tensors receive the update twice per call.  The table
    keeps that multiplicity (`reps`), because matching the reference's EMA weights bit for bit needs it."""

    def __init__(self, model, decay=0.9999, tau=2000, updates=0):
        self.ema = copy.deepcopy(model).eval()
        self.updates = updates
        self.decay = lambda x: decay * (1 - math.exp(-x / tau))
        for p in self.ema.parameters():
            p.requires_grad_(False)
        self.enabled = True
        self._tab = None

    def _tables(self, model):
        esd = self.ema.state_dict(keep_vars=True)   # keep_vars: the live tensors, so that re-pointed storage is noticed
        msd = model.state_dict(keep_vars=True)
        pairs, index = [], {}
        for k, v in esd.items():
            if not v.dtype.is_floating_point:
                continue
            m = msd[k]
            if v.dtype != torch.float32 or m.dtype != torch.float32 or not (v.is_contiguous() and m.is_contiguous()):
                raise Y3DError(f"ModelEMA: {k} must be a contiguous fp32 tensor on both models")
            if v.shape != m.shape:
                raise Y3DError(f"ModelEMA: {k} has shape {tuple(v.shape)} in the EMA model and {tuple(m.shape)} in the model")
            j = index.get(v.data_ptr())
            if j is None:
                index[v.data_ptr()] = len(pairs)
                pairs.append([v, m, 1])
            else:
                pairs[j][2] += 1
        dev = pairs[0][0].device
        if dev.type != "cuda":
            raise Y3DError("ModelEMA needs the models on a HIP device")
        ct, co = [], []
        for t, (v, _, _) in enumerate(pairs):
            for c in range((v.numel() + CHUNK - 1) // CHUNK):
                ct.append(t)
                co.append(c)
        i64 = lambda x: torch.tensor(x, dtype=torch.int64, device=dev)
        i32 = lambda x: torch.tensor(x, dtype=torch.int32, device=dev)
        self._tab = {"pairs": pairs, "n": len(ct), "sizes": i64([v.numel() for v, _, _ in pairs]), "reps": i32([r for _, _, r in pairs]),
                     "ct": i32(ct), "co": i32(co), "ekey": None, "mkey": None, "model": model}
        return self._tab

    @torch.no_grad()
    def update(self, model, guard=None):
        """guard: FusedSGD / FusedAdamW `.last_norm` of the optimizer step this update follows - the update is then skipped on the
        device when that step was (the reference's line runs unconditionally, engine/trainer.py:574-575, and re-applies the decay to
        an unchanged model; guard=None keeps that arithmetic)"""
        if not self.enabled:
            return
        self.updates += 1
        d = self.decay(self.updates)
        tb = self._tab
        if tb is None or tb["model"] is not model:
            tb = self._tables(model)
        ekey = [v.data_ptr() for v, _, _ in tb["pairs"]]
        mkey = [m.data_ptr() for _, m, _ in tb["pairs"]]
        if len(set(ekey)) != len(ekey):  # storage re-pointed so that entries merged or split: rebuild the multiplicities
            tb = self._tables(model)
            ekey = [v.data_ptr() for v, _, _ in tb["pairs"]]
            mkey = [m.data_ptr() for _, m, _ in tb["pairs"]]
        dev = tb["sizes"].device
        if ekey != tb["ekey"]:
            tb["eptr"], tb["ekey"] = torch.tensor(ekey, dtype=torch.int64, device=dev), ekey
        if mkey != tb["mkey"]:
            tb["mptr"], tb["mkey"] = torch.tensor(mkey, dtype=torch.int64, device=dev), mkey
        ops.bump_param_epoch()
        lib().mt_ema(tb["eptr"].data_ptr(), tb["mptr"].data_ptr(), tb["sizes"].data_ptr(), tb["reps"].data_ptr(), tb["ct"].data_ptr(),
                     tb["co"].data_ptr(), tb["n"], CHUNK, float(d), float(1 - d), guard.data_ptr() if guard is not None else None, ops.stream())

    def update_attr(self, model, include=(), exclude=("process_group", "reducer")):
        """utils/torch_utils.py:445-448 / copy_attr :342-349"""
        if self.enabled:
            for k, v in model.__dict__.items():
                if (len(include) and k not in include) or k.startswith("_") or k in exclude:
                    continue
                setattr(self.ema, k, v)


def build_optimizer(model, lr=0.01, momentum=0.937, decay=5e-4, name="SGD"):
    """the reference's three parameter groups (engine/trainer.py:766-790): biases, weights (decay), normalisation weights;
    name: "SGD" (nesterov) or "AdamW" (betas = (momentum, 0.999)), as :775-781"""
    g = [], [], []
    bn = tuple(v for k, v in torch.nn.__dict__.items() if "Norm" in k)
    seen = set()
    for mod in model.modules():
        for pn, p in mod.named_parameters(recurse=False):
            if id(p) in seen or not p.requires_grad:
                continue
            seen.add(id(p))
            if pn == "bias":
                g[2].append(p)
            elif isinstance(mod, bn):
                g[1].append(p)
            else:
                g[0].append(p)
    groups = [{"params": g[2], "weight_decay": 0.0}, {"params": g[0], "weight_decay": decay}, {"params": g[1], "weight_decay": 0.0}]
    if name == "AdamW":
        return FusedAdamW(groups, lr=lr, betas=(momentum, 0.999), weight_decay=0.0)
    if name != "SGD":
        raise NotImplementedError(f"optimizer {name!r}: the hot path provides SGD and AdamW (engine/trainer.py:775-785)")
    return FusedSGD(groups, lr=lr, momentum=momentum, nesterov=True)
