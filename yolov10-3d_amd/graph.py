"""hipGraph replay of launch-bound forwards.

The eval forward + NMS-free postprocess of YOLOv10-S-3D is ~200 kernel launches of 3-150 us each: enqueued one by one from Python the
host needs as long as the GPU (4.9 ms per batch of 32 for 4.8 ms of kernels, tools/eval_audit.py), so every kernel-side saving is
invisible.  Every entry point of liby3d_hip.so takes the stream it launches on and neither allocates nor synchronises, and the
torch-side buffers come from torch's graph-private memory pool during capture, so the whole forward records into ONE hipGraph
(`torch.cuda.graph`) and replays with a single host call.

Reference counterpart: none (the reference's validator launches eagerly, engine/validator.py:160-189); what is replayed is exactly
that loop body - `model(img)` + `v10_3Dpostprocess` - on a static input buffer.
"""
from __future__ import annotations

import torch

from . import ops
from ._lib import Y3DError


class GraphedForward:
    """`g = GraphedForward(fn, *example_inputs); out = g(*inputs)`

    `fn(*tensors) -> tensor | tuple | list | dict of tensors` must launch the same kernels on the same shapes every call (no data-
    dependent host branches, no host synchronisation): the eval forward of the model classes of tasks.py and the postprocess
    functions of loss.py qualify.  Inputs are copied into static buffers (skipped when the caller passes the static buffer itself,
    `g.inputs[i]`), outputs are the static output tensors of the capture - valid until the next call.

    The capture embeds the addresses of the packed weights and folded BatchNorm constants of the eval caches (ops._eval_consts), so it
    is tied to the parameter state: when a raw-pointer writer (optimizer step, EMA update, bn_finalize of a training forward) has
    moved `ops.PARAM_EPOCH`, or an input shape / dtype changed, the next call re-captures."""

    def __init__(self, fn, *example_inputs, warmup: int = 2):
        if not example_inputs or not all(torch.is_tensor(t) and t.is_cuda for t in example_inputs):
            raise Y3DError("GraphedForward needs HIP device tensors as example inputs")
        self.fn, self.warmup = fn, warmup
        self.captures = 0
        self._capture(example_inputs)

    def _capture(self, inputs):
        self.inputs = [t.detach().clone() for t in inputs]
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(self.warmup):  # fills the eval caches (packed weights, folded BatchNorm): a steady-state forward is captured
                self.fn(*self.inputs)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.outputs = self.fn(*self.inputs)
        self.epoch = ops.PARAM_EPOCH
        self.sig = [(tuple(t.shape), t.dtype) for t in inputs]
        self.captures += 1

    def __call__(self, *inputs):
        if len(inputs) != len(self.inputs):
            raise Y3DError(f"GraphedForward: {len(self.inputs)} inputs were captured, {len(inputs)} given")
        if ops.PARAM_EPOCH != self.epoch or [(tuple(t.shape), t.dtype) for t in inputs] != self.sig:
            self._capture(inputs)
        for s, t in zip(self.inputs, inputs):
            if s is not t:
                s.copy_(t, non_blocking=True)
        self.graph.replay()
        return self.outputs


class GraphedTrainStep:
    """One training step - forward + dual-assignment loss + backward + clip_grad_norm_ + optimizer step (engine/trainer.py:395-402,
    567-572) - as ONE hipGraph: `step = GraphedTrainStep(model, opt, batch); loss, items = step(batch)`.

    For the small models the step is bound by the host (N-3D: ~900 launches enqueued in 17 ms for ~12 ms of kernels); replayed from a
    graph the host cost is one call.  What makes the step capturable: the library never synchronises or allocates; target padding
    keeps its count on the device (loss.pad_targets); the optimizer's skip of non-finite steps and AdamW's step count live on the
    device (csrc/optim.hip); pointer tables travel through pinned buffers (optim.PtrUploader).

    Static shapes: the per-box label tensors (`ddp.PER_BOX_KEYS` + `batch_idx`) are padded to `label_capacity` rows (default: 64 per
    image, the assigner's own limit); padding rows carry batch_idx = -1, which no image matches.  `step(batch)` copies the batch into
    the static buffers and replays.  Left to the caller, eagerly, after the replay: `ema.update` (its decay ramp is a host-side
    function of the update count) and learning-rate changes through `opt.set_hyper` (in-place writes of the device tables).
    After every replay the host-side bookkeeping the captured Python code would have done is redone: BatchNorm `num_batches_tracked`
    counters and the parameter epochs that the eval caches / weight packs key on."""

    def __init__(self, model, opt, batch, max_norm: float | None = 10.0, label_capacity: int | None = None, warmup: int = 2):
        from .ddp import PER_BOX_KEYS
        from .modules import Conv
        self.model, self.opt, self.max_norm = model, opt, max_norm
        self.box_keys = tuple(k for k in (("batch_idx",) + PER_BOX_KEYS) if k in batch)
        B = batch["img"].shape[0]
        self.cap = int(label_capacity or 64 * B)
        self.static = {}
        for k, v in batch.items():
            if not torch.is_tensor(v):
                self.static[k] = v
            elif k in self.box_keys:
                t = torch.zeros((self.cap,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
                if k == "batch_idx":
                    t.fill_(-1)
                self.static[k] = t
            else:
                self.static[k] = v.detach().clone()
        self.convs = [m for m in model.modules() if isinstance(m, Conv)]
        self._load(batch)
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(max(2, warmup)):  # eager steps on the static batch: optimizer state, pointer / chunk tables of the weight
                # packs (built on the SECOND step, with a host-to-device copy that a capture does not allow), allocator warm
                self._body()
                opt.zero_grad(set_to_none=True)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        ops.bump_weight_epoch()  # the captured forward must contain the weight (re)packing launches of a fresh step
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.loss, self.items = self._body()
        for m in self.convs:  # the captured forward counted a step that has not run
            if m.training and m._nbt_pending > 0:
                m._nbt_pending -= 1

    def _body(self):
        loss, items = self.model(self.static)
        loss.backward()
        self.opt.step(max_norm=self.max_norm)
        return loss.detach(), items

    def _after(self):
        for m in self.convs:
            if m.training:
                m._nbt_pending += 1
        ops.bump_weight_epoch()

    def _load(self, batch):
        n = batch["batch_idx"].shape[0]
        if n > self.cap:
            raise Y3DError(f"GraphedTrainStep: {n} boxes in the batch, label capacity {self.cap}")
        for k, s in self.static.items():
            v = batch[k]
            if not torch.is_tensor(v) or v is s:
                continue
            if k in self.box_keys:
                s[:n].copy_(v, non_blocking=True)
                if k == "batch_idx":
                    s[n:].fill_(-1)
            else:
                if v.shape != s.shape:
                    raise Y3DError(f"GraphedTrainStep: batch entry {k!r} changed shape {tuple(s.shape)} -> {tuple(v.shape)}")
                s.copy_(v, non_blocking=True)

    def __call__(self, batch=None):
        if batch is not None:
            self._load(batch)
        self.graph.replay()
        self._after()
        return self.loss, self.items
