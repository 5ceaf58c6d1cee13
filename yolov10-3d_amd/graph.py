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

    N > 1: pass the `ddp.FlatGradReducer` - its bucket gathers and RCCL all-reduces are recorded into the graph (on the capturing stream:
    no side stream is forked inside a capture), every rank captures and replays the same sequence of collectives.
    Before constructing one, DROP every reference to an earlier step's autograd graph (`del loss`): a live graph keeps its
    AccumulateGrad nodes, which belong to the stream they were made on - the NULL stream for an eager step - and a capture that has to
    touch that stream dies in hipStreamEndCapture (torch warns "AccumulateGrad node's stream does not match"; seen as a segfault).

    Constructing one does NOT train: the warm-up steps it needs run on the first batch and are undone (parameters, BatchNorm buffers,
    optimizer state and step counts are restored in place before the capture).

    Static shapes: the per-box label tensors (`ddp.PER_BOX_KEYS` + `batch_idx`) are padded to `label_capacity` rows (default: 64 per
    image, the assigner's own limit); padding rows carry batch_idx = -1, which no image matches.  `step(batch)` copies the batch into
    the static buffers and replays.  Left to the caller, eagerly, after the replay: `ema.update` (its decay ramp is a host-side
    function of the update count) and learning-rate changes through `opt.set_hyper` (in-place writes of the device tables).
    After every replay the host-side bookkeeping the captured Python code would have done is redone: BatchNorm `num_batches_tracked`
    counters and the parameter epochs that the eval caches / weight packs key on."""

    def __init__(self, model, opt, batch, max_norm: float | None = 10.0, label_capacity: int | None = None, warmup: int = 2, reducer=None):
        from . import loss as _loss
        from .ddp import PER_BOX_KEYS
        from .modules import Conv
        from .optim import PtrUploader
        self.model, self.opt, self.max_norm, self.reducer = model, opt, max_norm, reducer
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
        # The warm-up below runs REAL steps (the weight packs build their pointer / chunk tables on the second one, with a host-to-device
        # copy that a capture does not allow; the optimizer builds its tables; the allocator warms up).  They must not train: parameters,
        # BatchNorm buffers and counters, optimizer state and step counts are snapshotted here and put back (in place: every address
        # stays) before the capture, so constructing a GraphedTrainStep leaves model and optimizer exactly as it found them.
        for m in model.modules():  # the fused head re-points its branch parameters / BatchNorm buffers at stacked storage on its first
            if hasattr(m, "restack"):  # forward: do it now, so that the snapshot below holds the tensors the model will keep
                m.restack()
        with torch.no_grad():
            seen, tensors = set(), []
            for t in list(model.parameters()) + list(model.buffers()):
                if t.data_ptr() not in seen:
                    seen.add(t.data_ptr())
                    tensors.append(t)
            snap = [t.detach().clone() for t in tensors]
        nbt = [m._nbt_pending for m in self.convs]
        had_state = opt._state is not None
        opt_snap = (opt._state["flat"].clone(), opt._state["norm_clip"].clone()) if had_state else None
        steps0 = opt._steps
        if reducer is not None:
            # the reducer's gather tables (source pointers through rotating pinned buffers): the warm-up below builds fresh ones, the capture
            # bakes their addresses in, and they are then set aside for the graph alone (eager steps build their own) - as the optimizer's
            reducer._tabs = {}
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(max(2, warmup)):
                self._body()
                opt.zero_grad(set_to_none=True)  # (under a reducer: p.grad was the reduced slot; the next backward starts from None)
            with torch.no_grad():
                for t, v in zip(tensors, snap):
                    t.copy_(v)
                if had_state:
                    opt._state["flat"].copy_(opt_snap[0])
                    opt._state["norm_clip"].copy_(opt_snap[1])
                else:
                    opt._state["flat"].zero_()
                    opt._state["norm_clip"].zero_()
        opt._steps = steps0
        for m, n in zip(self.convs, nbt):
            m._nbt_pending = n
        cur.wait_stream(side)
        torch.cuda.synchronize()
        ops.bump_weight_epoch()  # the captured forward must contain the weight (re)packing launches of a fresh step
        # the gradient pointer table of the captured optimizer launches must not live in the optimizer's ROTATING pinned buffers: a few
        # eager opt.step() calls later (a fallback for an over-capacity batch) would overwrite them and the next replay would read those
        # pointers (round-3 advisor finding).  The graph gets an uploader of its own; the optimizer builds a new one when it next needs it.
        st = opt._state
        st["gup"], st["gkey"] = PtrUploader(len(st["active"]), st["dev"], depth=1), None
        _loss._CAPTURED_COUNTS.clear()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.loss, self.items = self._body()
        if reducer is not None:
            self._red_tabs, reducer._tabs = reducer._tabs, {}  # kept alive for the replays; eager steps build new ones
        self._gup = st["gup"]  # kept alive: the graph's memcpy node reads its pinned buffer at every replay
        st["gup"], st["gkey"] = None, None
        # every device table the captured optimizer launches read (sizes, chunk maps, lr / wd, parameter and state pointers): the optimizer
        # REPLACES them when its active set or a parameter address changes (an eager step in between); the graph keeps reading these
        self._opt_tables = dict(st)
        opt._steps = steps0  # the captured Python counted a step that has not run
        # the device words [min(count, cap), true largest per-image count] of the captured pad_targets launches: graph-private static
        # memory, rewritten by every replay; read back after each one so that an image with more boxes than the kernels take is
        # reported as in the eager loop, not trained on silently truncated targets (round-3 advisor finding)
        self.counts = list(_loss._CAPTURED_COUNTS)
        _loss._CAPTURED_COUNTS.clear()
        for m in self.convs:  # the captured forward counted a step that has not run
            if m.training and m._nbt_pending > 0:
                m._nbt_pending -= 1

    def _body(self):
        loss, items = self.model(self.static)
        loss.backward()
        if self.reducer is not None:
            # N > 1 (ddp.FlatGradReducer): the bucket gathers and RCCL all-reduces that the gradient hooks launched on the reducer's side
            # stream during the backward above, and the join below, are part of the captured graph - a forked branch beside the body
            # backward, exactly the eager overlap - so a replay runs the whole data-parallel step with one host call
            self.reducer.finish()
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
        from . import loss as _loss
        _loss.check_target_overflow()  # raises Y3DError a step or two after a replay met an over-capacity image (no host sync)
        if batch is not None:
            self._load(batch)
        self.graph.replay()
        for n_used, cap in self.counts:
            _loss.watch_target_count(n_used, cap)
        self._after()
        return self.loss, self.items
