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
