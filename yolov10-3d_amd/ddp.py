"""Image-parallel data parallelism of the hot path: one process per GPU, gradients all-reduced over RCCL/xGMI.

Reference behaviour restated (engine/trainer.py:225-236, 280, 292, 401-402; utils/dist.py:55-65):
  * the global batch is split evenly by rank (`batch // world_size` images per process), every rank runs the whole
    model on its shard with its OWN BatchNorm statistics (no SyncBN) and its own assigner / loss;
  * the loss is `loss.sum() * local_batch` (utils/loss.py:900) and is multiplied by `world_size` before backward because the
    all-reduce AVERAGES gradients — so the update equals the single-process update on the global batch;
  * the only collective on the data path is that gradient all-reduce (fp32, S-3D: 120 MB per step), which the reference's
    DistributedDataParallel overlaps with the backward pass bucket by bucket (trainer.py:280).

torch.distributed is the plumbing (backend "nccl" == RCCL on ROCm, "gloo" for the CPU tests).  Two reducers: `FlatGradReducer`
(bench.py's N>1 path: the flat fp32 gradient buffer laid out head-first and all-reduced bucket by bucket on a side stream while
the backward of the earlier layers is still running) and `wrap` (torch DistributedDataParallel with buckets as views, kept as the
drop-in for code that expects a DDP module).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

# keys of the collated batch dict (reference data/datasets/kitti.py:421-442, collate_fn :579-599)
PER_BOX_KEYS = ("cls", "bboxes", "center_2d", "size_2d", "center_3d", "size_3d", "depth", "heading_bin", "heading_res")
# per-image entries: stacked tensors (`depth_map` is a (B, 1) placeholder unless depth maps are loaded, kitti.py:409-420) and the
# tuples of B python objects collate_fn leaves un-stacked (`ori_img`, `info`, `im_file`, `ori_shape`)
PER_IMAGE_KEYS = ("img", "calib", "mixed", "im_file", "ori_shape", "resized_shape", "ratio_pad", "info", "depth_map", "coord_range", "ori_img")
REPLICATED_KEYS = ("mean_sizes",)


def init(backend: str | None = None, device: torch.device | None = None):
    """Initialise the default process group from the torchrun environment (RANK / WORLD_SIZE / MASTER_*).  -> (rank, world)"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if (world > 1 or os.environ.get("Y3D_FORCE_DDP")) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29500")  # single-rank rehearsal without a launcher
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard_batch(batch: dict, rank: int, world: int) -> dict:
    """Rank's slice of a collated batch dict (schema: SURVEY §8b).  The split is decided BY KEY, never by comparing leading
    dimensions (a batch with as many boxes as images, or exactly `nc` boxes next to the (nc, 3) `mean_sizes`, is ambiguous by
    shape): per-image entries are split evenly, per-box entries follow their `batch_idx` (re-based to the local image range),
    `mean_sizes` is replicated.  Unknown tensor keys raise: silently mis-slicing a label tensor corrupts training."""
    if world == 1:
        return batch
    B = batch["img"].shape[0]
    assert B % world == 0, f"global batch {B} not divisible by world size {world}"
    per = B // world
    lo, hi = rank * per, (rank + 1) * per
    bi = batch["batch_idx"]
    sel = (bi >= lo) & (bi < hi)
    out = {}
    for k, v in batch.items():
        if k.startswith("_y3d"):
            continue  # per-step caches of the loss never travel between shards
        if k == "batch_idx":
            out[k] = v[sel] - lo
        elif k in PER_BOX_KEYS:
            assert v.shape[0] == bi.shape[0], f"shard_batch: per-box entry {k!r} has {v.shape[0]} rows for {bi.shape[0]} boxes"
            out[k] = v[sel]
        elif k in PER_IMAGE_KEYS:
            assert len(v) == B, f"shard_batch: per-image entry {k!r} has {len(v)} rows for {B} images"
            out[k] = v[lo:hi]
        elif k in REPLICATED_KEYS:
            out[k] = v
        elif not torch.is_tensor(v):
            # an un-stacked per-image sequence (collate_fn keeps non-tensor values as tuples of length B) is sliced; anything else
            # (scalars, strings, config objects) is replicated
            out[k] = v[lo:hi] if isinstance(v, (list, tuple)) and len(v) == B else v
        else:
            raise KeyError(f"shard_batch: do not know how to split batch entry {k!r} (add it to PER_BOX_KEYS / PER_IMAGE_KEYS / REPLICATED_KEYS)")
    return out


def wrap(model: torch.nn.Module, device_ids=None, bucket_cap_mb: float = 32.0):
    """DistributedDataParallel with gradient buckets as views (no extra copy) and a static bucket plan: every parameter of
    the YOLOv10(-3D) graph receives a gradient every step, so nothing is ever 'unused'."""
    return torch.nn.parallel.DistributedDataParallel(model, device_ids=device_ids, bucket_cap_mb=bucket_cap_mb,
                                                     gradient_as_bucket_view=True, static_graph=True)


class FlatGradReducer:
    """Gradient all-reduce over ONE flat fp32 buffer (S-3D: 120 MB), the data-path collective of the reference's DDP.

    Layout: parameters in REVERSE registration order, so the detect head (layer 23: 79 % of the bytes) comes first and
    the buffer fills front to back as the backward pass runs.  The buffer is cut into buckets of about `bucket_mb`; a post-
    accumulate hook on every parameter counts its bucket down, and as soon as a bucket AND all buckets before it are complete it is
    gathered (one multi-tensor launch, `y3d_mt_copy`) and all-reduced on a side stream, behind an event recorded on the compute
    stream at that point — the head buckets travel over xGMI while the backbone / neck backward is still computing.  Buckets are
    always launched in index order, so every rank issues the same collectives in the same order.  `finish()` (after backward)
    launches whatever is left (parameters without a gradient, e.g. an unused detect level, keep zero slots on every rank), makes
    the compute stream wait for the collectives, and re-points `p.grad` at the slots, so the fused optimizer runs on stable
    pointers.  SUM of the unscaled local losses' gradients equals the reference's `loss * world_size` + averaged gradients
    (trainer.py:401-402).
    Why not torch DDP: its autograd hooks copy each of the ~570 gradient tensors into a bucket view one by one (+17 % step time
    measured at one rank); the gather is one launch per bucket at HBM rate.
    `overlap=False` (env Y3D_DDP_OVERLAP=0): everything is gathered and reduced in `finish()`, after the backward.

    Contract: ONE `finish()` per optimizer step, gradients cleared with `zero_grad(set_to_none=True)` in between
    (after `finish()` `p.grad` IS the all-reduced slot: a backward that accumulated into it would be reduced a second time; the
    next launch raises when it sees that).  Gradient accumulation (reference: `accumulate = nbs / batch` micro-steps, trainer.py:383-386,
    DDP `no_sync`) runs the first micro-steps under `with reducer.no_sync():` - no hook counts, nothing is launched, autograd adds
    into `p.grad` as usual - and the last one outside it, followed by `finish()`.  A parameter without a gradient in this step has
    its slot ZEROED before the collective, whatever an earlier step left there (another rank may own a gradient for it) - and after
    `finish()` EVERY rank holds the reduced sum in `p.grad` for every parameter that has a gradient on some rank."""

    CHUNK = 16384
    MASK_EVERY = 256

    def __init__(self, params, bucket_mb: float = 32.0, overlap: bool | None = None, timing: bool = False):
        seen, plist = set(), []
        for p in params:
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                plist.append(p)
        if not plist:
            raise ValueError("FlatGradReducer: no parameters")
        self.params = plist[::-1]  # head first
        dev = self.params[0].device
        self.device = dev
        self.sizes = [p.numel() for p in self.params]
        self.flat = torch.zeros(sum(self.sizes), dtype=torch.float32, device=dev)
        self.views, self.offs, off = [], [], 0
        for p, n in zip(self.params, self.sizes):
            self.views.append(self.flat[off:off + n].view_as(p))
            self.offs.append(off)
            off += n
        # buckets: contiguous parameter ranges [i0, i1) of about bucket_mb
        cap = max(1, int(bucket_mb * (1 << 20) / 4))
        self.buckets, i0, acc = [], 0, 0
        for i, n in enumerate(self.sizes):
            acc += n
            if acc >= cap or i == len(self.sizes) - 1:
                self.buckets.append((i0, i + 1))
                i0, acc = i + 1, 0
        self.bucket_of = [0] * len(self.params)
        for b, (a, e) in enumerate(self.buckets):
            for i in range(a, e):
                self.bucket_of[i] = b
        if overlap is None:
            overlap = os.environ.get("Y3D_DDP_OVERLAP", "1") != "0"
        self.overlap = overlap
        self.timing = timing
        self.comm = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self._tabs = {}
        self._dirty = [False] * len(self.params)  # slot i holds something other than zeros
        self._reset()
        self._hooks = []
        if overlap:
            for i, p in enumerate(self.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        self.last_times = None
        self._sync = True
        self._any, self._nfinish = None, 0  # which parameters have a gradient on some rank (finish())
        # one-rank rehearsals (Y3D_FORCE_DDP=1: a process group of size 1 over RCCL) issue the collectives too, so that the whole N > 1
        # code path - including its capture into a hipGraph - runs on a one-GPU box
        self.always_collective = bool(os.environ.get("Y3D_FORCE_DDP"))

    def no_sync(self):
        """context manager for the non-final micro-steps of gradient accumulation (torch DDP's `no_sync`)"""
        import contextlib

        @contextlib.contextmanager
        def cm():
            prev, self._sync = self._sync, False
            try:
                yield self
            finally:
                self._sync = prev
        return cm()

    # ---- per-step state ------------------------------------------------------------------------------------------
    def _reset(self):
        self._pending = [e - a for a, e in self.buckets]
        self._ready = [False] * len(self.params)
        self._next = 0
        self._works = []
        self._events = []
        self._keep = []

    def _make_hook(self, i):
        def hook(p):
            if not self._sync:
                return  # accumulation micro-step: the gradient stays local in p.grad
            if self._ready[i]:
                return  # accumulated twice in one backward (shared parameter): the bucket was counted once
            self._ready[i] = True
            b = self.bucket_of[i]
            self._pending[b] -= 1
            while self._next < len(self.buckets) and self._pending[self._next] == 0:
                self._launch(self._next)
                self._next += 1
        return hook

    def broadcast_parameters(self, module: torch.nn.Module):
        """rank 0's parameters and buffers to every rank (what DistributedDataParallel does when it wraps a module)"""
        if dist.is_initialized() and dist.get_world_size() > 1:
            seen = set()
            for t in list(module.parameters()) + list(module.buffers()):
                if id(t) in seen:
                    continue
                seen.add(id(t))
                dist.broadcast(t.data, 0)

    # ---- gather + collective of one bucket -----------------------------------------------------------------------
    def _gather(self, idx, grads):
        """copy the gradient tensors `grads` of parameters `idx` into their slots (one HIP launch on the current stream)"""
        from ._lib import Y3DError, lib
        from . import ops
        dev = self.device
        if dev.type != "cuda":
            raise Y3DError("FlatGradReducer needs gradients on a HIP device")
        key = tuple(idx)
        tb = self._tabs.get(key)
        if tb is None:
            sizes = [self.sizes[i] for i in idx]
            ct, co = [], []
            for t, n in enumerate(sizes):
                for c in range((n + self.CHUNK - 1) // self.CHUNK):
                    ct.append(t)
                    co.append(c)
            from .optim import PtrUploader
            tb = {"sizes": torch.tensor(sizes, dtype=torch.int64, device=dev), "ct": torch.tensor(ct, dtype=torch.int32, device=dev),
                  "co": torch.tensor(co, dtype=torch.int32, device=dev), "n": len(ct),
                  "dst": torch.tensor([self.views[i].data_ptr() for i in idx], dtype=torch.int64, device=dev),
                  "up": PtrUploader(len(idx), dev)}
            self._tabs[key] = tb
        src = tb["up"].upload([g.data_ptr() for g in grads])  # non-blocking: the host must not wait for the backward to drain
        lib().mt_copy(src.data_ptr(), tb["dst"].data_ptr(), tb["sizes"].data_ptr(), tb["ct"].data_ptr(), tb["co"].data_ptr(), tb["n"], self.CHUNK,
                      1.0, ops.stream())

    def _launch(self, b):
        a, e = self.buckets[b]
        idx, grads, stale = [], [], []
        for i in range(a, e):
            g = self.params[i].grad
            if g is None:
                if self._dirty[i]:
                    stale.append(i)  # a gradient of an earlier step is still in the slot: this rank contributes zero now
                continue
            self._dirty[i] = True
            if g.data_ptr() == self.views[i].data_ptr():
                # p.grad is still the slot finish() re-pointed it at: this backward accumulated into an all-reduced sum, which the
                # collective below would reduce a second time
                raise RuntimeError("FlatGradReducer: a gradient was accumulated into its already all-reduced slot - clear gradients "
                                   "with zero_grad(set_to_none=True) after every finish(), and run the non-final micro-steps of "
                                   "gradient accumulation under reducer.no_sync()")
            if g.dtype != torch.float32 or not g.is_contiguous():
                g = g.float().contiguous()
            idx.append(i)
            grads.append(g)
        lo = self.offs[a]
        hi = self.offs[e - 1] + self.sizes[e - 1]
        multi = dist.is_initialized() and (dist.get_world_size() > 1 or self.always_collective)
        capturing = self.comm is not None and torch.cuda.is_current_stream_capturing()
        timing = self.timing and not capturing  # (timing events cannot be recorded into a hipGraph capture)
        if capturing:
            # inside a hipGraph capture (graph.GraphedTrainStep): gather + collective on the capturing stream itself - no side stream to
            # fork and join inside the capture; the graph's node order is the eager order of one stream
            for i in stale:
                self.views[i].zero_()
                self._dirty[i] = False
            if idx:
                self._gather(idx, grads)
            if multi:
                dist.all_reduce(self.flat[lo:hi])
            self._keep.append(grads)
        elif self.comm is not None:
            ev = torch.cuda.Event(enable_timing=timing)
            ev.record()  # compute stream: everything that produced this bucket's gradients has been enqueued
            self.comm.wait_event(ev)
            with torch.cuda.stream(self.comm):
                # the side stream is ONE timeline: gather(b), all-reduce(b), gather(b+1), ...  (with RCCL `wait()` only orders the
                # stream behind the collective; host-staged backends block the host here, which a rehearsal tolerates)
                for i in stale:
                    self.views[i].zero_()
                    self._dirty[i] = False
                if idx:
                    self._gather(idx, grads)
                t0 = t1 = None
                if timing:
                    t0 = torch.cuda.Event(enable_timing=True)
                    t0.record()
                if multi:
                    dist.all_reduce(self.flat[lo:hi], async_op=True).wait()
                if timing:
                    t1 = torch.cuda.Event(enable_timing=True)
                    t1.record()
                    self._events.append((ev, t0, t1, hi - lo))
            self._keep.append(grads)  # the sources stay alive until finish() has ordered the compute stream behind the copies
        else:
            for i in stale:
                self.views[i].zero_()
                self._dirty[i] = False
            if idx:
                self._gather(idx, grads)
            if multi:
                self._works.append(dist.all_reduce(self.flat[lo:hi], async_op=True))

    def finish(self):
        """Call after backward, before the optimizer step: launch the remaining buckets, wait for the collectives (the compute
        stream waits, not the host), p.grad = slot views.  -> the flat buffer"""
        if not self._sync:
            raise RuntimeError("FlatGradReducer.finish() inside no_sync(): run the last micro-step's backward outside the context")
        while self._next < len(self.buckets):
            self._launch(self._next)
            self._next += 1
        for w in self._works:
            w.wait()
        if self.comm is not None and not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream().wait_stream(self.comm)
        local = [p.grad is not None for p in self.params]
        take = local
        if dist.is_initialized() and dist.get_world_size() > 1 and (self._any is None or self.comm is None or not torch.cuda.is_current_stream_capturing()):
            # (while a hipGraph of the step is being captured the exchange below - a host read - cannot run: the mask of the warm-up
            # steps before the capture is used, graph.GraphedTrainStep)
            # a parameter whose gradient exists on ANOTHER rank only must still receive the reduced sum here (torch DDP hands every rank
            # the reduced gradient), or this rank's optimizer would skip it and the replicas drift apart (round-3 advisor finding).  Which
            # parameters have a gradient on some rank is a property of the graph: it is exchanged at the first finish() and re-checked
            # every MASK_EVERY steps (one small MAX all-reduce + host read; every rank does it at the same step count).
            if self._any is None or self._nfinish % self.MASK_EVERY == 0:
                t = torch.tensor(local, dtype=torch.int32, device=self.device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                self._any = [bool(v) for v in t.cpu().tolist()]
            self._nfinish += 1
            take = [a or b for a, b in zip(self._any, local)]
        if dist.is_initialized() and dist.get_world_size() > 1 and self._any is not None and take is local:
            take = [a or b for a, b in zip(self._any, local)]  # capturing: the cached mask
        for i, p in enumerate(self.params):
            if take[i]:
                p.grad = self.views[i]
        events = self._events
        self._reset()
        self._events_done = events if (self.timing and events) else None
        return self.flat

    reduce = finish  # round-1 name

    def times(self):
        """timing=True: per-bucket (bytes, ms from 'gradients ready' to 'all-reduce done', all-reduce ms) of the last finished step
        (call after a synchronize)"""
        ev = getattr(self, "_events_done", None)
        if not ev:
            return None
        return [(4 * n, e0.elapsed_time(t1), t0.elapsed_time(t1)) for e0, t0, t1, n in ev]


def scale_loss(loss: torch.Tensor, world: int) -> torch.Tensor:
    """reference engine/trainer.py:401-402"""
    return loss * world if world > 1 else loss


def max_over_ranks(value: float, device) -> float:
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t)
