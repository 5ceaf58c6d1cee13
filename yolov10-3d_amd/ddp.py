"""Image-parallel data parallelism of the hot path: one process per GPU, gradients all-reduced over RCCL/xGMI.

Reference behaviour restated (engine/trainer.py:225-236, 280, 292, 401-402; utils/dist.py:55-65):
  * the global batch is split evenly by rank (`batch // world_size` images per process), every rank runs the whole
    model on its shard with its OWN BatchNorm statistics (no SyncBN) and its own assigner / loss;
  * the loss is `loss.sum() * local_batch` (utils/loss.py:900) and is multiplied by `world_size` before backward because the
    all-reduce AVERAGES gradients — so the update equals the single-process update on the global batch;
  * the only collective on the data path is that gradient all-reduce (fp32, S-3D: 120 MB per step).

torch.distributed is the plumbing (backend "nccl" == RCCL on ROCm, "gloo" for the CPU tests).  Two reducers: `FlatGradReducer`
(bench.py's N>1 path: one gather launch + one all-reduce of the flat 120 MB buffer per step) and `wrap` (torch
DistributedDataParallel with buckets as views, kept as the drop-in for code that expects a DDP module).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


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
    """Rank's slice of a collated batch dict (schema: SURVEY §8b).  Per-image tensors (`img`, `calib`, `mixed`) are split
    evenly; per-box tensors follow their `batch_idx`, which is re-based to the local image range."""
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
        if not torch.is_tensor(v):
            out[k] = v
        elif k == "batch_idx":
            out[k] = v[sel] - lo
        elif v.dim() > 0 and v.shape[0] == bi.shape[0] and k not in ("img", "calib", "mixed"):
            out[k] = v[sel]
        elif v.dim() > 0 and v.shape[0] == B and k != "mean_sizes":
            out[k] = v[lo:hi]
        else:
            out[k] = v
    return out


def wrap(model: torch.nn.Module, device_ids=None, bucket_cap_mb: float = 32.0):
    """DistributedDataParallel with gradient buckets as views (no extra copy) and a static bucket plan: every parameter of
    the YOLOv10(-3D) graph receives a gradient every step, so nothing is ever 'unused'."""
    return torch.nn.parallel.DistributedDataParallel(model, device_ids=device_ids, bucket_cap_mb=bucket_cap_mb,
                                                     gradient_as_bucket_view=True, static_graph=True)


class FlatGradReducer:
    """Gradient all-reduce over ONE flat fp32 buffer (S-3D: 120 MB), the data-path collective of the reference's DDP.

    After backward every gradient tensor is gathered into its slot of the flat buffer by one multi-tensor launch (`y3d_mt_copy`),
    the buffer is all-reduced (SUM: with the unscaled local losses this equals the reference's `loss * world_size` + averaged
    gradients, trainer.py:401-402), and `p.grad` is re-pointed at the slot, so the fused optimizer runs on stable pointers.
    Why not torch DDP: its autograd hooks copy each of the ~570 gradient tensors into a bucket view one by one (+7 % step time
    measured at one rank); the gather is one launch at HBM rate.  Parameters without a gradient (an unused detect level) keep
    `grad = None` on every rank and their slots stay zero."""

    CHUNK = 16384

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatGradReducer: no parameters")
        dev = self.params[0].device
        self.sizes = [p.numel() for p in self.params]
        self.flat = torch.zeros(sum(self.sizes), dtype=torch.float32, device=dev)
        self.views, off = [], 0
        for p, n in zip(self.params, self.sizes):
            self.views.append(self.flat[off:off + n].view_as(p))
            off += n
        self._active, self._tab = None, None

    def broadcast_parameters(self, module: torch.nn.Module):
        """rank 0's parameters and buffers to every rank (what DistributedDataParallel does when it wraps a module)"""
        if dist.is_initialized() and dist.get_world_size() > 1:
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, 0)

    def _gather(self, active, grads):
        from ._lib import Y3DError, lib
        from . import ops
        dev = self.flat.device
        if dev.type != "cuda":
            raise Y3DError("FlatGradReducer needs gradients on a HIP device")
        if self._active != active:
            sizes = [self.sizes[i] for i in active]
            ct, co = [], []
            for t, n in enumerate(sizes):
                for c in range((n + self.CHUNK - 1) // self.CHUNK):
                    ct.append(t)
                    co.append(c)
            self._tab = {"sizes": torch.tensor(sizes, dtype=torch.int64, device=dev), "ct": torch.tensor(ct, dtype=torch.int32, device=dev),
                         "co": torch.tensor(co, dtype=torch.int32, device=dev), "n": len(ct),
                         "dst": torch.tensor([self.views[i].data_ptr() for i in active], dtype=torch.int64, device=dev)}
            self._active = active
        tb = self._tab
        if tb.get("up") is None:
            from .optim import PtrUploader
            tb["up"] = PtrUploader(len(active), dev)
        src = tb["up"].upload([g.data_ptr() for g in grads])  # non-blocking: the host must not wait for the backward to drain
        lib().mt_copy(src.data_ptr(), tb["dst"].data_ptr(), tb["sizes"].data_ptr(), tb["ct"].data_ptr(), tb["co"].data_ptr(), tb["n"], self.CHUNK,
                      1.0, ops.stream())

    def reduce(self):
        """gather -> all-reduce(SUM) -> p.grad = slot views.  Call after backward, before the optimizer step."""
        active = [i for i, p in enumerate(self.params) if p.grad is not None]
        grads = []
        for i in active:
            g = self.params[i].grad
            if g.dtype != torch.float32 or not g.is_contiguous():
                g = g.float().contiguous()
            grads.append(g)
        self._gather(active, grads)
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat)
        for i in active:
            self.params[i].grad = self.views[i]
        return self.flat


def scale_loss(loss: torch.Tensor, world: int) -> torch.Tensor:
    """reference engine/trainer.py:401-402"""
    return loss * world if world > 1 else loss


def max_over_ranks(value: float, device) -> float:
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t)
