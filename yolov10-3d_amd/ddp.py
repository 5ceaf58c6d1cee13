"""Image-parallel data parallelism of the hot path: one process per GPU, gradients all-reduced over RCCL/xGMI.

Reference behaviour restated (engine/trainer.py:225-236, 280, 292, 401-402; utils/dist.py:55-65):
  * the global batch is split evenly by rank (`batch // world_size` images per process), every rank runs the whole
    model on its shard with its OWN BatchNorm statistics (no SyncBN) and its own assigner / loss;
  * the loss is `loss.sum() * local_batch` (utils/loss.py:900) and is multiplied by `world_size` before backward because the
    all-reduce AVERAGES gradients — so the update equals the single-process update on the global batch;
  * the only collective on the data path is that gradient all-reduce (fp32, S-3D: 120 MB per step).

torch.distributed is the plumbing (backend "nccl" == RCCL on ROCm, "gloo" for the CPU tests).  Buckets are sized so that the
head's gradients (79 % of the S-3D bytes, produced FIRST in backward — SURVEY §8e) are already in flight while the
backbone's dgrad/wgrad kernels run.
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


def scale_loss(loss: torch.Tensor, world: int) -> torch.Tensor:
    """reference engine/trainer.py:401-402"""
    return loss * world if world > 1 else loss


def max_over_ranks(value: float, device) -> float:
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t)
