"""CPU, world_size 2 over gloo: the N>1 path (shard the collated batch by rank, per-rank loss * world, averaged gradients)
reproduces the single-process update on the global batch — the reference's DDP contract (trainer.py:292, 401-402)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import yolov10_3d_amd  # noqa: F401
from yolov10_3d_amd import ddp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class TinyNet(torch.nn.Module):
    """stand-in with the model-level call convention: net(batch) -> (loss_sum * local_batch, items)"""

    def __init__(self):
        super().__init__()
        self.c1 = torch.nn.Conv2d(3, 8, 3, 2, 1)
        self.c2 = torch.nn.Conv2d(8, 4, 3, 2, 1)

    def forward(self, batch):
        y = self.c2(torch.nn.functional.silu(self.c1(batch["img"])))
        B = y.shape[0]
        per_img = y.flatten(1).pow(2).mean(1)                       # one term per image
        per_box = (batch["depth"] * per_img[batch["batch_idx"].long()]).sum()  # one term per GT box
        loss = (per_img.mean() + per_box / B)                        # "mean over the local batch" loss ...
        return loss * B, loss.detach().reshape(1)                    # ... times the local batch (utils/loss.py:900)


def _make_batch(B=8, seed=0):
    g = torch.Generator().manual_seed(seed)
    counts = torch.randint(1, 4, (B,), generator=g)
    return {"img": torch.rand(B, 3, 16, 16, generator=g), "batch_idx": torch.repeat_interleave(torch.arange(B), counts).float(),
            "depth": torch.rand(int(counts.sum()), generator=g), "calib": torch.rand(B, 6, generator=g),
            "mean_sizes": torch.rand(3, 3, generator=g)}


def _worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w = ddp.init("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    net = TinyNet()
    dnet = ddp.wrap(net)
    full = _make_batch()
    local = ddp.shard_batch(full, rank, world)
    assert local["img"].shape[0] == full["img"].shape[0] // world
    assert local["batch_idx"].min() >= 0 and local["batch_idx"].max() < local["img"].shape[0]
    assert local["mean_sizes"].shape == full["mean_sizes"].shape
    loss, items = dnet(local)
    ddp.scale_loss(loss, world).backward()
    t = ddp.max_over_ranks(float(rank), "cpu")
    assert t == world - 1
    ret[rank] = [p.grad.clone() for p in net.parameters()]
    dist.destroy_process_group()


class _CpuReducer(ddp.FlatGradReducer):
    """test-only: the gather launch (`y3d_mt_copy`, HIP) replaced by slot copies so that the collective logic runs over gloo on CPU"""

    def _gather(self, active, grads):
        for i, g in zip(active, grads):
            self.views[i].copy_(g)


def _worker_flat(rank, world, port, ret):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    ddp.init("gloo")
    torch.manual_seed(rank)  # different initial weights per rank: broadcast_parameters must repair that
    net = TinyNet()
    net.c2.bias.requires_grad_(True)
    red = _CpuReducer(net.parameters())
    red.broadcast_parameters(net)
    local = ddp.shard_batch(_make_batch(), rank, world)
    for _ in range(2):  # the second step re-uses the slot views
        for p in net.parameters():
            p.grad = None
        loss, _ = net(local)
        loss.backward()
        red.reduce()
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(red.params, red.views))
    ret[rank] = [p.grad.clone() for p in net.parameters()] + [p.detach().clone() for p in net.parameters()]
    dist.destroy_process_group()


def test_world2_gloo_flat_reducer_matches_single_process():
    """ddp.FlatGradReducer: SUM all-reduce of the unscaled per-rank losses' gradients == the single-process gradient on the global
    batch (the reference's `loss * world_size` + averaged gradients), parameters broadcast from rank 0"""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_flat, args=(world, port, ret), nprocs=world, join=True)
    torch.manual_seed(0)
    net = TinyNet()
    loss, _ = net(_make_batch())
    loss.backward()
    ref = [p.grad for p in net.parameters()]
    n = len(ref)
    for r in range(world):
        for a, b in zip(ret[r][:n], ref):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), (a - b).abs().max()
        for a, b in zip(ret[r][n:], net.parameters()):
            assert torch.equal(a, b.detach())


def test_world2_gloo_matches_single_process():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    torch.manual_seed(0)
    net = TinyNet()
    loss, _ = net(_make_batch())
    loss.backward()
    ref = [p.grad for p in net.parameters()]
    for r in range(world):
        for a, b in zip(ret[r], ref):
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)


def test_shard_batch_partitions_boxes():
    full = _make_batch(B=8, seed=3)
    n = 0
    for r in range(4):
        loc = ddp.shard_batch(full, r, 4)
        n += loc["batch_idx"].numel()
        assert loc["img"].shape[0] == 2 and loc["depth"].shape == loc["batch_idx"].shape
    assert n == full["batch_idx"].numel()
