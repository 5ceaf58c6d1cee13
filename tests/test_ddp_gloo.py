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
    """test-only: the gather launch (`y3d_mt_copy`, HIP) replaced by slot copies so that the collective logic (bucket plan, hooks,
    in-order launches, finish) runs over gloo on CPU"""

    def _gather(self, idx, grads):
        for i, g in zip(idx, grads):
            self.views[i].copy_(g)


def _worker_flat(rank, world, port, ret):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    ddp.init("gloo")
    torch.manual_seed(rank)  # different initial weights per rank: broadcast_parameters must repair that
    net = TinyNet()
    net.c2.bias.requires_grad_(True)
    red = _CpuReducer(net.parameters(), bucket_mb=0.0005, overlap=True)  # tiny buckets: several collectives per step
    assert len(red.buckets) >= 2
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


def _tiny3d():
    import yolov10_3d_amd as y3d
    cfg = y3d.yaml_model_load("yolov10s_3D.yaml")
    cfg.update(scales={"n": [0.33, 0.125, 1024]}, scale="n",
               channels={k + "_c": 16 for k in ("cls", "o2d", "s2d", "o3d", "s3d", "hd", "dep", "dep_un")})
    return y3d.YOLOv10_3DDetectionModel(cfg)


def _fake_grad(p, i, rank):
    """deterministic per (parameter, rank) gradient"""
    g = torch.Generator().manual_seed(1000 * i + rank)
    return torch.randn(p.shape, generator=g)


def _worker_real_model(rank, world, port, ret, overlap):
    """the REAL tiny 3D model's parameter set: stacked head storage (restack), broadcast from rank 0, bucketed reduce"""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    ddp.init("gloo")
    torch.manual_seed(100 + rank)  # ranks start from DIFFERENT weights and BatchNorm statistics
    model = _tiny3d()
    for b in model.modules():
        if isinstance(b, torch.nn.BatchNorm2d):
            b.running_mean.uniform_(-1, 1)
    head = model.model[-1]
    head.restack()  # bench.py order: stack first, then build the reducer, then broadcast
    ptrs = {id(p): p.data_ptr() for p in model.parameters()}
    red = _CpuReducer(model.parameters(), bucket_mb=0.25, overlap=overlap)
    red.broadcast_parameters(model)
    head.restack()
    assert all(p.data_ptr() == ptrs[id(p)] for p in model.parameters()), "broadcast / restack moved a parameter"
    # the stacked views still alias the per-branch parameters after the broadcast
    _, _, s1, s2, _, _ = head._stacks(0)
    w, gm, bt, rm, rv = s1.tensors()
    off = 0
    for c in s1.convs:
        n = c.conv.weight.numel()
        assert c.conv.weight.data_ptr() == w.data_ptr() + 4 * off and torch.equal(c.conv.weight.detach().reshape(-1), w.reshape(-1)[off:off + n])
        off += n
    assert len(red.params) == len(list(model.parameters())) and len(red.buckets) >= 4
    assert red.params[0] is list(model.parameters())[-1], "layout is head-first (reverse registration order)"
    plist = list(model.parameters())
    index = {id(p): i for i, p in enumerate(plist)}
    skip = {id(plist[3])}  # one parameter without a gradient (an unused detect level): its slot stays zero on every rank
    for step in range(2):
        for p in plist:
            p.grad = None
        # a backward pass stand-in that fires the post-accumulate hooks in reverse registration order (head first)
        loss = sum((p * _fake_grad(p, index[id(p)], rank + 10 * step)).sum() for p in plist if id(p) not in skip)
        loss.backward()
        flat = red.finish()
        assert plist[3].grad is None
        assert all(p.grad.data_ptr() == red.views[i].data_ptr() for i, p in enumerate(red.params) if p.grad is not None)
    ret[rank] = {"grads": [None if p.grad is None else p.grad.clone() for p in plist], "params": [p.detach().clone() for p in plist],
                 "buffers": [b.clone() for b in model.buffers()], "nbuckets": len(red.buckets)}
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_world2_gloo_reducer_on_real_tiny_model(overlap):
    """W5 of round 1: the N>1 path of bench.py on the real model's ~570 re-pointed / stacked parameters — restack() +
    broadcast_parameters order, head-first bucket plan, hook-driven in-order launches (overlap) and the after-backward path"""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_real_model, args=(world, port, ret, overlap), nprocs=world, join=True)
    torch.manual_seed(100)
    ref = _tiny3d()  # rank 0's initial model
    for b in ref.modules():
        if isinstance(b, torch.nn.BatchNorm2d):
            b.running_mean.uniform_(-1, 1)
    plist = list(ref.parameters())
    for r in range(world):
        for a, b in zip(ret[r]["params"], plist):
            assert torch.equal(a, b.detach()), "parameters were not broadcast from rank 0"
        for a, b in zip(ret[r]["buffers"], ref.buffers()):
            assert torch.equal(a, b)
        for i, (g, p) in enumerate(zip(ret[r]["grads"], plist)):
            if i == 3:
                assert g is None
                continue
            want = _fake_grad(p, i, 10) + _fake_grad(p, i, 11)  # step 1: SUM over the two ranks
            assert torch.allclose(g, want, rtol=1e-6, atol=1e-6), i


def test_shard_batch_decides_by_key_not_by_shape():
    """exactly nc = 3 boxes next to the (3, 3) mean_sizes, and as many boxes as images"""
    B = 4
    for nbox in (3, B):
        bi = torch.tensor([0.0, 2.0, 3.0] if nbox == 3 else [0.0, 1.0, 2.0, 3.0])
        full = {"img": torch.rand(B, 3, 8, 8), "batch_idx": bi, "cls": torch.arange(nbox).float().view(-1, 1), "bboxes": torch.rand(nbox, 4),
                "depth": torch.arange(nbox).float(), "calib": torch.arange(B * 6).float().view(B, 6),
                "mean_sizes": torch.arange(9).float().view(3, 3), "mixed": torch.zeros(B, dtype=torch.uint8), "im_file": [f"{i}.png" for i in range(B)]}
        seen = 0
        for r in range(2):
            loc = ddp.shard_batch(full, r, 2)
            assert torch.equal(loc["mean_sizes"], full["mean_sizes"])           # replicated, never sliced
            assert torch.equal(loc["calib"], full["calib"][2 * r:2 * r + 2])     # per image
            assert loc["im_file"] == full["im_file"][2 * r:2 * r + 2]
            sel = (bi >= 2 * r) & (bi < 2 * r + 2)
            assert torch.equal(loc["depth"], full["depth"][sel]) and torch.equal(loc["batch_idx"], bi[sel] - 2 * r)
            seen += loc["depth"].numel()
        assert seen == nbox
    with pytest.raises(KeyError):
        ddp.shard_batch({**full, "mystery": torch.zeros(B)}, 0, 2)


def test_bench_spawns_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` without a launcher starts `python -m torch.distributed.run ... bench.py --gpus N` as a CHILD
    process (reference scheme: utils/dist.py:55-65) and relays its exit code; under a launcher it does not spawn"""
    import subprocess
    import sys
    import bench
    calls = []
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: calls.append((cmd, env)) or 7)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7 and len(calls) == 1
    cmd, env = calls[0]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "8", "--steps", "3"]
    assert os.path.basename(cmd[cmd.index("--master-port") + 2]) == "bench.py"
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


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


def _worker_accum(rank, world, port, ret):
    """two micro-steps per optimizer step (no_sync + final), a parameter that has a gradient in step 0 only (stale
    slot), and the misuse the contract forbids (backward into an already reduced slot) raising instead of double-reducing"""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    ddp.init("gloo")
    torch.manual_seed(0)
    net = TinyNet()
    extra = torch.nn.Parameter(torch.ones(5))  # gets a gradient in step 0 on every rank, in step 1 on rank 1 only
    params = list(net.parameters()) + [extra]
    red = _CpuReducer(params, bucket_mb=0.0005, overlap=True)
    red.broadcast_parameters(net)
    micro = [ddp.shard_batch(_make_batch(seed=s), rank, world) for s in (0, 1)]
    out = {}
    for step in range(2):
        for p in params:
            p.grad = None
        with red.no_sync():
            loss, _ = net(micro[0])
            if step == 0 or rank == 1:
                loss = loss + (extra * (1.0 + rank + step)).sum()
            loss.backward()
            with pytest.raises(RuntimeError):
                red.finish()
        loss, _ = net(micro[1])
        loss.backward()
        red.finish()
        out[step] = [None if p.grad is None else p.grad.clone() for p in params]
    # misuse: a second backward without clearing the gradients accumulates into the reduced slots
    loss, _ = net(micro[0])
    with pytest.raises(RuntimeError, match="already all-reduced"):
        loss.backward()
        red.finish()
    ret[rank] = out
    dist.destroy_process_group()


def test_world2_gloo_reducer_accumulation_and_stale_slots():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_accum, args=(world, port, ret), nprocs=world, join=True)
    torch.manual_seed(0)
    net = TinyNet()
    for s in (0, 1):  # single process: both micro-batches, whole
        loss, _ = net(_make_batch(seed=s))
        loss.backward()
    ref = [p.grad for p in net.parameters()]
    for r in range(world):
        for step in range(2):
            got = ret[r][step]
            for a, b in zip(got[:-1], ref):
                assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), (step, (a - b).abs().max())
        # step 0: every rank contributes 1 + rank; step 1: only rank 1 (2 + 1): rank 0's stale step-0 slot must not be added again
        assert torch.allclose(ret[r][0][-1], torch.full((5,), 1.0 + 2.0))
    # rank 0 had no gradient for it in step 1, rank 1 did: BOTH ranks hold the reduced sum (as under torch DDP), so their optimizers
    # take the same step (round-3 advisor finding: rank 0 used to keep p.grad = None and the replicas diverged)
    for r in range(world):
        assert torch.allclose(ret[r][1][-1], torch.full((5,), 3.0)), ret[r][1][-1]


def test_shard_batch_takes_the_reference_collate_key_set():
    """exactly the keys of the reference's collate_fn (data/datasets/kitti.py:421-442, 579-599) - incl. the stacked
    `depth_map` placeholder, the un-stacked `ori_img` / `info` / `im_file` / `ori_shape` tuples"""
    B, nbox = 4, 6
    bi = torch.tensor([0.0, 0.0, 1.0, 2.0, 3.0, 3.0])
    full = {"img": torch.rand(B, 3, 8, 8), "ori_img": tuple(object() for _ in range(B)), "calib": torch.rand(B, 6),
            "info": tuple({"img_id": i} for i in range(B)), "cls": torch.zeros(nbox), "bboxes": torch.rand(nbox, 4), "batch_idx": bi,
            "im_file": tuple(f"{i:06d}.txt" for i in range(B)), "ori_shape": tuple((375, 1242) for _ in range(B)), "ratio_pad": torch.rand(B, 2, 2),
            "center_2d": torch.rand(nbox, 2), "center_3d": torch.rand(nbox, 2), "size_2d": torch.rand(nbox, 2), "size_3d": torch.rand(nbox, 3),
            "depth": torch.rand(nbox), "depth_map": torch.empty(B, 1), "mean_sizes": torch.rand(3, 3), "heading_bin": torch.zeros(nbox),
            "heading_res": torch.rand(nbox), "mixed": torch.zeros(B, dtype=torch.uint8)}
    for r in range(2):
        loc = ddp.shard_batch(full, r, 2)
        assert set(loc) == set(full)
        assert loc["depth_map"].shape == (2, 1) and loc["ratio_pad"].shape == (2, 2, 2)
        assert loc["ori_img"] == full["ori_img"][2 * r:2 * r + 2] and loc["info"] == full["info"][2 * r:2 * r + 2]
        assert loc["ori_shape"] == full["ori_shape"][2 * r:2 * r + 2]
        assert loc["size_3d"].shape[0] == loc["batch_idx"].numel() == 3
