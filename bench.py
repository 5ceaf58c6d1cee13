"""bench.py — images/sec of the YOLOv10-S-3D 640x640 hot path on N MI355X (one process per GPU, RCCL over xGMI).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus 8            # no launcher: spawns its 8 ranks itself (child `python -m torch.distributed.run ...`, the
                                        # reference's own scheme, utils/dist.py:55-65) before touching the GPU, relays rank 0's line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one training pass of the hot path over one synthetic batch that is already resident in HBM:
forward (backbone + neck + PSA + dual 3D head) + dual-assignment loss + backward + gradient clip + optimizer
step.  Workload = BASELINE.json configs[1]: YOLOv10-S + 3D head, 640x640, bf16, batch 32 per GPU (weak scaling:
the reference's image-parallel DDP, trainer.py:292).  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline     — dominant kernel (3x3 128->128 @80x80 implicit-GEMM conv forward, 54 % of the step's FLOPs),
                 timed live with HIP events on the launch stream during the timed region; algorithmic FLOPs
                 2*B*Ho*Wo*Cout*Cin*9 per launch against the dense bf16 MFMA peak (2.5 PFLOP/s, MI355X_MICROARCH.md).
  cpu_baseline — the CPU oracle restatement (oracle/restate.py, fp32, all host cores) on a bounded sample of the same
                 workload (B=2 steps, 2 warm-ups + median of 5), kind "port".  A reported baseline, never the product path.

The timed loop rotates over NBATCH pre-staged batch dicts (different seeds, all resident in HBM), so every step sees a batch it
has not just processed and per-batch work (target padding, its kernels) is inside the timed region, as in the reference's loop.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md §Chip-level parameters
MFMA_FP8_DENSE_PEAK_TFLOPS = 5000.0   # same guide: block-scaled fp8 (v_mfma_scale_f32_16x16x128_f8f6f4), dense
MFMA_F32_PEAK_TFLOPS = 157.3


def synth_batch(B, H, W, seed, device, nc=3):
    """SURVEY §8d synthetic recipe: n ~ U{1..8} boxes per image, KITTI-like calibration and mean sizes."""
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(B, 3, H, W, generator=g)
    counts = torch.randint(1, 9, (B,), generator=g)
    bi = torch.repeat_interleave(torch.arange(B), counts).float()
    n = int(counts.sum())
    cxy = 0.2 + 0.6 * torch.rand(n, 2, generator=g)
    wh = 0.05 + 0.25 * torch.rand(n, 2, generator=g)
    scale = torch.tensor([W, H], dtype=torch.float32)
    c2 = cxy * scale
    batch = {
        "img": img, "batch_idx": bi, "cls": torch.randint(0, nc, (n, 1), generator=g).float(), "bboxes": torch.cat((cxy, wh), 1),
        "center_2d": c2, "size_2d": wh * scale, "center_3d": c2 + 2.0 * torch.randn(n, 2, generator=g),
        "size_3d": 0.1 * torch.randn(n, 3, generator=g), "depth": 5 + 55 * torch.rand(n, generator=g),
        "heading_bin": torch.randint(0, 12, (n,), generator=g).float(), "heading_res": (torch.rand(n, generator=g) - 0.5) * (math.pi / 6),
        "calib": torch.tensor([[W / 2, H / 2, 700.0, 700.0, 0.06, -0.002]]).repeat(B, 1),
        "mean_sizes": torch.tensor([[1.76255119, 0.66068622, 0.84422524], [1.52563191, 1.62856739, 3.88311640],
                                    [1.73698127, 0.59706367, 1.76282397]]),
    }
    return {k: v.to(device) for k, v in batch.items()}


def conv_key_flops(key):
    """algorithmic FLOPs of one launch of an ops._timed conv key: 2 * B * Ho * Wo * Cout * (Cin / g) * k * k"""
    _, _, B, H, W, Cin, Cout, k, s, g = key[:10]
    pad = key[10] if len(key) > 10 else k // 2
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    return 2.0 * B * Ho * Wo * Cout * (Cin // g) * k * k


def roofline_key(model, seen, dt_code, B, S):
    """the launch bench.py's `roofline` object describes.  3D heads: the fused second head layer at the stride-8 level (16 sibling
    128->128 3x3 convs as ONE grouped launch, SURVEY §0.4's 54 %-of-FLOPs shape).  2D models (no stacked head): the 3x3 stride-1
    forward launch with the most algorithmic FLOPs among the launches a warm-up step made."""
    head = model.model[-1]
    if hasattr(head, "o2o_heads"):
        mid = head.o2o_heads[0][0][1].conv.in_channels
        k2 = head.o2o_heads[0][0][1].conv.kernel_size[0]
        if k2 == 3 and len({h[0][1].conv.in_channels for h in head.o2o_heads}) == 1:
            key = ("conv_fwd", dt_code, B, S // 8, S // 8, 16 * mid, 16 * mid, 3, 1, 16)
            f8 = ("conv_fwd_fp8",) + key[1:]  # the same launch on the fp8 MFMA kernel (--weights fp8)
            return f8 if f8 in seen else key
    cands = [k for k in seen if k[0] in ("conv_fwd", "conv_fwd_fp8") and k[7] == 3 and k[8] == 1]
    return max(cands, key=lambda k: (conv_key_flops(k), k)) if cands else None


NBATCH = 4


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD process tree (torch.distributed.run), before this
    process has made any HIP call, hand through stdout / stderr (rank 0 prints the JSON line) and return the child's exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL's buffer exchange needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    return subprocess.call(cmd, env=env)


def cpu_baseline(model_name, imgsz, seed, steps=5, B=2, warm=2):
    """oracle restatement (fp32) timed on the host cores: fwd + loss + bwd on a bounded sample (B=2 per step)."""
    import yaml

    from oracle import restate as RS

    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))  # the GPU box gives one GPU's CPU share (16 cores); more threads only oversubscribe
    torch.set_num_threads(cores)
    with open(os.path.join(ROOT, "yolov10-3d_amd", "cfg", "models", "v10-3D", model_name)) as f:
        cfg = yaml.safe_load(f)
    cfg["scale"] = RS.guess_scale(model_name)
    spec = RS.build_spec(cfg)
    st = RS.init_state(spec, seed=0, randomize_bn=False)
    params = [v.requires_grad_(True) for k, v in st.items() if v.is_floating_point() and "running" not in k]
    batch = synth_batch(B, imgsz, imgsz, seed, "cpu")
    strides = RS.model_strides(spec)
    times = []
    for i in range(steps + warm):
        t0 = time.perf_counter()
        preds = RS.forward(spec, st, batch["img"], True)
        loss, items, _ = RS.loss3d(preds, batch, strides, 3)
        loss.backward()
        for p in params:
            p.grad = None
        if i >= warm:
            times.append(time.perf_counter() - t0)
    t = sorted(times)[len(times) // 2]
    return {"value": B / t, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"oracle/restate.py fp32 train step (fwd+loss+bwd), {model_name} {imgsz}x{imgsz}, B={B}, median of {steps} steps after {warm} warm-ups"}


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--imgsz", type=int, default=640)
    ap.add_argument("--model", default="yolov10s_3D.yaml")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--weights", default="full", choices=["full", "fp8"],
                    help="fp8: conv weights as OCP e4m3 codes with per-output-channel power-of-two scales (BASELINE configs[4]; csrc/fp8w.hip)")
    ap.add_argument("--fp8-emulate", action="store_true",
                    help="with --weights fp8: keep every convolution on the bf16 matrix cores (fp8-VALUED weights only: the round-3 form), for A/B runs "
                         "against the fp8 MFMA convolutions (csrc/conv3x3_fp8.hip) that --weights fp8 turns on by default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--infer-steps", type=int, default=40)
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL) | gloo (rehearsal of the N>1 path on one GPU)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--phases", action="store_true", help="per-phase device times (backward / all-reduce / optimizer) in the JSON line (default for N > 1)")
    ap.add_argument("--no-phases", action="store_true", help="N > 1: no per-phase events")
    ap.add_argument("--train-graph", action="store_true", help="(default at N = 1; kept for old command lines)")
    ap.add_argument("--no-train-graph", action="store_true",
                    help="N = 1: do not ALSO time the training step replayed from one hipGraph (graph.GraphedTrainStep).  By default both the "
                         "launch-by-launch loop and the replay are timed over --steps steps each; `value` is the faster of the two (`train_mode` "
                         "says which: the eager loop is bound by the host on a slow CPU, the replay never is), both rates are in the line")
    ap.add_argument("--no-val-batch", action="store_true", help="skip the second inference leg at the reference's validation batch (2 x --batch, engine/trainer.py:297)")
    ap.add_argument("--infer-mode", default="graph", choices=["graph", "eager"],
                    help="inference leg: replay the eval forward + postprocess from one captured hipGraph per batch (default), or launch eagerly")
    args = ap.parse_args(argv)
    if args.gpus > 1 and not args.no_phases:
        args.phases = True  # four event records per step; the exposed all-reduce time is what a scaling run is read for

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, list(sys.argv[1:] if argv is None else argv)))  # nothing in this process has touched the GPU yet

    import torch.distributed as dist

    import yolov10_3d_amd as y3d
    from yolov10_3d_amd import ops
    from yolov10_3d_amd.loss import v10_3Dpostprocess, v10postprocess

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.same_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from yolov10_3d_amd import ddp
    ddp.init(args.backend, dev)

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    y3d.set_compute_dtype(dtype)
    if args.weights == "fp8":
        y3d.set_weight_quant("fp8")
        y3d.set_fp8_conv(not args.fp8_emulate and dtype == torch.bfloat16)
    torch.manual_seed(0)
    is3d = "3D" in args.model  # the 2D yamls (BASELINE configs[0] / L-2D) run through the same step for profiling
    model = (y3d.YOLOv10_3DDetectionModel if is3d else y3d.YOLOv10DetectionModel)(args.model).to(dev).train()
    from yolov10_3d_amd.optim import ModelEMA, build_optimizer
    opt = build_optimizer(model)  # reference engine/trainer.py:734-790 groups; fused clip + SGD(nesterov) HIP step
    net = model
    if hasattr(model.model[-1], "restack"):
        model.model[-1].restack()  # sibling-branch parameter stacking must be in place before parameters / buffers are recorded
    ema = ModelEMA(model) if rank == 0 else None  # reference: rank 0 only (engine/trainer.py:294-302), updated in optimizer_step (:574-575)
    reducer = None
    if world > 1 or os.environ.get("Y3D_FORCE_DDP"):
        if os.environ.get("Y3D_TORCH_DDP"):
            net = ddp.wrap(model, device_ids=[local])  # torch DistributedDataParallel (buckets as views)
        else:
            # head-first flat gradient buffer, bucketed RCCL all-reduce on a side stream behind the head backward
            reducer = ddp.FlatGradReducer(model.parameters(), timing=args.phases)
            reducer.broadcast_parameters(model)
            if hasattr(model.model[-1], "restack"):
                model.model[-1].restack()  # no-op check: the broadcast wrote in place, the stacked storage is still the parameters' storage
    B, S = args.batch, args.imgsz
    # NBATCH different batches resident in HBM before the timed region; the loop rotates over them
    batches = [synth_batch(B, S, S, seed=1 + rank + 1000 * j, device=dev, nc=model.yaml["nc"]) for j in range(NBATCH)]
    batch = batches[0]
    counter = [0]
    phase_ev = []

    def step():
        bt = batches[counter[0] % NBATCH]
        counter[0] += 1
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if args.phases else None
        if ev:
            ev[0].record()
        loss, items = net(bt)
        if reducer is not None:
            loss.backward()   # SUM all-reduce of the unscaled losses' gradients == reference's loss * world + averaged gradients
            if ev:
                ev[1].record()
            reducer.finish()  # compute stream waits for the bucket collectives that are still in flight
        else:
            ddp.scale_loss(loss, world).backward()  # reference trainer.py:401-402 (the all-reduce averages gradients)
            if ev:
                ev[1].record()
        if ev:
            ev[2].record()
        opt.step(max_norm=10.0)  # clip_grad_norm_(10) + SGD nesterov (trainer.py:570-571) in three multi-tensor launches
        opt.zero_grad(set_to_none=True)
        if ema is not None:
            ema.update(model, guard=opt.last_norm)  # one multi-tensor launch over the whole state_dict; skipped with a skipped step
        if ev:
            ev[3].record()
            phase_ev.append(ev)
        return items

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    log(f"model {args.model} built, {sum(p.numel() for p in model.parameters()) / 1e6:.2f} M params; warm-up {args.warmup} steps")
    seen = {}

    def collect(key):  # a warm-up pass records every conv launch key (nothing is bracketed): the 2D models pick their roofline launch from it
        seen[key] = seen.get(key, 0) + 1
        return False

    for i in range(args.warmup):
        tw = time.perf_counter()
        if i == args.warmup - 1:
            ops.TIMER = ops.KernelTimer(collect)
        items = step()
        ops.TIMER = None
        torch.cuda.synchronize()
        log(f"warm-up step {i}: {1e3 * (time.perf_counter() - tw):.1f} ms, loss items {[round(float(v), 3) for v in items.float().cpu()]}")
    dt_code = 1 if dtype == torch.bfloat16 else 0
    k1_key = roofline_key(model, seen, dt_code, B, S)
    ops.TIMER = ops.KernelTimer(lambda key: key == k1_key)
    phase_ev.clear()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        items = step()
    sync()
    dt_s = time.perf_counter() - t0
    timer, ops.TIMER = ops.TIMER, None
    tmax = torch.tensor([dt_s], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt_s = float(tmax)
    log(f"timed {args.steps} steps: {1e3 * dt_s / args.steps:.2f} ms/step")
    assert torch.isfinite(items).all(), f"non-finite loss items {items}"
    train_ips = world * B * args.steps / dt_s

    train_graph_ips = None
    if not args.no_train_graph and net is model:
        # the same step - forward, loss, backward, (N > 1: the reducer's bucket all-reduces on their side stream,) clip, SGD - replayed
        # from one captured hipGraph, EMA eagerly behind it; the batches rotate through the graph's static buffers.  Every rank
        # captures and replays the same graph, so the collectives inside line up as they do in the eager loop.
        watchdog = None
        if dist.is_initialized():
            # a capture or replay that hangs inside a collective must not cost the run its (already measured) eager number: after 240 s
            # rank 0 prints the line with the eager rate and every rank leaves
            import threading

            def eager_line():
                return json.dumps({"metric": "train_images_per_sec", "value": round(train_ips, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
                                      "warmup": args.warmup, "ms_per_step": round(1e3 * world * B / train_ips, 3), "higher_is_better": True, "scaling": "weak",
                                      "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                                      "config": {"workload": f"{args.model}, {S}x{S}, {args.dtype}, batch {B}/GPU, train step = fwd+loss+bwd+clip+SGD+EMA",
                                                 "global_batch": world * B, "imgsz": S, "parallelism": f"dp{world}"},
                                      "train_mode": "eager", "note": "the hipGraph leg of the data-parallel step did not finish (240 s limit or a fatal signal): eager rate only"})

            def bail():
                if rank == 0:
                    print(eager_line(), flush=True)
                os._exit(0)
            # ... and a leg that KILLS the process (a fatal signal inside the capture) must not either: tools/crash_line.c prints the same line
            # from the signal handler on rank 0 and leaves quietly on the others
            crash_guard = None
            try:
                import ctypes
                crash_guard = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "libcrash_line.so"))
                sys.stdout.flush()
                crash_guard.crash_line_arm(eager_line().encode() if rank == 0 else b"", 0)
            except OSError:
                crash_guard = None
            watchdog = threading.Timer(240.0, bail)
            watchdog.daemon = True
            watchdog.start()
        try:
            from yolov10_3d_amd.graph import GraphedTrainStep
            opt.zero_grad(set_to_none=True)
            gstep = GraphedTrainStep(model, opt, batches[0], max_norm=10.0, reducer=reducer)
            for j in range(2):
                gstep(batches[j % NBATCH])
                if ema is not None:
                    ema.update(model, guard=opt.last_norm)
            sync()
            tg = time.perf_counter()
            for j in range(args.steps):
                gl, gitems = gstep(batches[j % NBATCH])
                if ema is not None:
                    ema.update(model, guard=opt.last_norm)
            sync()
            tg = time.perf_counter() - tg
            tgm = torch.tensor([tg], device=dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(tgm, op=dist.ReduceOp.MAX)
            tg = float(tgm)
            assert torch.isfinite(gitems).all(), f"non-finite loss items from the graphed step {gitems}"
            train_graph_ips = world * B * args.steps / tg
            log(f"graph-replayed train step: {1e3 * tg / args.steps:.2f} ms/step ({train_graph_ips:.1f} images/s)")
            opt.zero_grad(set_to_none=True)
        except Exception as e:
            log(f"hipGraph capture of the training step failed ({type(e).__name__}: {e})")
        if watchdog is not None:
            watchdog.cancel()
            if crash_guard is not None:
                crash_guard.crash_line_disarm()

    # inference leg: eval forward + NMS-free top-k postprocess (reference validator.py:178,190: "Speed: ... ms per image").  The forward
    # is ~200 launches of 3-150 us: enqueued eagerly the host is the bound (tools/eval_audit.py), so the timed loop replays ONE captured
    # hipGraph per batch (yolov10-3d_amd/graph.py; --infer-mode eager keeps the launch-by-launch loop); the eager rate is reported next to it
    model.eval()
    infer_ips = infer_eager_ips = infer_val_ips = None
    infer_mode = args.infer_mode
    roof_infer = None
    nc = model.yaml["nc"]

    def infer_once(img):
        y = model(img)["one2one"][0]
        if is3d:
            return v10_3Dpostprocess(y.permute(0, 2, 1), 50, nc)
        return v10postprocess(y.permute(0, 2, 1), 300, nc)

    def timed_infer(fn, n):
        sync()
        t1 = time.perf_counter()
        for _ in range(n):
            fn()
        sync()
        ti = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(ti, op=dist.ReduceOp.MAX)
        return world * B * n / float(ti)

    if args.infer_steps > 0:
        with torch.no_grad():
            for _ in range(2):
                infer_once(batch["img"])
            # dominant eval kernel, timed launch by launch with HIP events in an eager pass (a graph replay has no per-kernel events)
            ekeys = {}
            ops.TIMER = ops.KernelTimer(lambda key: ekeys.setdefault(key, 0) is None)
            infer_once(batch["img"])
            ops.TIMER = None
            ecands = [k for k in ekeys if k[0] == "conv_eval"]
            ekey = max(ecands, key=lambda k: (conv_key_flops(k), k)) if ecands else None
            if ekey is not None:
                ops.TIMER = ops.KernelTimer(lambda key: key == ekey)
                for _ in range(3):
                    infer_once(batch["img"])
                etimes = ops.TIMER.results().get(ekey, [])
                ops.TIMER = None
                if etimes:
                    ems = sum(etimes) / len(etimes)
                    eflops = conv_key_flops(ekey)
                    peak = MFMA_BF16_DENSE_PEAK_TFLOPS if dtype == torch.bfloat16 else MFMA_F32_PEAK_TFLOPS
                    roof_infer = {"bound": "mfma", "kernel": "eval conv + folded BatchNorm + SiLU, %dx%d s%d %d->%d (%d groups) on %d maps of %dx%d, pad %d (largest eval launch by FLOPs)"
                                  % (ekey[7], ekey[7], ekey[8], ekey[5], ekey[6], ekey[9], ekey[2], ekey[3], ekey[4], ekey[10]),
                                  "achieved": round(eflops / (ems * 1e-3) / 1e12, 2), "peak": peak, "unit": "TFLOP/s",
                                  "frac": round(eflops / (ems * 1e-3) / 1e12 / peak, 4), "launches_timed": len(etimes), "avg_launch_ms": round(ems, 4),
                                  "flops_per_launch": eflops, "traffic": None}
            infer_eager_ips = timed_infer(lambda: infer_once(batch["img"]), args.infer_steps)
            infer_ips = infer_eager_ips
            if infer_mode == "graph":
                try:
                    from yolov10_3d_amd.graph import GraphedForward
                    gf = GraphedForward(infer_once, batch["img"])
                    gf(gf.inputs[0])
                    infer_ips = timed_infer(lambda: gf(gf.inputs[0]), args.infer_steps)
                except Exception as e:  # a capture failure must not cost the run its train number
                    log(f"hipGraph capture of the eval forward failed ({type(e).__name__}: {e}); reporting the eager rate")
                    infer_mode = "eager"
            # the reference validates DURING training with twice the training batch (engine/trainer.py:297: test_loader batch_size * 2):
            # the same leg at that batch, reported next to the batch-B figure (which stays `infer_images_per_sec`, comparable across rounds)
            if infer_mode == "graph" and world == 1 and not args.no_val_batch:
                try:
                    img2 = torch.cat((batch["img"], batches[1 % NBATCH]["img"]), 0)
                    gf2 = GraphedForward(infer_once, img2)
                    gf2(gf2.inputs[0])
                    infer_val_ips = 2 * timed_infer(lambda: gf2(gf2.inputs[0]), args.infer_steps)
                    del gf2, img2
                except Exception as e:
                    log(f"eval leg at the reference's validation batch failed ({type(e).__name__}: {e})")
    log(f"infer: {infer_ips:.1f} images/s ({infer_mode}; eager {infer_eager_ips:.1f}{'; at the validation batch %d: %.1f' % (2 * B, infer_val_ips) if infer_val_ips else ''})"
        if infer_ips else "infer: skipped")

    if rank == 0:
        res = timer.results().get(k1_key, []) if k1_key is not None else []
        roof = None
        if res:
            avg_ms = sum(res) / len(res)
            _, _, kB, kH, kW, kCin, kCout, kk, ks, kg = k1_key
            flops = conv_key_flops(k1_key)
            ach = flops / (avg_ms * 1e-3) / 1e12
            is_f8 = k1_key[0] == "conv_fwd_fp8"
            peak = MFMA_FP8_DENSE_PEAK_TFLOPS if is_f8 else (MFMA_BF16_DENSE_PEAK_TFLOPS if dtype == torch.bfloat16 else MFMA_F32_PEAK_TFLOPS)
            # HBM bytes per launch of this kernel: PMC counters cannot be read from inside the run, so this is the figure of the
            # committed rocprofv3 passes over this same command (tools/pmc_summary.py), labelled with the commit they were taken at
            traffic, traffic_src = None, None
            pj = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_headline.json")
            default_cfg = args.model == "yolov10s_3D.yaml" and dtype == torch.bfloat16 and B == 32 and S == 640 and args.weights == "full"
            if os.path.exists(pj) and default_cfg:
                with open(pj) as f:
                    pm = json.load(f)
                traffic = pm.get("traffic_bytes_per_launch")
                traffic_src = "profiles/pmc_headline.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes at commit %s)" % pm.get("commit", "?")
            what = "fused head layer 2" if kg == 16 else "largest 3x3 s1 forward launch of the step"
            roof = {"bound": "mfma", "kernel": "3x3 s1 conv forward %s, %s @%dx%d B=%d (%s)"
                    % ("fp8 e4m3 x e4m3, MX block scales, v_mfma_scale_f32_16x16x128_f8f6f4 (persistent resident-halo kernel, csrc/conv3x3_fp8.hip)" if is_f8
                       else args.dtype + " (persistent resident-halo implicit GEMM, csrc/conv3x3_wide3.hip)", ("%d groups of %d->%d" % (kg, kCin // kg, kCout // kg)) if kg > 1 else "%d->%d" % (kCin, kCout), kH, kW, kB, what),
                    "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "launches_timed": len(res), "avg_launch_ms": round(avg_ms, 4), "flops_per_launch": flops}
        cpu = None
        if not args.no_cpu_baseline:
            log("cpu baseline (oracle restatement on the host cores) ...")
            cpu = cpu_baseline(args.model, S, seed=1)
            log(f"cpu baseline: {cpu['value']:.3f} images/s on {cpu['cores']} cores")
        # the step launched kernel by kernel and the SAME step replayed from one hipGraph were both timed over --steps steps:
        # the headline is the faster one (the roofline launch and the phases are always measured in the eager loop)
        best_ips, train_mode = (train_graph_ips, "hipgraph") if (train_graph_ips and train_graph_ips > train_ips) else (train_ips, "eager")
        out = {
            "metric": "train_images_per_sec", "value": round(best_ips, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * world * B / best_ips, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": ("fp8" if y3d.fp8_conv() else args.dtype), "data": "synthetic",
            "config": {"workload": f"{args.model}{' (3D head)' if is3d else ' (2D head)'}, {S}x{S}, {args.dtype}{'' if args.weights == 'full' else (' + fp8: e4m3 conv weights; the 3x3 stride-1 convolutions the fp8 kernel serves (both head layers) run e4m3 x e4m3 with MX block-scaled activations on v_mfma_scale_f32_16x16x128_f8f6f4 in the forward, bf16 in the backward' if y3d.fp8_conv() else ' MFMA on fp8 e4m3-VALUED conv weights (--fp8-emulate: weight-format emulation, no fp8 MFMA instruction runs)')}, batch {B}/GPU, train step = fwd+loss+bwd+clip+SGD+EMA",
                       "global_batch": world * B, "imgsz": S, "parallelism": f"dp{world}"},
            "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 1, "backend": args.backend if dist.is_initialized() else None,
            "steps_skipped_nonfinite": int(opt.last_norm[4]) if opt.last_norm is not None else None,
            "train_mode": train_mode, "train_images_per_sec_eager": round(train_ips, 2),
            "train_images_per_sec_graph": round(train_graph_ips, 2) if train_graph_ips else None,
            "infer_images_per_sec": round(infer_ips, 2) if infer_ips else None, "infer_mode": infer_mode if infer_ips else None,
            "infer_images_per_sec_eager": round(infer_eager_ips, 2) if infer_eager_ips else None,
            "infer_images_per_sec_val_batch": round(infer_val_ips, 2) if infer_val_ips else None, "infer_val_batch": 2 * B if infer_val_ips else None,
            "roofline_infer": roof_infer,
            "loss_items": [round(float(v), 5) for v in items.float().cpu()],
            "roofline": roof, "cpu_baseline": cpu,
        }
        if args.phases and phase_ev:
            def med(i, j):
                v = sorted(e[i].elapsed_time(e[j]) for e in phase_ev)
                return round(v[len(v) // 2], 3)
            out["phases_ms"] = {"fwd_loss_bwd": med(0, 1), "allreduce_exposed": med(1, 2), "clip_sgd_ema": med(2, 3)}
            bt = reducer.times() if reducer is not None else None
            if bt:
                out["phases_ms"]["buckets"] = [{"MB": round(nb / 1e6, 1), "ready_to_done_ms": round(a, 3), "allreduce_ms": round(b, 3)} for nb, a, b in bt]
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
