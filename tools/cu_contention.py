"""One-GPU bound for the RCCL-footprint caveat of DESIGN §5: the backward pass's conv kernels are persistent
one-workgroup-per-CU grids; an RCCL all-reduce kernel resident on some CUs during the backward delays the workgroups the dispatcher
would have put there.  This runs the training step with a DUMMY resident kernel (N workgroups of 256 threads spinning on s_sleep +
a trickle of memory traffic, on a side stream) alive during every backward pass, and reports the step time against the undisturbed
step: an upper bound of what an N-CU collective costs the overlapped design.
    python tools/cu_contention.py [N ...]        (GPU box; default N = 8 16 32)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import yolov10_3d_amd as y3d
from yolov10_3d_amd.optim import build_optimizer
from bench import synth_batch

dev = torch.device("cuda", 0)
y3d.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
model = y3d.YOLOv10_3DDetectionModel("yolov10s_3D.yaml").to(dev).train()
opt = build_optimizer(model)
model.model[-1].restack()
batches = [synth_batch(32, 640, 640, 1 + j, dev) for j in range(4)]
L = y3d.lib()
side = torch.cuda.Stream()
flag = torch.zeros(1, dtype=torch.int32, device=dev)
buf = torch.zeros(64 << 20, dtype=torch.float32, device=dev)


def run(ncu, steps=12):
    ts = []
    for i in range(steps + 3):
        bt = batches[i % 4]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss, _ = model(bt)
        if ncu:
            flag.zero_()
            side.wait_stream(torch.cuda.current_stream())
            L.occupy_cus(ncu, flag.data_ptr(), buf.data_ptr(), buf.numel(), side.cuda_stream)  # resident until the flag is set
        loss.backward()
        if ncu:
            flag.fill_(1)  # on the compute stream, after the backward: releases the resident workgroups
            torch.cuda.current_stream().wait_stream(side)
        opt.step(max_norm=10.0)
        opt.zero_grad()
        torch.cuda.synchronize()
        if i >= 3:
            ts.append(time.perf_counter() - t0)
    ts.sort()
    return 1e3 * ts[len(ts) // 2]


base = run(0)
print(f"undisturbed step: {base:.2f} ms")
for n in [int(a) for a in sys.argv[1:]] or [8, 16, 32]:
    t = run(n)
    print(f"{n:3d} workgroups resident during the backward: {t:.2f} ms/step  (+{100 * (t / base - 1):.1f} %)")
