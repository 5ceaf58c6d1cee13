"""Host enqueue time of one training step vs its GPU time: is the step GPU-bound or launch-bound?
    python tools/host_time.py [yaml] [imgsz] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import yolov10_3d_amd as y3d
from yolov10_3d_amd.optim import ModelEMA, build_optimizer
from bench import synth_batch

name = sys.argv[1] if len(sys.argv) > 1 else "yolov10s_3D.yaml"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 640
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device("cuda", 0)
y3d.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
model = y3d.YOLOv10_3DDetectionModel(name).to(dev).train()
opt = build_optimizer(model)
model.model[-1].restack()
ema = ModelEMA(model)
batch = synth_batch(B, S, S, 1, dev)


def step(marks=None):
    t = [time.perf_counter()]
    loss, items = model(batch); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    opt.step(max_norm=10.0); t.append(time.perf_counter())
    opt.zero_grad(); t.append(time.perf_counter())
    ema.update(model); t.append(time.perf_counter())
    if marks is not None:
        marks.append([1e3 * (b - a) for a, b in zip(t, t[1:])])


for _ in range(5):
    step()
torch.cuda.synchronize()
# (a) each step followed by a sync: host enqueue time with an EMPTY queue in front of it
marks = []
for _ in range(5):
    step(marks)
    torch.cuda.synchronize()
m = [sum(c) / len(c) for c in zip(*marks)]
print(f"host enqueue per step (queue drained before each): fwd {m[0]:.2f} bwd {m[1]:.2f} opt {m[2]:.2f} zero_grad {m[3]:.2f} ema {m[4]:.2f} = {sum(m):.2f} ms")
# (b) free running
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 20
for _ in range(N):
    step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N
print(f"free-running step: {1e3 * dt:.2f} ms")
