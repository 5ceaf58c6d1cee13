"""Per-shape timing of every conv launch (forward / data gradient / weight gradient) of the training step, with the achieved
TFLOP/s: which shapes cost the step the most.  usage (GPU box): python tools/layer_report.py [model.yaml] [imgsz] [batch] [rows] [eval]
(`eval` as fifth argument: the eval forward + postprocess instead of the training step; `fp8` as sixth: fp8 weights + fp8 MFMA convs)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import yolov10_3d_amd as y3d
from yolov10_3d_amd import ops
from yolov10_3d_amd.optim import build_optimizer
from bench import conv_key_flops, synth_batch

name = sys.argv[1] if len(sys.argv) > 1 else "yolov10s_3D.yaml"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 640
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device("cuda", 0)
y3d.set_compute_dtype(torch.bfloat16)
if "fp8" in sys.argv[5:]:
    y3d.set_weight_quant("fp8")
    y3d.set_fp8_conv(True)
torch.manual_seed(0)
model = (y3d.YOLOv10_3DDetectionModel if "3D" in name else y3d.YOLOv10DetectionModel)(name).to(dev).train()
opt = build_optimizer(model)
if hasattr(model.model[-1], "restack"):
    model.model[-1].restack()
batch = synth_batch(B, S, S, 1, dev, nc=model.yaml["nc"])


EVAL = "eval" in sys.argv[5:]
if EVAL:
    from yolov10_3d_amd.loss import v10_3Dpostprocess, v10postprocess
    model.eval()


def step():
    if EVAL:
        with torch.no_grad():
            y = model(batch["img"])["one2one"][0]
            return (v10_3Dpostprocess if "3D" in name else v10postprocess)(y.permute(0, 2, 1), 50 if "3D" in name else 300, model.yaml["nc"])
    loss, _ = model(batch)
    loss.backward()
    opt.step(max_norm=10.0)
    opt.zero_grad()


for _ in range(2):
    step()
ops.TIMER = ops.KernelTimer(lambda key: True)
N = 3
for _ in range(N):
    step()
res = ops.TIMER.results()
ops.TIMER = None
rows = []
for key, ts in res.items():
    kind, dt, b, h, w, cin, cout, k, s, g = key[:10]
    fl = conv_key_flops(key)  # output size from the launch's own padding (the sparse head's patch convs run unpadded: 5x5 -> 3x3)
    ms = sum(ts) / N  # per step (all launches of this shape)
    rows.append((ms, kind, (h, w, cin, cout, k, s, g), len(ts) / N, fl * len(ts) / N / (ms * 1e-3) / 1e12))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"{name} {S}x{S} B={B}: conv launches {tot:.2f} ms/step (timed with events, includes launch gaps)")
for ms, kind, shp, n, tf in rows[:int(sys.argv[4]) if len(sys.argv) > 4 else 400]:
    print(f"{ms:7.3f} ms/step  {kind:13s} HxW={shp[0]:3d}x{shp[1]:<3d} {shp[2]:5d}->{shp[3]:<5d} k{shp[4]} s{shp[5]} g{shp[6]:<3d} n={n:4.1f}  {tf:7.1f} TFLOP/s")
