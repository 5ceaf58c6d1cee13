"""A/B loop of the round-3 experiment "weight gradients on a second stream" (measured, not kept - the `ops.WGRAD_OVERLAP`
switch it toggles went with the experiment, so at HEAD both settings time the same step):  python tools/probe/wg_overlap.py [yaml] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import yolov10_3d_amd as y3d
from yolov10_3d_amd import ops
from yolov10_3d_amd.optim import build_optimizer
from bench import synth_batch
name = sys.argv[1] if len(sys.argv) > 1 else "yolov10s_3D.yaml"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
y3d.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
model = (y3d.YOLOv10_3DDetectionModel if "3D" in name else y3d.YOLOv10DetectionModel)(name).cuda().train()
opt = build_optimizer(model)
if hasattr(model.model[-1], "restack"):
    model.model[-1].restack()
batch = synth_batch(B, 640, 640, 1, "cuda", nc=model.yaml["nc"])


def step():
    loss, _ = model(batch)
    loss.backward()
    opt.step(max_norm=10.0)
    opt.zero_grad()


for flag in (True, False, True, False):
    ops.WGRAD_OVERLAP = flag
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    print(f"overlap {flag}: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms/step", flush=True)
