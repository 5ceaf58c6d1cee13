"""Which torch ops still launch copy / elementwise kernels inside the training step (they should be few): op counts with shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import yolov10_3d_amd as y3d
from yolov10_3d_amd.optim import build_optimizer
from bench import synth_batch

dev = torch.device("cuda", 0)
y3d.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
model = y3d.YOLOv10_3DDetectionModel("yolov10s_3D.yaml").to(dev).train()
opt = build_optimizer(model)
model.model[-1].restack()
batch = synth_batch(32, 640, 640, 1, dev)


def step():
    loss, _ = model(batch)
    loss.backward()
    opt.step(max_norm=10.0)
    opt.zero_grad()


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    step()
    torch.cuda.synchronize()
rows = {}
for e in prof.events():
    if e.name.startswith("aten::") and e.name not in ("aten::empty", "aten::empty_strided", "aten::view", "aten::as_strided", "aten::slice", "aten::select",
                                                       "aten::detach", "aten::reshape", "aten::permute", "aten::alias", "aten::_unsafe_view", "aten::narrow",
                                                       "aten::unsqueeze", "aten::squeeze", "aten::transpose", "aten::expand", "aten::t", "aten::empty_like"):
        k = (e.name, str(e.input_shapes)[:90])
        r = rows.setdefault(k, [0, 0.0])
        r[0] += 1
        r[1] += e.device_time_total if hasattr(e, "device_time_total") else 0.0
for (name, shp), (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"{n:5d}  {t / 1e3:8.3f} ms  {name:28s} {shp}")
