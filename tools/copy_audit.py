"""Which host-side conversions still launch copy / add kernels inside the training step: every `ops.to_nhwc` / `ops._dense_any`
call that had to COPY (with its caller, shape and strides), and the aten ops autograd's gradient accumulation adds.
    python tools/copy_audit.py        (GPU box)"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import yolov10_3d_amd as y3d
from yolov10_3d_amd import ops
from yolov10_3d_amd.optim import build_optimizer
from bench import synth_batch

dev = torch.device("cuda", 0)
y3d.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
model = y3d.YOLOv10_3DDetectionModel("yolov10s_3D.yaml").to(dev).train()
opt = build_optimizer(model)
model.model[-1].restack()
batch = synth_batch(32, 640, 640, 1, dev)
log = collections.Counter()
orig_to, orig_dense = ops.to_nhwc, ops._dense_any


def caller():
    st = traceback.extract_stack(limit=6)[:-2]
    return " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in reversed(st[-3:]))


def to_nhwc(x, dtype=None, dense=False):
    out = orig_to(x, dtype, dense)
    if out is not x:
        log[("to_nhwc", tuple(x.shape), tuple(x.stride()), str(x.dtype), dense, caller())] += 1
    return out


def dense_any(x, dtype):
    log[("_dense_any", tuple(x.shape), tuple(x.stride()), str(x.dtype), True, caller())] += 1
    return orig_dense(x, dtype)


def step():
    loss, _ = model(batch)
    loss.backward()
    opt.step(max_norm=10.0)
    opt.zero_grad()


for _ in range(2):
    step()
ops.to_nhwc, ops._dense_any = to_nhwc, dense_any
step()
ops.to_nhwc, ops._dense_any = orig_to, orig_dense
torch.cuda.synchronize()
print("---- conversions that copied (one step) ----")
for k, n in sorted(log.items(), key=lambda kv: -kv[1]):
    print(n, k)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = {}
for e in prof.events():
    if e.name in ("aten::add", "aten::add_", "aten::copy_", "aten::mul", "aten::cat", "aten::fill_", "aten::zero_", "aten::sum", "aten::clone", "aten::contiguous"):
        k = (e.name, str(e.input_shapes)[:100])
        r = rows.setdefault(k, [0, 0.0])
        r[0] += 1
        r[1] += getattr(e, "device_time_total", 0.0)
print("---- aten ops of one step ----")
for (name, shp), (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{n:5d}  {t / 1e3:8.3f} ms  {name:14s} {shp}")
