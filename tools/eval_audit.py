"""What one steady-state EVAL forward + postprocess launches besides the y3d kernels: aten ops (with stack), memcpys, and the host
time of the forward.     python tools/eval_audit.py [model yaml] [batch]        (GPU box)"""
import collections, os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import yolov10_3d_amd as y3d
from yolov10_3d_amd.loss import v10_3Dpostprocess, v10postprocess
from bench import synth_batch

name = sys.argv[1] if len(sys.argv) > 1 else "yolov10s_3D.yaml"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda", 0)
y3d.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
is3d = "3D" in name
model = (y3d.YOLOv10_3DDetectionModel if is3d else y3d.YOLOv10DetectionModel)(name).to(dev).eval()
nc = model.yaml["nc"]
batch = synth_batch(B, 640, 640, 1, dev, nc=nc)


def fwd():
    with torch.no_grad():
        y = model(batch["img"])["one2one"][0]
        return (v10_3Dpostprocess(y.permute(0, 2, 1), 50, nc) if is3d else v10postprocess(y.permute(0, 2, 1), 300, nc))


for _ in range(3):
    fwd()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    fwd()
th = time.perf_counter() - t0
torch.cuda.synchronize()
tw = time.perf_counter() - t0
print(f"host enqueue {1e3 * th / 10:.2f} ms per forward, wall {1e3 * tw / 10:.2f} ms per forward ({B * 10 / tw:.0f} images/s)")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    fwd()
    torch.cuda.synchronize()
rows = collections.OrderedDict()
for e in prof.events():
    if e.name.startswith("aten::") and e.name not in ("aten::empty", "aten::empty_strided", "aten::view", "aten::permute", "aten::as_strided", "aten::slice",
                                                        "aten::select", "aten::reshape", "aten::detach", "aten::empty_like", "aten::_unsafe_view", "aten::alias",
                                                        "aten::split", "aten::split_with_sizes", "aten::narrow", "aten::unsqueeze", "aten::expand", "aten::t",
                                                        "aten::transpose", "aten::_reshape_alias", "aten::resolve_conj", "aten::resolve_neg", "aten::result_type",
                                                        "aten::lift_fresh", "aten::is_pinned", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::chunk"):
        st = [f for f in (e.stack or []) if "yolov10" in f or "bench" in f or "tools" in f]
        k = (e.name, str(e.input_shapes)[:70], st[0][-70:] if st else "")
        r = rows.setdefault(k, [0, 0.0])
        r[0] += 1
        r[1] += getattr(e, "device_time_total", 0.0)
print("---- aten ops of one eval forward (views excluded) ----")
for (nm, shp, st), (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][0])[:50]:
    print(f"{n:4d} {t:8.1f} us  {nm:22s} {shp:70s} {st}")
kern = collections.Counter()
for e in prof.events():
    if e.device_type is not None and str(e.device_type).endswith("CUDA"):
        kern[e.name[:60]] += 1
print("---- device activities ----")
for k, n in kern.most_common(12):
    print(f"{n:4d}  {k}")
