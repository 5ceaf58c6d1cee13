"""eval forward + postprocess loop (for rocprofv3):  python tools/infer_loop.py [iters]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolov10_3d_amd as y3d
import bench
from yolov10_3d_amd.loss import v10_3Dpostprocess
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
y3d.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
model = y3d.YOLOv10_3DDetectionModel("yolov10s_3D.yaml").cuda().eval()
batch = bench.synth_batch(32, 640, 640, 1, "cuda")
with torch.no_grad():
    for _ in range(3):
        y = model(batch["img"])["one2one"][0]
        v10_3Dpostprocess(y.permute(0, 2, 1), 50, 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        y = model(batch["img"])["one2one"][0]
        v10_3Dpostprocess(y.permute(0, 2, 1), 50, 3)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"{32 * n / dt:.1f} images/s, {1e3 * dt / n:.2f} ms per batch of 32")
