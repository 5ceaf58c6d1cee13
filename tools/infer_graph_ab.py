"""captured eval forward + postprocess, detection levels on one stream vs one hipGraph branch per level (ops.EVAL_LEVEL_STREAMS):
python tools/infer_graph_ab.py [batch] [replays]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolov10_3d_amd as y3d
import bench
from yolov10_3d_amd import ops
from yolov10_3d_amd.graph import GraphedForward
from yolov10_3d_amd.loss import v10_3Dpostprocess

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
y3d.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
model = y3d.YOLOv10_3DDetectionModel("yolov10s_3D.yaml").cuda().eval()
img = bench.synth_batch(B, 640, 640, 1, "cuda")["img"]


def once(im):
    y = model(im)["one2one"][0]
    return v10_3Dpostprocess(y.permute(0, 2, 1), 50, 3)


outs = {}
for rnd in range(2):
    for flag in (False, True):
        ops.EVAL_LEVEL_STREAMS = flag
        g = GraphedForward(once, img)
        for _ in range(5):
            g(g.inputs[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            o = g(g.inputs[0])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        outs[flag] = [t.clone() for t in o]
        print(f"level streams {flag!s:5}: {B * n / dt:9.1f} images/s, {1e3 * dt / n:.3f} ms per batch of {B}", flush=True)
        del g
print("outputs identical:", all(torch.equal(a, b) for a, b in zip(outs[False], outs[True])))
