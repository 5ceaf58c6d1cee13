"""which concat members are still COPIED (y3d_copy2d) in one eval / training forward: pixels, channels, caller   (GPU box)"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import yolov10_3d_amd as y3d
from yolov10_3d_amd import ops
from bench import synth_batch
y3d.set_compute_dtype(torch.bfloat16)
model = y3d.YOLOv10_3DDetectionModel("yolov10s_3D.yaml").cuda()
batch = synth_batch(32, 640, 640, 1, "cuda")
L = ops.lib()
orig = L.copy2d
log = collections.Counter()


def copy2d(dt, x, xsw, y, ysw, P, C, st):
    fr = traceback.extract_stack(limit=8)[:-1]
    who = " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr[-5:]))
    log[(P, C, xsw, ysw, who)] += 1
    return orig(dt, x, xsw, y, ysw, P, C, st)


for mode in ("eval", "train"):
    model.train(mode == "train")
    for it in range(2):
        if it == 1:
            L.copy2d = copy2d
        if mode == "eval":
            with torch.no_grad():
                model(batch["img"])
        else:
            model(batch)[0].backward()
    L.copy2d = orig
    print(f"---- {mode}: {sum(log.values())} copies, {sum(k[0] * k[1] * 2 * v for k, v in log.items()) / 1e6:.0f} MB")
    for k, v in sorted(log.items(), key=lambda kv: -kv[0][0] * kv[0][1]):
        print(f"{v:3d} x  P={k[0]:8d} C={k[1]:4d} xsw={k[2]:4d} ysw={k[3]:4d}  {k[4]}")
    log.clear()
