"""cProfile of the host side of the eager training step (where do the ~19 us per launch go?)   python tools/probe/host_profile.py   (GPU box)"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import yolov10_3d_amd as y3d
from yolov10_3d_amd.optim import ModelEMA, build_optimizer
from bench import synth_batch
y3d.set_compute_dtype(torch.bfloat16)
model = y3d.YOLOv10_3DDetectionModel("yolov10s_3D.yaml").cuda().train()
opt = build_optimizer(model)
model.model[-1].restack()
batch = synth_batch(32, 640, 640, 1, "cuda")


def step():
    loss, _ = model(batch)
    loss.backward()
    opt.step(max_norm=10.0)
    opt.zero_grad()


for _ in range(4):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
    torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
