"""Run the BASELINE.json parity-case configurations through train steps + an eval pass on the GPU and report time and finiteness.
usage: python tools/config_check.py yolov10m_3D.yaml:640:32 yolov10l.yaml:1280:4 ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import yolov10_3d_amd as y3d
from yolov10_3d_amd.optim import build_optimizer
from yolov10_3d_amd.loss import v10_3Dpostprocess, v10postprocess
from bench import synth_batch

dev = torch.device("cuda", 0)
y3d.set_compute_dtype(torch.bfloat16)
for spec in sys.argv[1:]:
    name, S, B = spec.split(":")
    S, B = int(S), int(B)
    is3d = "3D" in name
    torch.manual_seed(0)
    model = (y3d.YOLOv10_3DDetectionModel if is3d else y3d.YOLOv10DetectionModel)(name).to(dev).train()
    nc = model.yaml["nc"]
    opt = build_optimizer(model)
    if hasattr(model.model[-1], "restack"):
        model.model[-1].restack()
    batch = synth_batch(B, S, S, 1, dev, nc=nc)
    ts = []
    for i in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss, items = model(batch)
        loss.backward()
        opt.step(max_norm=10.0)
        opt.zero_grad()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    assert torch.isfinite(items).all(), items
    model.eval()
    with torch.no_grad():
        y = model(batch["img"])
        y = y["one2one"][0] if isinstance(y, dict) else y
        out = (v10_3Dpostprocess(y.permute(0, 2, 1), 50, nc) if is3d else v10postprocess(y.permute(0, 2, 1), 300, nc))
    torch.cuda.synchronize()
    print(f"{name} {S}x{S} B={B}: {sum(p.numel() for p in model.parameters())/1e6:.1f} M params, train {1e3*min(ts):.1f} ms/step = {B/min(ts):.0f} img/s, "
          f"loss items {[round(float(v), 3) for v in items.float().cpu()][:6]}, eval out {[tuple(o.shape) for o in out]}, mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
    del model, opt, batch
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
