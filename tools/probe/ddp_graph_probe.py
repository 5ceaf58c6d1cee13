"""one-rank RCCL rehearsal of the captured N > 1 step: prints where a capture fails.  Y3D_FORCE_DDP=1 python tools/probe/ddp_graph_probe.py"""
import faulthandler, os, sys, traceback, torch
faulthandler.enable()
os.environ.setdefault("Y3D_FORCE_DDP", "1")
sys.path.insert(0, ".")
import yolov10_3d_amd as y3d
from yolov10_3d_amd import ddp
from yolov10_3d_amd.graph import GraphedTrainStep
from yolov10_3d_amd.optim import build_optimizer
from bench import synth_batch
dev = torch.device("cuda", 0)
if not os.environ.get('Y3D_PROBE_NOINIT'):
    ddp.init("nccl", dev)
torch.manual_seed(0)
model = y3d.YOLOv10_3DDetectionModel("yolov10n_3D.yaml").to(dev).train()
opt = build_optimizer(model)
model.model[-1].restack()
red = ddp.FlatGradReducer(model.parameters(), overlap=not os.environ.get('Y3D_PROBE_NOHOOKS'))
red.broadcast_parameters(model)
if os.environ.get('Y3D_PROBE_NOCOLL'):
    red.always_collective = False
if os.environ.get('Y3D_PROBE_NOOVERLAP'):
    for h in red._hooks:
        h.remove()
    red._hooks, red.overlap = [], False
b = [synth_batch(4, 256, 256, s, dev) for s in (1, 2, 3)]
state = {k: v.clone() for k, v in model.state_dict().items()}
print('eager reducer step ...', flush=True)
loss, _ = model(b[0]); loss.backward(); red.finish(); opt.step(max_norm=10.0); opt.zero_grad(); torch.cuda.synchronize()
del loss, _  # an autograd graph kept alive keeps its AccumulateGrad nodes (made on the NULL stream) alive: the capture would have to touch that stream
model.load_state_dict(state); opt._state['flat'].zero_()
print('capture ...', flush=True)
try:
    step = GraphedTrainStep(model, opt, b[0], reducer=None if os.environ.get('Y3D_PROBE_NORED') else red)
    print('captured; replaying', flush=True)
    for x in b:
        step(x)
    torch.cuda.synchronize()
    g_state = {k: v.clone() for k, v in model.state_dict().items()}
    model.load_state_dict(state); opt._state["flat"].zero_(); opt.zero_grad()
    for x in b:
        loss, _ = model(x); loss.backward(); red.finish(); opt.step(max_norm=10.0); opt.zero_grad()
    torch.cuda.synchronize()
    bad = [(k, float((v.float() - g_state[k].float()).abs().max())) for k, v in model.state_dict().items()
           if not torch.equal(v, g_state[k]) and not k.endswith("num_batches_tracked")]  # (the probe's own warm step is still pending in those counters)
    print("captured reducer step == eager reducer step:", not bad, len(bad), bad[:6])
    sys.exit(1 if bad else 0)
except Exception:
    traceback.print_exc()
    sys.exit(2)
