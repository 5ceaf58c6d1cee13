"""Round-4 probe: replays of graph.GraphedTrainStep from ONE restored state must agree with each other (head maps, assignment, loss items).
They did not: hipMemsetAsync nodes recorded into the big graph (the zeroing of the assigner's atomicMax accumulators) took no effect on
later replays, so maxima of the previous replay leaked in.  The library now zeroes with kernels (tal_loss3d.hip: gt_prep_kernel)."""
import copy, sys, torch
sys.path.insert(0, ".")
import yolov10_3d_amd as y3d
from bench import synth_batch
from yolov10_3d_amd import ops
from yolov10_3d_amd.graph import GraphedTrainStep
from yolov10_3d_amd.optim import build_optimizer
DEV = "cuda"
y3d.set_compute_dtype(torch.bfloat16)
torch.manual_seed(3)
model = y3d.YOLOv10_3DDetectionModel("yolov10n_3D.yaml").to(DEV).train()
opt = build_optimizer(model, lr=0.01)
model.model[-1].restack()
b0, b1 = synth_batch(2, 256, 256, 31, DEV), synth_batch(2, 256, 256, 32, DEV)
# keep the graph's head maps: the loss receives them as preds["_y3d_maps"]
kept = {}
crit_call = type(model.init_criterion()).__call__
def spy(self, preds, batch):
    kept["maps"] = list(preds["_y3d_maps"])
    return crit_call(self, preds, batch)
type(model.init_criterion()).__call__ = spy
step = GraphedTrainStep(model, opt, b0)
tens = list(model.parameters()) + list(model.buffers())
a1 = model.criterion.one2one._assignment
am = model.criterion.one2many._assignment
maps = kept["maps"]
n_used = step.counts[0][0]

def snap():
    return [t.detach().clone() for t in tens], opt._state["flat"].clone(), opt._state["norm_clip"].clone()
def restore(s):
    with torch.no_grad():
        for t, v in zip(tens, s[0]):
            t.copy_(v)
        opt._state["flat"].copy_(s[1]); opt._state["norm_clip"].copy_(s[2])
    ops.bump_weight_epoch()
s0 = snap()
outs = []
for r in range(3):
    restore(s0)
    _, it = step(b0)
    torch.cuda.synchronize()
    outs.append(dict(items=it.clone(), maps=[m.clone() for m in maps], a1=[t.clone() for t in a1], am=[t.clone() for t in am], n=n_used.clone()))
for r in (1, 2):
    o, p = outs[0], outs[r]
    print(f"replay {r} vs 0: items {bool(torch.equal(o['items'], p['items']))} maps {[bool(torch.equal(a, b)) for a, b in zip(o['maps'], p['maps'])]} "
          f"o2o fg/gi/ts {[bool(torch.equal(a, b)) for a, b in zip(o['a1'], p['a1'])]} o2m {[bool(torch.equal(a, b)) for a, b in zip(o['am'], p['am'])]} n_used {o['n'].tolist()} {p['n'].tolist()}")
    print("    o2o positives", int(o['a1'][0].sum()), int(p['a1'][0].sum()), " ts sum", float(o['a1'][2].sum()), float(p['a1'][2].sum()))
