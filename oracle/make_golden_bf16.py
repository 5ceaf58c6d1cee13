"""TEST INFRASTRUCTURE — mints the 16-bit yardstick from the REFERENCE (build container only).

    python -m oracle.make_golden_bf16        # writes tests/golden/autocast_bf16.npz

The reference trains under `torch.cuda.amp.autocast` (engine/trainer.py:395): its own 16-bit path is the honest yardstick for
this repo's bf16 performance mode.  For every module fixture of oracle/make_golden.py (same explicit weights, same inputs, read
back from tests/golden/<name>.npz) the reference module is run once more under CPU `torch.autocast(dtype=bfloat16)` and its
outputs / gradients are stored; likewise the tiny end-to-end 3D model (head maps, loss items, selected gradients).  The GPU
tests then hold  dist(HIP-bf16, fp32 golden) <= 1.5 x dist(reference-autocast, fp32 golden)  (tests/test_hip_bench_path.py).
Fixtures are data only: no reference source text is stored.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_shim as R  # noqa: E402
from oracle.make_golden import OUT, synth_batch, tiny_cfg  # noqa: E402,F401


def load(name):
    z = np.load(os.path.join(OUT, name + ".npz"))
    flat = {k: torch.from_numpy(z[k]) for k in z.files}
    out = {}
    for k, v in flat.items():
        if "/" in k:
            a, b = k.split("/", 1)
            out.setdefault(a, {})[b] = v
        else:
            out[k] = v
    return out


def main():
    R.import_reference()
    from ultralytics.nn.modules import Conv, C2f, C2fCIB, SCDown, SPPF, PSA

    mods = {
        "conv_k1": lambda: Conv(16, 24, 1, 1), "conv_k3s1": lambda: Conv(16, 24, 3, 1), "conv_k3s2": lambda: Conv(16, 24, 3, 2),
        "conv_dw3": lambda: Conv(16, 16, 3, 1, None, 16), "conv_dw7": lambda: Conv(16, 16, 7, 1, 3, 16, 1, False),
        "c2f_shortcut": lambda: C2f(32, 32, 2, True), "c2f_neck": lambda: C2f(48, 32, 1, False),
        "c2fcib_lk": lambda: C2fCIB(32, 32, 1, True, True), "c2fcib": lambda: C2fCIB(32, 32, 1, True, False),
        "scdown": lambda: SCDown(16, 32, 3, 2), "sppf": lambda: SPPF(32, 32, 5),
        "psa_1head": lambda: PSA(128, 128), "psa_2head": lambda: PSA(256, 256),
    }
    arrs = {}
    for name, ctor in mods.items():
        g = load(name)
        m = ctor()
        sd = {k[len("model.0."):]: v for k, v in g["state"].items()}
        m.load_state_dict(sd, strict=True)
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.eps, mod.momentum = 1e-3, 0.03
        m.train()
        x = g["x"].clone().requires_grad_(True)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            y = m(x)
        (y.float() * g["r"]).sum().backward()
        named = dict(m.named_parameters())
        arrs[f"{name}/y_train"] = y.detach().float().numpy()
        arrs[f"{name}/dx"] = x.grad.float().numpy()
        for k in g.get("grads", {}):
            arrs[f"{name}/grads/{k}"] = named[k[len("model.0."):]].grad.float().numpy()
        e = float((y.detach().float() - g["y_train"]).norm() / g["y_train"].norm())
        print(f"{name}: reference autocast(bf16) vs fp32 golden, relative L2 of y_train = {e:.3e}")

    # tiny end-to-end 3D model: the fixture's own weights / batch
    g = load("e2e_tiny3d_s")
    cfg = tiny_cfg("v10-3D/yolov10s_3D.yaml")
    m = R.build_model(cfg, seed=0)
    missing, unexpected = m.load_state_dict(g["state"], strict=False)
    assert not unexpected, unexpected
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eps, mod.momentum = 1e-3, 0.03
    bt = {k: v for k, v in g["batch"].items()}
    bt["img"] = g["img"]
    m.train()
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        preds = m.predict(g["img"]) if hasattr(m, "predict") else m.forward(g["img"])
    m.load_state_dict(g["state"], strict=False)  # undo the running-statistics update of that pass
    for j, t in enumerate(preds["one2many"]):
        arrs[f"e2e_tiny3d_s/o2m/{j}"] = t.float().numpy()
    for j, t in enumerate(preds["one2one"]):
        arrs[f"e2e_tiny3d_s/o2o/{j}"] = t.float().numpy()
    with R.cpu_cuda_noop(), torch.autocast("cpu", dtype=torch.bfloat16):
        loss, items = m(bt)
    loss.backward()
    named = dict(m.named_parameters())
    arrs["e2e_tiny3d_s/loss"] = loss.detach().float().numpy()
    arrs["e2e_tiny3d_s/items"] = items.detach().float().numpy()
    for k in g["grads"]:
        arrs[f"e2e_tiny3d_s/grads/{k}"] = named[k].grad.float().numpy()
    print("e2e_tiny3d_s: autocast items", [round(float(v), 4) for v in items], "fp32 items", [round(float(v), 4) for v in g["items"]])
    path = os.path.join(OUT, "autocast_bf16.npz")
    np.savez_compressed(path, **arrs)
    print(f"autocast_bf16: {os.path.getsize(path) / 1024:.1f} KB, {len(arrs)} arrays")


if __name__ == "__main__":
    main()
