"""TEST INFRASTRUCTURE — mints the BASELINE.json configs[0] fixture from the REFERENCE (build container only):
YOLOv10-N 2D-only (the shipped cfg/models/v10/yolov10n.yaml, nc=80), 320x320, batch 2, one training step and one eval pass on the
reference's CPU PyTorch path.

    python -m oracle.make_golden_configs        # writes tests/golden/e2e_n2d_320.npz

The fixture holds data only: the seeded initial state, the (8-bit) images, the targets and the reference's outputs.
"""
from __future__ import annotations

import copy
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_shim as R  # noqa: E402
from oracle.make_golden import save, set_bn, synth_batch  # noqa: E402


def main():
    R.import_reference()
    cfg = R.load_yaml("v10/yolov10n.yaml")
    cfg = {k: v for k, v in cfg.items() if k in ("backbone", "head", "nc", "scales")}
    cfg["scale"] = "n"
    m = R.build_model(cfg, seed=0)
    gen = torch.Generator().manual_seed(11)
    set_bn(m, gen)
    with torch.no_grad():
        for n_, p_ in m.model[-1].named_parameters():
            if n_.startswith("one2one"):  # the one-to-one head starts as a deep copy of the one-to-many head: separate them
                p_.add_(0.01 * torch.randn(p_.shape, generator=gen))
    H = 320
    img8 = torch.randint(0, 256, (2, 3, H, H), generator=gen, dtype=torch.uint8)  # stored as bytes; the model sees img8 / 255
    img = img8.float() / 255
    bt = synth_batch(2, H, H, 4, gen, nc=80)
    bt["img"] = img
    state = {k: v.clone() for k, v in m.state_dict().items()}
    m.train()
    loss, items = m(bt)
    loss.backward()
    names = dict(m.named_parameters())
    keys = list(names)
    sel = keys[:6] + [k for k in keys if k.startswith(f"model.{len(m.model) - 1}.")][:8] + keys[len(keys) // 2: len(keys) // 2 + 4]
    gsel = {k: names[k].grad.clone() for k in sel if names[k].grad is not None}
    gnorm = {k: names[k].grad.norm().reshape(1) for k in keys if names[k].grad is not None}  # every parameter: gradient L2 norm
    after = {k: v.clone() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k}
    me = copy.deepcopy(m).eval()
    with torch.no_grad():
        oe = me(img)
    from ultralytics.utils import ops as ref_ops
    bx, sc, lab = ref_ops.v10postprocess(oe["one2one"][0].permute(0, 2, 1), 300, 80)
    save("e2e_n2d_320", img8=img8, batch={k: v for k, v in bt.items() if k in ("batch_idx", "cls", "bboxes")}, loss=loss, items=items,
         state=state, grads=gsel, grad_norms=gnorm, state_after=after, y_eval_o2o=oe["one2one"][0], y_eval_o2m=oe["one2many"][0],
         post_boxes=bx, post_scores=sc, post_labels=lab, strides=m.stride)


if __name__ == "__main__":
    main()
