"""TEST INFRASTRUCTURE — pins the NON-DEFAULT constructor options of the reference's v10Detect3d (build container only).

    python -m oracle.make_golden_headopts     # writes tests/golden/head3d_opt_*.npz

`nn/modules/head.py:554-634` builds the 3D head from yaml switches that every shipped yaml leaves off: `dsconv` (each k x k Conv becomes a
depth-wise Conv + a 1x1 Conv, :645-650), `half_channels` (second conv mid -> mid/2, :629-636), `common_head` (one shared 3x3 Conv per
level, then branches of ONE conv + projection, :600-609, 638-643, 724-725) and `use_predecessors` (a branch's input is its level's
feature map concatenated with the DETACHED outputs of earlier branches, depth divided by 65, :583-606, 727-737).  This script runs the
reference's head with each switch on seeded inputs and explicit weights: training forward + backward, and the eval forward where the
reference's own eval path runs at all (with `use_predecessors` it raises: `inference_forward_feat`, :694-716, feeds the branches bare
feature patches).  `common_head` cannot be minted: the reference's own training forward fails on it (`single_head_forward`, :745-749,
asserts three layers; the small heads have two) - the script prints that and moves on.  The eval fixtures hold whatever the reference
computes, including what its patch path does with `dsconv` (the nested Sequentials keep their padding, :706-708 only looks at
top-level Conv layers, and the output is read at patch cell (0, 0)).  Data only.
"""
from __future__ import annotations

import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_shim as R  # noqa: E402
from oracle.make_golden import save, set_bn  # noqa: E402

#            name        dsconv half   common pred
VARIANTS = (("dsconv", True, False, False, False),
            ("half", False, True, False, False),
            ("ds_half", True, True, False, False),
            ("pred", False, False, False, True),
            ("pred_half", False, True, False, True),
            ("common", False, False, True, False))
KEEP = ("0", "4", "6", "7")  # branches whose parameter gradients are stored: cls, s3d (sees cls), dep (cls, s3d), dep_un (cls, s3d, dep)


def main():
    R.import_reference()
    from ultralytics.nn.modules.head import v10Detect3d
    ch, nl = (8, 16, 32), 2
    chan = {k + "_c": 16 for k in ("cls", "o2d", "s2d", "o3d", "s3d", "hd", "dep", "dep_un")}
    for name, ds, half, common, pred in VARIANTS:
        g = torch.Generator().manual_seed(20 + len(name))
        torch.manual_seed(31)
        hd = v10Detect3d(3, ch, ds, chan, pred, True, False, common, nl, half, False, 3, 3)
        hd.stride = torch.tensor([8.0, 16.0])
        hd.bias_init()
        set_bn(hd, g)
        with torch.no_grad():
            for p_ in hd.o2m_heads.parameters():
                p_.add_(0.01 * torch.randn(p_.shape, generator=g))
        pick = lambda k: k.startswith(("o2o_heads.", "o2m_heads.", "common."))
        sd = {f"model.0.{k}": v.clone() for k, v in hd.state_dict().items() if pick(k)}
        xs = [torch.randn(2, ch[i], s, s, generator=g) for i, s in enumerate((16, 8))]
        hd.train()
        xin = [x.clone().requires_grad_(True) for x in xs]
        try:
            out = hd(list(xin))
        except AssertionError:  # common_head: single_head_forward (head.py:745-749) asserts three layers, the small heads have two
            import traceback
            print(f"{name}: the reference's TRAINING forward raises AssertionError at {traceback.extract_tb(sys.exc_info()[2])[-1].name}")
            continue
        maps = out["one2many"] + out["one2one"]
        rs = [torch.rand(t.shape, generator=g) - 0.5 for t in maps]
        sum((t * r).sum() for t, r in zip(maps, rs)).backward()
        grads = {f"model.0.{k}": p.grad.clone() for k, p in hd.named_parameters()
                 if p.grad is not None and ((k.startswith(("o2o_heads.", "o2m_heads.")) and k.split(".")[1] in KEEP) or k.startswith("common."))}
        after = {f"model.0.{k}": v.clone() for k, v in hd.state_dict().items() if pick(k) and ("running" in k or "num_batches" in k)
                 and (k.startswith("common.") or k.split(".")[1] == "6")}
        save(f"head3d_opt_{name}_train", x=xs, r=rs, o2m=out["one2many"], o2o=out["one2one"], o2m_embs=out["o2m_embs"], o2o_embs=out["o2o_embs"],
             dx=[x.grad for x in xin], state=sd, grads=grads, state_after=after, meta=np.array([int(ds), int(half), int(common), int(pred)]))
        he = copy.deepcopy(hd).eval()
        xe = [torch.randn(2, ch[i], s, s, generator=g) for i, s in enumerate((32, 16))]
        try:
            with torch.no_grad():
                y, emaps = he(list(xe))["one2one"]
        except RuntimeError as e:
            print(f"{name}: the reference's eval forward raises {type(e).__name__}: {str(e).splitlines()[0][:120]}")
            continue
        sde = {f"model.0.{k}": v.clone() for k, v in he.state_dict().items() if k.startswith("o2o_heads.")}
        save(f"head3d_opt_{name}_eval", x=xe, y=y, maps=emaps, state=sde, meta=np.array([int(ds), int(half), int(common), int(pred)]))


if __name__ == "__main__":
    main()
