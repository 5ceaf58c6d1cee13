"""TEST INFRASTRUCTURE — pins the reference's initialisation constants (build container only).

    python -m oracle.make_golden_init        # writes tests/golden/bias_init.npz

`v10Detect3d.bias_init` (nn/modules/head.py:847-871), `Detect.bias_init` / `v10Detect.bias_init` (head.py:95-109, 535-543) and
`initialize_weights` (utils/torch_utils.py:327-337) decide the parity constants of a freshly built model: the head biases
(deterministic), the ranges / spread of the re-drawn size / depth projection weights, BatchNorm eps / momentum.  The fixture holds
the reference's values for the 3-level and the 2-level 3D head and for the 2D head; tests/test_host_logic.py holds the build's
`bias_init` / `initialize_weights` to them.  Data only.
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
from oracle.make_golden import OUT  # noqa: E402


def main():
    R.import_reference()
    from ultralytics.nn.modules.head import v10Detect, v10Detect3d
    from ultralytics.utils.torch_utils import initialize_weights

    arrs = {}
    ch = (16, 32, 64)
    chan = {k + "_c": 16 for k in ("cls", "o2d", "s2d", "o3d", "s3d", "hd", "dep", "dep_un")}
    for nl in (3, 2):
        torch.manual_seed(0)
        hd = v10Detect3d(3, ch, False, chan, False, False, False, False, nl, False, False, 3, 3)
        hd.stride = torch.tensor([8.0, 16.0, 32.0][:nl])
        hd.bias_init()
        for name in ("cls", "o2d", "s2d", "o3d", "s3d", "hd", "dep", "dep_un"):
            for i in range(nl):
                proj = getattr(hd, name)[i][-1]
                if name not in ("hd", "dep_un"):  # untouched by bias_init: default nn.Conv2d init (random)
                    arrs[f"head3d_nl{nl}/{name}/{i}/bias"] = proj.bias.detach().numpy()
                if name in ("s3d", "dep"):
                    w = proj.weight.detach()
                    arrs[f"head3d_nl{nl}/{name}/{i}/weight_stats"] = np.array([float(w.min()), float(w.max()), float(w.mean()), float(w.std())])
        # o2o / o2m are rebuilt by bias_init (aliases + deep copy): the one-to-many set must equal the one-to-one set
        same = all(torch.equal(a, b) for a, b in zip(hd.o2o_heads.state_dict().values(), hd.o2m_heads.state_dict().values()))
        arrs[f"head3d_nl{nl}/o2m_equals_o2o"] = np.array(int(same))
    torch.manual_seed(0)
    h2 = v10Detect(80, ch)
    h2.stride = torch.tensor([8.0, 16.0, 32.0])
    h2.bias_init()
    for i in range(3):
        arrs[f"head2d/cv2/{i}/bias"] = h2.cv2[i][-1].bias.detach().numpy()
        arrs[f"head2d/cv3/{i}/bias"] = h2.cv3[i][-1].bias.detach().numpy()
        arrs[f"head2d/one2one_cv2/{i}/bias"] = h2.one2one_cv2[i][-1].bias.detach().numpy()
        arrs[f"head2d/one2one_cv3/{i}/bias"] = h2.one2one_cv3[i][-1].bias.detach().numpy()
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8), torch.nn.SiLU())
    initialize_weights(m)
    arrs["initialize_weights/bn_eps_momentum"] = np.array([m[1].eps, m[1].momentum])
    arrs["initialize_weights/silu_inplace"] = np.array(int(m[2].inplace))
    path = os.path.join(OUT, "bias_init.npz")
    np.savez_compressed(path, **arrs)
    print(f"bias_init: {os.path.getsize(path) / 1024:.1f} KB, {len(arrs)} arrays")
    for k in sorted(arrs):
        if "weight_stats" in k or k.endswith("/0/bias"):
            print(k, np.round(arrs[k], 4)[:6])


if __name__ == "__main__":
    main()
