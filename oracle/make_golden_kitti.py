"""TEST INFRASTRUCTURE — mints the KITTI decode fixture from the REFERENCE (build container only):
`KITTIDataset.decode_preds` (data/datasets/kitti.py:519-576) on seeded post-processed predictions, with the reference's own
`Calibration` objects (data/datasets/kitti_utils.py:178-196) built from float32 P2 matrices.

    python -m oracle.make_golden_kitti        # writes tests/golden/kitti_decode.npz

The fixture holds data only: predictions, calibration constants, ratio / inverse affine, and the reference's rows padded to
(B, K, 14) with a keep mask.  (numpy here is 2.x: python-float x float32-scalar arithmetic stays float32 under NEP 50, the
reference's pinned numpy 1.x promotes it to float64 -- the fixture is therefore float32-accurate in `ry`, 1e-7 relative.)
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_shim as R  # noqa: E402
from oracle.make_golden import save  # noqa: E402


def synth(gen, B, K):
    u = lambda *s: torch.rand(*s, generator=gen)
    x1, y1 = u(B, K) * 1000, u(B, K) * 250
    w, h = 20 + u(B, K) * 250, 20 + u(B, K) * 120
    c3 = torch.stack((x1 + w / 2 + torch.randn(B, K, generator=gen) * 3, y1 + h / 2 + torch.randn(B, K, generator=gen) * 3), -1)
    preds = torch.cat((torch.stack((x1, y1, x1 + w, y1 + h), -1), c3, torch.randn(B, K, 3, generator=gen) * 0.2,
                       torch.randn(B, K, 24, generator=gen), 3 + u(B, K, 1) * 60, torch.randn(B, K, 1, generator=gen),
                       torch.randn(B, K, 1, generator=gen) * 4, torch.randint(0, 3, (B, K, 1), generator=gen).float()), -1)
    preds[:, 0, 34] = 9.0    # exp(-9) * sigmoid(.) < 0.001: dropped by the threshold
    preds[:, 1, 35] = -12.0
    return preds.float()


def main():
    R.import_reference()
    from ultralytics.data.datasets.kitti import KITTIDataset
    from ultralytics.data.datasets.kitti_utils import Calibration
    gen = torch.Generator().manual_seed(5)
    B, K = 3, 12
    preds = synth(gen, B, K)
    P2s, calibs = [], []
    for i in range(B):
        f = 707.0493 + 10 * i
        P2 = np.array([[f, 0, 604.0814 + 3 * i, 45.75831 - i], [0, f, 180.5066 - 2 * i, -0.3454157 + 0.1 * i], [0, 0, 1, 0.004981016]], dtype=np.float32)
        P2s.append(P2)
        calibs.append(Calibration({"P2": P2, "R0": np.eye(3, dtype=np.float32), "Tr_velo2cam": np.eye(3, 4, dtype=np.float32)}))
    ratio_pad = torch.tensor(np.array([[[1280 / (1242.0 - 2 * i), 384 / (375.0 - i)], [0, 0]] for i in range(B)]))   # kitti.py:404, 431
    inv_trans = [np.array([[0.97 + 0.01 * i, 0.0, 1.5 * i], [0.0, 0.976 - 0.01 * i, -0.75 * i]], dtype=np.float64) for i in range(B)]
    files = [f"{i:06d}.png" for i in range(B)]
    out = {}
    for tag, undo, cam in (("aug", True, False), ("noaug", False, False), ("camdis", True, True)):
        ds = types.SimpleNamespace(cls_mean_size=np.array(  # kitti.py:38-41
            [[1.52563191462, 1.62856739989, 3.88311640418], [1.76255119, 0.66068622, 0.84422524], [1.73698127, 0.59706367, 1.76282397]]),
            use_camera_dis=cam)
        res = KITTIDataset.decode_preds(ds, preds.clone(), calibs, files, ratio_pad, inv_trans, undo_augment=undo, threshold=0.001)
        rows = np.zeros((B, K, 14), dtype=np.float64)
        counts = np.zeros((B,), dtype=np.int64)
        for i, f_ in enumerate(files):
            t = res[f_]
            counts[i] = len(t)
            for j, r in enumerate(t):
                rows[i, j] = np.asarray([float(np.asarray(v).reshape(-1)[0]) for v in r], dtype=np.float64)
        out[f"rows_{tag}"] = torch.from_numpy(rows)
        out[f"count_{tag}"] = torch.from_numpy(counts)
    calib6 = torch.tensor([[float(c.cu), float(c.cv), float(c.fu), float(c.fv), float(c.tx), float(c.ty)] for c in calibs], dtype=torch.float64)
    save("kitti_decode", preds=preds, calib=calib6, ratio=ratio_pad[:, 0].clone(), inv_trans=torch.from_numpy(np.stack(inv_trans)), **out)


if __name__ == "__main__":
    main()
