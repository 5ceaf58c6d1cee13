"""TEST INFRASTRUCTURE — pins the NON-DEFAULT modes of the reference's TaskAlignedAssigner3d (build container only).

    python -m oracle.make_golden_modes        # writes tests/golden/tal3d_modes.npz

`cfg/default.yaml:116-119` exposes `tal_2d`, `tal_3d`, `kps_dist_metric` and `constrain_anchors`; `utils/loss.py:787-791` hands them to
`TaskAlignedAssigner3d` (`utils/tal.py:465-497`: box-only / keypoint-only / combined metric, l1 / l2 keypoint distance, candidates
restricted to anchors inside the box or not).  The shipped yamls use the defaults (the `tal3d_topk*` fixtures); this fixture runs the
reference's assigner on the INPUTS of `tal3d_topk8.npz` in the other modes and stores its outputs.  Data only.
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

MODES = {  # name -> (use_2d, use_3d, kps_dist_metric, constrain_anchors, topk)
    "box_only": (True, False, "l1", True, 8),
    "kps_only_l2": (False, True, "l2", True, 8),
    "both_l2": (True, True, "l2", True, 8),
    "both_unconstrained": (True, True, "l1", False, 8),
    "kps_only_unconstrained_top1": (False, True, "l1", False, 1),
}


def main():
    R.import_reference()
    from ultralytics.utils.tal import TaskAlignedAssigner3d
    z = np.load(os.path.join(OUT, "tal3d_topk8.npz"))
    t = lambda k: torch.from_numpy(z[k])
    gt = t("gt")
    gts = gt.split((1, 4, 2, 2, 2, 3, 1, 1, 1), 2)
    anc, st = t("anc"), t("stride")
    arrs = {}
    for name, (u2, u3, metric, con, topk) in MODES.items():
        asg = TaskAlignedAssigner3d(topk=topk, num_classes=3, alpha=0.5, beta=1.0, gamma=1.0, use_2d=u2, use_3d=u3, kps_dist_metric=metric,
                                    constrain_anchors=con)
        targets, fg, gi, _, _ = asg(t("pd_scores"), t("pd_bboxes"), t("pd_3d"), anc * st, gts, t("mask_gt"), st, t("calib"), t("mean_sizes"))
        arrs[f"{name}/fg_mask"] = fg.numpy()
        arrs[f"{name}/target_gt_idx"] = gi.numpy()
        arrs[f"{name}/target_labels"] = targets[0].numpy()
        arrs[f"{name}/target_scores"] = targets[1].numpy()
        arrs[f"{name}/mode"] = np.array([int(u2), int(u3), int(metric == "l2"), int(con), topk])
        print(name, "positives:", int(fg.sum()))
    np.savez_compressed(os.path.join(OUT, "tal3d_modes.npz"), **arrs)


if __name__ == "__main__":
    main()
