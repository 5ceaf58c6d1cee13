"""Test-only torch formulations of the two task-aligned assigners (utils/tal.py:19-264, 355-753), CIoU (utils/metrics.py:78-134),
the 3D keypoints (utils/keypoint_utils.py:11-118) and the target padding (utils/loss.py:795-810) on the tensors' own device.

They are comparators for the HIP kernels of tal_loss2d.hip / tal_loss3d.hip at sizes where the CPU oracle is slow (A = 8400) and are
themselves pinned to the reference's fixtures by tests/test_host_logic.py.  Nothing in the product package imports this module.
Tie rule of the top-k: lowest index first."""
import math

import torch
import torch.nn.functional as F


def ciou(b1, b2, eps=1e-7):
    """utils/metrics.py:78-134 (xywh=False, CIoU=True) on broadcastable (...,4) boxes"""
    x11, y11, x12, y12 = b1.unbind(-1)
    x21, y21, x22, y22 = b2.unbind(-1)
    w1, h1 = x12 - x11, y12 - y11 + eps
    w2, h2 = x22 - x21, y22 - y21 + eps
    inter = (torch.minimum(x12, x22) - torch.maximum(x11, x21)).clamp(min=0) * \
            (torch.minimum(y12, y22) - torch.maximum(y11, y21)).clamp(min=0)
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    cw = torch.maximum(x12, x22) - torch.minimum(x11, x21)
    chh = torch.maximum(y12, y22) - torch.minimum(y11, y21)
    c2 = cw.pow(2) + chh.pow(2) + eps
    rho2 = ((x21 + x22 - x11 - x12).pow(2) + (y21 + y22 - y11 - y12).pow(2)) / 4
    v = (4 / math.pi ** 2) * ((w2 / h2).atan() - (w1 / h1).atan()).pow(2)
    with torch.no_grad():
        alpha = v / (v - iou + (1 + eps))
    return iou - (rho2 / c2 + v * alpha)


def keypoints_3d(center, dep, size3d, hbin, hres, calib):
    """utils/keypoint_utils.py:11-118: 8 box corners in the camera frame, (B,N,8,3)"""
    cu, cv, fu, fv, tx, ty = [calib[:, None, k:k + 1] for k in range(6)]
    X = (center[..., 0:1] - cu) * dep / fu + tx
    Y = (center[..., 1:2] - cv) * dep / fv + ty
    loc = torch.cat((X, Y, dep), -1)
    hl, hw, hh = size3d[..., 2:3] / 2, size3d[..., 1:2] / 2, size3d[..., 0:1] / 2
    cx = torch.cat((hl, hl, -hl, -hl, hl, hl, -hl, -hl), -1)
    cy = torch.cat((hw, -hw, hw, -hw, hw, -hw, hw, -hw), -1)
    cz = torch.cat((-hh, -hh, -hh, -hh, hh, hh, hh, hh), -1)
    corners = torch.stack((cx, cy, cz), -1)
    bi = hbin.argmax(-1) if hbin.shape[-1] > 1 else hbin[..., 0].long()
    res = hres.gather(-1, bi.unsqueeze(-1))[..., 0] if hres.shape[-1] > 1 else hres[..., 0]
    ang = bi.to(res.dtype) * (2 * math.pi / 12.0) + res
    ang = torch.where(ang > math.pi, ang - 2 * math.pi, ang)
    ry = ang.unsqueeze(-1) + torch.arctan2(center[..., 0:1] - cu, fu)
    ry = torch.where(ry > math.pi, ry - 2 * math.pi, ry)
    ry = torch.where(ry < -math.pi, ry + 2 * math.pi, ry)
    # R = Rx(pi/2) @ Ry(-ry) applied as out_i = sum_j R[j,i] p_j (keypoint_utils.py:87-110), written out explicitly:
    # a (B,N,3,3) batched matmul costs a library GEMM launch per call for what is 9 multiply-adds per box
    a = -ry
    ca, sa = torch.cos(a), torch.sin(a)
    cx_, sx_ = math.cos(math.pi / 2), math.sin(math.pi / 2)
    px, py, pz = corners[..., 0], corners[..., 1], corners[..., 2]
    ox = ca * px + (sx_ * sa) * py - (cx_ * sa) * pz
    oy = cx_ * py + sx_ * pz
    oz = sa * px - (sx_ * ca) * py + (cx_ * ca) * pz
    return torch.stack((ox, oy, oz), -1) + loc.unsqueeze(-2)


def _topk_mask(metric, k, valid_gt):
    """utils/tal.py:615-649 with lowest-index-first ties"""
    B, n, A = metric.shape
    order = torch.sort(metric, dim=-1, descending=True, stable=True)[1][..., :k]
    order = torch.where(valid_gt.expand(-1, -1, k).bool(), order, torch.zeros_like(order))
    cnt = torch.zeros(B, n, A, dtype=torch.int32, device=metric.device)
    cnt.scatter_add_(-1, order, torch.ones_like(order, dtype=torch.int32))
    return torch.where(cnt > 1, torch.zeros_like(cnt), cnt).to(metric.dtype)


def _resolve(mask_pos, overlaps):
    """utils/tal.py:728-753"""
    n = mask_pos.shape[1]
    fg = mask_pos.sum(-2)
    multi = (fg.unsqueeze(1) > 1).expand(-1, n, -1)
    onehot = torch.zeros_like(mask_pos)
    onehot.scatter_(1, overlaps.argmax(1).unsqueeze(1), 1)
    mask_pos = torch.where(multi, onehot, mask_pos)
    return mask_pos.argmax(-2), mask_pos.sum(-2), mask_pos


def _in_gts(anc, gt_bboxes, eps=1e-9):
    lt, rb = gt_bboxes[..., None, :2], gt_bboxes[..., None, 2:]
    return (torch.cat((anc[None, None] - lt, rb - anc[None, None]), -1).amin(-1) > eps).to(gt_bboxes.dtype)


class TaskAlignedAssigner:
    """utils/tal.py:19-264"""

    def __init__(self, topk=13, num_classes=80, alpha=1.0, beta=6.0, eps=1e-9):
        self.topk, self.num_classes, self.alpha, self.beta, self.eps = topk, num_classes, alpha, beta, eps

    @torch.no_grad()
    def __call__(self, pd_scores, pd_bboxes, anc, gt_labels, gt_bboxes, mask_gt):
        B, A = pd_scores.shape[:2]
        n, nc = gt_bboxes.shape[1], self.num_classes
        dev = pd_scores.device
        if n == 0:
            return (torch.full((B, A), float(nc), device=dev), torch.zeros_like(pd_bboxes), torch.zeros_like(pd_scores),
                    torch.zeros(B, A, dtype=torch.bool, device=dev), torch.zeros(B, A, dtype=torch.long, device=dev))
        in_g = _in_gts(anc, gt_bboxes)
        m = (in_g * mask_gt).bool()
        lab = gt_labels.squeeze(-1).long()
        sc = pd_scores.gather(2, lab.clamp(min=0)[:, None, :].expand(-1, A, -1)).permute(0, 2, 1)
        sc = torch.where(m, sc, torch.zeros_like(sc))
        ov = torch.where(m, ciou(gt_bboxes[:, :, None, :], pd_bboxes[:, None, :, :]).clamp(min=0), torch.zeros_like(sc))
        align = sc.pow(self.alpha) * ov.pow(self.beta)
        mask_pos = _topk_mask(align, self.topk, mask_gt) * in_g * mask_gt
        gt_idx, fg, mask_pos = _resolve(mask_pos, ov)
        flat = gt_idx + torch.arange(B, device=dev)[:, None] * n
        t_lab = lab.flatten()[flat].clamp(min=0)
        t_box = gt_bboxes.reshape(-1, 4)[flat]
        t_sc = F.one_hot(t_lab, nc).to(pd_scores.dtype) * (fg > 0).unsqueeze(-1)
        align = align * mask_pos
        pa = align.amax(-1, keepdim=True)
        po = (ov * mask_pos).amax(-1, keepdim=True)
        norm = (align * po / (pa + self.eps)).amax(-2).unsqueeze(-1)
        return t_lab, t_box, t_sc * norm, fg.bool(), gt_idx


class TaskAlignedAssigner3d:
    """utils/tal.py:355-753 (use_2d and use_3d, 'l1' keypoint metric, constrain_anchors: cfg/default.yaml:112-119)"""

    def __init__(self, topk=8, num_classes=3, alpha=0.5, beta=3.0, gamma=3.0, eps=1e-9, use_2d=True, use_3d=True,
                 kps_dist_metric="l1", constrain_anchors=True):
        if not (use_2d and use_3d and kps_dist_metric == "l1" and constrain_anchors):
            raise NotImplementedError("only the default 2D+3D / l1 / constrained assignment is built")
        self.topk, self.num_classes, self.alpha, self.beta, self.gamma, self.eps = topk, num_classes, alpha, beta, gamma, eps

    @torch.no_grad()
    def __call__(self, pd_scores, pd_bboxes, pd_3d, anc, gts, mask_gt, stride_tensor, calibs, mean_sizes):
        gl, gb, gc2, gs2, gc3, gs3, gd, ghb, ghr = gts
        B, A = pd_scores.shape[:2]
        n, nc = gb.shape[1], self.num_classes
        dev = pd_scores.device
        o3d, s3d, hd, dep, _ = pd_3d.split((2, 3, 24, 1, 1), -1)
        pc3 = anc + o3d * stride_tensor
        ps3 = mean_sizes[pd_scores.argmax(-1)] + s3d
        lab = gl.squeeze(-1).long()
        g_kps = keypoints_3d(gc3, gd, mean_sizes[lab.clamp(min=0)] + gs3, ghb, ghr, calibs)
        p_kps = keypoints_3d(pc3, dep, ps3, hd[..., :12], hd[..., 12:], calibs)
        in_g = _in_gts(anc, gb)
        m = (in_g * mask_gt).bool()
        sc = pd_scores.gather(2, lab.clamp(min=0)[:, None, :].expand(-1, A, -1)).permute(0, 2, 1)
        sc = torch.where(m, sc, torch.zeros_like(sc))
        dist = torch.zeros(B, n, A, device=dev)
        for g in range(n):  # keeps the (B,n,A,8,3) temporary of the reference (tal.py:593-595) out of memory
            dist[:, g] = (p_kps - g_kps[:, g:g + 1]).abs().sum((-1, -2)) / 24
        sim = torch.where(m, 1 / torch.exp(dist), torch.zeros_like(dist))
        ov = torch.where(m, ciou(gb[:, :, None, :], pd_bboxes[:, None, :, :]).clamp(min=0), torch.zeros_like(dist))
        align = sc.pow(self.alpha) * ov.pow(self.beta) * sim.pow(self.gamma)
        mask_pos = _topk_mask(align, self.topk, mask_gt) * in_g * mask_gt
        gt_idx, fg, mask_pos = _resolve(mask_pos, sim)
        flat = gt_idx + torch.arange(B, device=dev)[:, None] * n
        t_lab = lab.flatten()[flat].clamp(min=0)

        def take(t):
            return t.reshape(-1, t.shape[-1])[flat]

        t_sc = F.one_hot(t_lab, nc).to(pd_scores.dtype) * (fg > 0).unsqueeze(-1)
        align = align * mask_pos
        pa = align.amax(-1, keepdim=True)
        po = (sim * mask_pos).amax(-1, keepdim=True)
        norm = (align * po / (pa + self.eps)).amax(-2).unsqueeze(-1)
        targets = [t_lab, t_sc * norm, take(gc2), take(gs2), take(gc3), take(gs3), take(gd), take(ghb), take(ghr)]
        return targets, fg.bool(), gt_idx, p_kps, g_kps


def pad_targets_torch(rows, B, width, scale):
    """utils/loss.py:795-810"""
    dev = rows.device
    if rows.shape[0] == 0:
        return torch.zeros(B, 0, width, device=dev)
    bi = rows[:, 0].long()
    counts = torch.bincount(bi, minlength=B)
    nmax = int(counts.max())
    order = torch.argsort(bi, stable=True)
    start = torch.cumsum(counts, 0) - counts
    pos = torch.arange(rows.shape[0], device=dev) - start[bi[order]]
    out = torch.zeros(B, nmax, width, device=dev)
    out[bi[order], pos] = rows[order, 1:]
    xywh = out[..., 1:5] * scale
    xy, wh = xywh[..., :2], xywh[..., 2:]
    out[..., 1:5] = torch.cat((xy - wh / 2, xy + wh / 2), -1)
    return out
