"""Losses + task-aligned assignment for the YOLOv10 (2D) and YOLOv10-3D heads — device side.

Mirrors the reference interfaces (utils/loss.py: v8DetectionLoss :157, v10DetectLoss :727, DetectLoss3d :740,
DDDetectionLoss :774; utils/tal.py: TaskAlignedAssigner :19, TaskAlignedAssigner3d :355;
utils/keypoint_utils.py; utils/metrics.py:78 bbox_iou).  All arithmetic is fp32 (the reference runs these
under autocast's fp32 policy), on the tensors' own device.

The 3D training loss (DetectLoss3d / DDDetectionLoss: assignment + six loss terms + gradient wrt the head maps) runs on the
fused HIP kernels of csrc/tal_loss3d.hip (`Loss3dFn`).  The torch-op classes below (`TaskAlignedAssigner*`, `v8DetectionLoss`)
express the same algorithms with torch *device* ops; they serve the 2D (config C1) loss, which has no HIP kernel yet, and the
GPU tests that hold the HIP assigner to them.  Tie rule of the top-k is pinned to lowest-index-first.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

import ctypes

from . import ops
from ._lib import Y3DError, lib
from .modules import make_anchors


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


def _pad_targets(rows, B, width, scale):
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


def _flatten_maps(feats):
    """list of (B, no, H, W) NHWC maps -> (B, A, no) fp32 (a free view per level + one cat)"""
    B, no = feats[0].shape[:2]
    return torch.cat([f.permute(0, 2, 3, 1).reshape(B, -1, no) for f in feats], 1).float()


class Loss3dFn(torch.autograd.Function):
    """One head set: task-aligned assignment (no grad) + the six 3D loss terms + d(sum of terms)/d(head maps), on the fused HIP
    kernels.  apply(cfg, gt(B,n,17), calib, mean_sizes, *maps) -> (sum_of_items, items[6], fg_mask, target_gt_idx, target_scores)."""

    @staticmethod
    def forward(ctx, cfg, gt, calib, mean_sizes, *maps):
        L = lib()
        strides, nc, topk, alpha, beta, gamma, w = cfg
        dtype = maps[0].dtype
        dt = ops.code(dtype)
        st = ops.stream()
        dev = maps[0].device
        nl = len(maps)
        for m in maps:
            if not (m.is_cuda and ops.px_dense(m)):
                raise Y3DError("Loss3dFn: head maps must be pixel-dense NHWC tensors on a HIP device")
        B, no = maps[0].shape[:2]
        Hs = [m.shape[2] for m in maps]
        Ws = [m.shape[3] for m in maps]
        A = sum(h * w_ for h, w_ in zip(Hs, Ws))
        n = gt.shape[1]
        gt = gt.float().contiguous()
        calib = calib.float().contiguous()
        mean_sizes = mean_sizes.float().contiguous()
        grads = [ops.nhwc_empty(B, no, h, w_, dtype, dev) for h, w_ in zip(Hs, Ws)]
        PV = ctypes.c_void_p * nl
        c_maps = PV(*[m.data_ptr() for m in maps])
        c_grads = PV(*[g.data_ptr() for g in grads])
        c_psw = (ctypes.c_int64 * nl)(*[m.stride(3) for m in maps])
        c_gsw = (ctypes.c_int64 * nl)(*[no] * nl)
        c_H, c_W = (ctypes.c_int * nl)(*Hs), (ctypes.c_int * nl)(*Ws)
        c_st = (ctypes.c_float * nl)(*strides)
        nsc = L.tal3d_scratch_floats(B, n, A, topk)
        if nsc < 0:
            raise Y3DError("Loss3dFn: assignment scratch exceeds 2^31 floats")
        scratch = torch.empty(nsc, dtype=torch.float32, device=dev)
        fg = torch.empty(B, A, dtype=torch.uint8, device=dev)
        gi = torch.empty(B, A, dtype=torch.int32, device=dev)
        ts = torch.empty(B, A, nc, dtype=torch.float32, device=dev)
        scal = torch.empty(2, dtype=torch.float32, device=dev)
        L.tal3d_assign(dt, nl, c_maps, c_psw, c_H, c_W, c_st, B, nc, gt.data_ptr(), n, calib.data_ptr(), mean_sizes.data_ptr(), topk,
                       alpha, beta, gamma, scratch.data_ptr(), fg.data_ptr(), gi.data_ptr(), ts.data_ptr(), scal.data_ptr(), st)
        nblk = (B * A + 255) // 256
        part = torch.empty(nblk * 6, dtype=torch.float32, device=dev)
        items = torch.empty(6, dtype=torch.float32, device=dev)
        L.loss3d(dt, nl, c_maps, c_psw, c_grads, c_gsw, c_H, c_W, c_st, B, nc, gt.data_ptr(), n, fg.data_ptr(), gi.data_ptr(), ts.data_ptr(),
                 scal.data_ptr(), w[0], w[1], w[2], w[3], w[4], w[5], 1.0, part.data_ptr(), items.data_ptr(), st)
        ctx.save_for_backward(*grads)
        total = items.sum()
        ctx.mark_non_differentiable(items, fg, gi, ts)
        return total, items, fg, gi, ts

    @staticmethod
    def backward(ctx, d_total, *unused):
        grads = ctx.saved_tensors
        return (None, None, None, None, *[g * d_total.to(g.dtype) for g in grads])


class DDDetectionLoss:
    """utils/loss.py:774-963"""

    def __init__(self, model, tal_topk=10):
        h = model.args
        m = model.model[-1]
        self.hyp = h
        self.stride = [float(s) for s in m.stride]
        self.nc, self.no = m.nc, m.no
        self.assigner = TaskAlignedAssigner3d(topk=tal_topk, num_classes=self.nc, alpha=h.tal_alpha, beta=h.tal_beta, gamma=h.tal_gamma,
                                              use_2d=h.tal_2d, use_3d=h.tal_3d, kps_dist_metric=h.kps_dist_metric,
                                              constrain_anchors=h.constrain_anchors)
        if getattr(h, "distillation", False):
            raise NotImplementedError("distillation needs the DINOv2 teacher (network); pinned off (SURVEY §0.5)")

    def __call__(self, preds, batch, embeddings=None):
        feats = preds[1] if isinstance(preds, tuple) else preds
        dev = feats[0].device
        if not feats[0].is_cuda:
            raise Y3DError("the 3D loss runs on the HIP kernels of tal_loss3d.hip: head maps must live on a HIP device (no CPU fallback)")
        B = feats[0].shape[0]
        H, W = feats[0].shape[2:]
        imgsz = torch.tensor([H, W], dtype=torch.float32, device=dev) * self.stride[0]
        # the padded targets (one host sync for the per-image maximum) are shared by the one-to-many and one-to-one losses of a step
        keys = ("batch_idx", "cls", "bboxes", "center_2d", "size_2d", "center_3d", "size_3d", "depth", "heading_bin", "heading_res")
        memo_key = (B, H, W, self.stride[0], tuple((batch[k].data_ptr(), batch[k]._version) for k in keys))
        memo = batch.get("_y3d_gt3d")
        if memo is not None and memo[0] == memo_key:
            g = memo[1]
        else:
            rows = torch.cat([batch[k].to(dev).float().view(batch[k].shape[0], -1) for k in keys], 1)
            g = _pad_targets(rows, B, 17, imgsz[[1, 0, 1, 0]])
            batch["_y3d_gt3d"] = (memo_key, g)
        if g.shape[1] == 0:
            loss = torch.zeros(6, device=dev)
            return loss.sum() * B, loss  # reference: graph-less zeros (loss.py:873-877); callers skip the step
        h = self.hyp
        cfg = (self.stride[: len(feats)], self.nc, self.assigner.topk, float(h.tal_alpha), float(h.tal_beta), float(h.tal_gamma),
               (float(h.loss2d), float(h.cls), float(h.depth), float(h.offset3d), float(h.size3d), float(h.heading)))
        maps = [f if f.dtype == ops.compute_dtype() else f.to(ops.compute_dtype()) for f in feats]
        total, items, fg, gt_idx, t_sc = Loss3dFn.apply(cfg, g, batch["calib"].to(dev), batch["mean_sizes"].to(dev), *maps)
        self.last_assignment = (fg.bool(), gt_idx.long(), t_sc)
        return total * B, items


class DetectLoss3d:
    """utils/loss.py:740-771"""

    def __init__(self, model):
        self.one2many = DDDetectionLoss(model, tal_topk=model.args.tal_topk)
        self.one2one = DDDetectionLoss(model, tal_topk=1)

    def __call__(self, preds, batch):
        l1, i1 = self.one2one(preds["one2one"], batch, embeddings=preds.get("o2o_embs"))
        if preds.get("one2many", None):
            lm, im = self.one2many(preds["one2many"], batch, embeddings=preds.get("o2m_embs"))
            return lm + l1, torch.cat((im, i1))
        return torch.zeros(1), i1


class Loss2dFn(torch.autograd.Function):
    """2D head set: assignment + (box, cls, dfl) + gradient wrt the head maps on the fused HIP kernels of tal_loss2d.hip.
    apply(cfg, gt(B,n,5), *maps) -> (sum_of_items, items[3], fg_mask, target_gt_idx, target_scores)"""

    @staticmethod
    def forward(ctx, cfg, gt, *maps):
        L = lib()
        strides, nc, topk, alpha, beta, w = cfg
        dtype = maps[0].dtype
        dt = ops.code(dtype)
        st = ops.stream()
        dev = maps[0].device
        nl = len(maps)
        for m in maps:
            if not (m.is_cuda and ops.px_dense(m)):
                raise Y3DError("Loss2dFn: head maps must be pixel-dense NHWC tensors on a HIP device")
        B, no = maps[0].shape[:2]
        Hs = [m.shape[2] for m in maps]
        Ws = [m.shape[3] for m in maps]
        A = sum(h * w_ for h, w_ in zip(Hs, Ws))
        n = gt.shape[1]
        gt = gt.float().contiguous()
        grads = [ops.nhwc_empty(B, no, h, w_, dtype, dev) for h, w_ in zip(Hs, Ws)]
        PV = ctypes.c_void_p * nl
        c_maps, c_grads = PV(*[m.data_ptr() for m in maps]), PV(*[g.data_ptr() for g in grads])
        c_psw = (ctypes.c_int64 * nl)(*[m.stride(3) for m in maps])
        c_gsw = (ctypes.c_int64 * nl)(*[no] * nl)
        c_H, c_W = (ctypes.c_int * nl)(*Hs), (ctypes.c_int * nl)(*Ws)
        c_st = (ctypes.c_float * nl)(*strides)
        nsc = L.tal3d_scratch_floats(B, n, A, topk)
        if nsc < 0:
            raise Y3DError("Loss2dFn: assignment scratch exceeds 2^31 floats")
        scratch = torch.empty(nsc, dtype=torch.float32, device=dev)
        fg = torch.empty(B, A, dtype=torch.uint8, device=dev)
        gi = torch.empty(B, A, dtype=torch.int32, device=dev)
        ts = torch.empty(B, A, nc, dtype=torch.float32, device=dev)
        scal = torch.empty(2, dtype=torch.float32, device=dev)
        L.tal2d_assign(dt, nl, c_maps, c_psw, c_H, c_W, c_st, B, nc, gt.data_ptr(), n, topk, alpha, beta, scratch.data_ptr(), fg.data_ptr(),
                       gi.data_ptr(), ts.data_ptr(), scal.data_ptr(), st)
        nblk = (B * A + 255) // 256
        part = torch.empty(nblk * 3, dtype=torch.float32, device=dev)
        items = torch.empty(3, dtype=torch.float32, device=dev)
        L.loss2d(dt, nl, c_maps, c_psw, c_grads, c_gsw, c_H, c_W, c_st, B, nc, gt.data_ptr(), n, fg.data_ptr(), gi.data_ptr(), ts.data_ptr(),
                 scal.data_ptr(), w[0], w[1], w[2], 1.0, part.data_ptr(), items.data_ptr(), st)
        ctx.save_for_backward(*grads)
        total = items.sum()
        ctx.mark_non_differentiable(items, fg, gi, ts)
        return total, items, fg, gi, ts

    @staticmethod
    def backward(ctx, d_total, *unused):
        return (None, None, *[g * d_total.to(g.dtype) for g in ctx.saved_tensors])


class v8DetectionLoss:
    """utils/loss.py:157-257 (+BboxLoss :73-113) on the HIP kernels of tal_loss2d.hip"""

    def __init__(self, model, tal_topk=10):
        m = model.model[-1]
        self.hyp = model.args
        self.stride = [float(s) for s in m.stride]
        self.nc, self.no, self.reg_max = m.nc, m.no, m.reg_max
        if self.reg_max != 16:
            raise NotImplementedError("the DFL kernels are built for reg_max = 16")
        self.topk = tal_topk
        self.assigner = TaskAlignedAssigner(topk=tal_topk, num_classes=self.nc, alpha=0.5, beta=6.0)  # torch-op formulation (tests)

    def __call__(self, preds, batch):
        feats = preds[1] if isinstance(preds, tuple) else preds
        dev = feats[0].device
        if not feats[0].is_cuda:
            raise Y3DError("the 2D loss runs on the HIP kernels of tal_loss2d.hip: head maps must live on a HIP device (no CPU fallback)")
        B = feats[0].shape[0]
        H, W = feats[0].shape[2:]
        imgsz = torch.tensor([H, W], dtype=torch.float32, device=dev) * self.stride[0]
        rows = torch.cat((batch["batch_idx"].view(-1, 1), batch["cls"].view(-1, 1), batch["bboxes"]), 1).to(dev).float()
        g = _pad_targets(rows, B, 5, imgsz[[1, 0, 1, 0]])
        if g.shape[1] == 0:
            # no boxes at all: background-only classification loss (loss.py:244), dense BCE against zero targets
            sc = _flatten_maps(feats)[..., self.reg_max * 4:]
            l_cls = F.binary_cross_entropy_with_logits(sc, torch.zeros_like(sc), reduction="none").sum() * self.hyp.cls
            loss = torch.stack((torch.zeros((), device=dev), l_cls, torch.zeros((), device=dev)))
            return loss.sum() * B, loss.detach()
        h = self.hyp
        cfg = (self.stride[: len(feats)], self.nc, self.topk, 0.5, 6.0, (float(h.box), float(h.cls), float(h.dfl)))
        maps = [f if f.dtype == ops.compute_dtype() else f.to(ops.compute_dtype()) for f in feats]
        total, items, fg, gt_idx, t_sc = Loss2dFn.apply(cfg, g, *maps)
        self.last_assignment = (fg.bool(), gt_idx.long(), t_sc)
        return total * B, items


class v10DetectLoss:
    """utils/loss.py:727-737"""

    def __init__(self, model):
        self.one2many = v8DetectionLoss(model, tal_topk=10)
        self.one2one = v8DetectionLoss(model, tal_topk=1)

    def __call__(self, preds, batch):
        lm, im = self.one2many(preds["one2many"], batch)
        l1, i1 = self.one2one(preds["one2one"], batch)
        return lm + l1, torch.cat((im, i1))


def _postprocess(preds, max_det, nc, boxes_first):
    """preds (B, A, C) as the reference passes it (a permuted view of the head's (B, C, A) output): one HIP launch"""
    if not preds.is_cuda:
        raise Y3DError("v10 postprocess runs on the HIP kernel of post.hip: predictions must live on a HIP device")
    B, A, C = preds.shape
    y = preds.permute(0, 2, 1)
    if not (y.is_contiguous() and y.dtype == torch.float32):
        y = y.float().contiguous()
    nr = C - nc
    reg = torch.empty(B, max_det, nr, dtype=torch.float32, device=preds.device)
    scores = torch.empty(B, max_det, dtype=torch.float32, device=preds.device)
    labels = torch.empty(B, max_det, dtype=torch.int64, device=preds.device)
    ns = lib().v10_postprocess_scratch_floats(B, A, nc, max_det)  # hi-res maps: the per-image score row lives in HBM instead of LDS
    scratch = torch.empty(ns, dtype=torch.float32, device=preds.device) if ns else None
    lib().v10_postprocess(y.data_ptr(), B, C, A, nc, max_det, int(boxes_first), reg.data_ptr(), scores.data_ptr(), labels.data_ptr(),
                          scratch.data_ptr() if ns else None, ops.stream())
    return reg, scores, labels


def v10_3Dpostprocess(preds, max_det, nc=3):
    """utils/ops.py:867-880: top-k over anchors of the max-class score, then over the k*nc scores -> (reg, scores, labels)"""
    assert preds.shape[-1] == nc + 35
    return _postprocess(preds, max_det, nc, False)


def v10postprocess(preds, max_det, nc=80):
    """utils/ops.py:852-865"""
    assert 4 + nc == preds.shape[-1]
    return _postprocess(preds, max_det, nc, True)
