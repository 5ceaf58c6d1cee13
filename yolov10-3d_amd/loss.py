"""Losses + task-aligned assignment for the YOLOv10 (2D) and YOLOv10-3D heads — device side.

Mirrors the reference interfaces (utils/loss.py: v8DetectionLoss :157, v10DetectLoss :727, DetectLoss3d :740,
DDDetectionLoss :774; utils/tal.py: TaskAlignedAssigner :19, TaskAlignedAssigner3d :355;
utils/keypoint_utils.py; utils/metrics.py:78 bbox_iou).  All arithmetic is fp32 (the reference runs these
under autocast's fp32 policy), on the tensors' own device.

Both training losses (3D: DetectLoss3d / DDDetectionLoss, 2D: v10DetectLoss / v8DetectionLoss — target padding, assignment, the loss
terms and the gradient wrt the head maps) run on the fused HIP kernels of csrc/tal_loss3d.hip / tal_loss2d.hip (`Loss3dFn`,
`Loss2dFn`).  Tie rule of the top-k is pinned to lowest-index-first.  Torch formulations of the assigners used
as test comparators live in tests/torch_assigners.py, not here.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

import ctypes

from . import ops
from ._lib import Y3DError, lib
from .modules import make_anchors


_OVERFLOW_PENDING = []  # (event, pinned int32[1], cap) of pad_targets launches whose per-image box count has not been looked at yet
TARGET_CAP = 64  # rows per image the assigner kernels take (tal_loss3d.hip / tal_loss2d.hip)


def check_target_overflow(wait=False):
    """Raise Y3DError if a `pad_targets` launch met an image with more boxes than its capacity (its surplus boxes were dropped).
    The count comes back through a pinned host word behind an event: `pad_targets` polls the finished ones at its next call (the
    training step keeps running without a host synchronisation and the error surfaces a step or two late); wait=True synchronises
    on all of them (validators, tests)."""
    keep = []
    for ev, host, cap in _OVERFLOW_PENDING:
        if wait:
            ev.synchronize()
        if not ev.query():
            keep.append((ev, host, cap))
            continue
        n = int(host[0])
        if n > cap:
            _OVERFLOW_PENDING[:] = keep
            raise Y3DError(f"pad_targets: an image of a recent batch has {n} ground-truth boxes, the assigner kernels take at most {cap} per "
                           f"image (KITTI's max_objs is 50); its last {n - cap} boxes were not trained on")
    _OVERFLOW_PENDING[:] = keep


def pad_targets(rows, B, width, scale_xy, cap=TARGET_CAP):
    """utils/loss.py:795-810 on the HIP kernel `y3d_pad_targets`: ragged rows (nbox, 1+width) -> (B, cap, width) + the device-side
    largest per-image box count.  The reference sizes the padded tensor with a host-side `counts.max()`; here the capacity is fixed
    (`cap`, the assigner kernels' limit of 64 rows; KITTI's max_objs is 50, data/datasets/kitti.py:23) and the count stays on the
    device, so the step has no host synchronisation.  An image with more than `cap` boxes (crowded 2D data, mosaics) is NOT trained
    on silently truncated targets: the true count travels to a pinned host word asynchronously and `check_target_overflow` raises
    (ADVICE round 2).  -> (gt (B, cap, width) fp32, n_used (1,) int32: min(largest count, cap))"""
    if not rows.is_cuda:
        raise Y3DError("pad_targets runs on the HIP kernel of tal_loss3d.hip: the batch must live on a HIP device (no CPU fallback)")
    capturing = torch.cuda.is_current_stream_capturing()
    if not capturing:
        check_target_overflow()
    rows = rows.float().contiguous()
    out = torch.empty(B, cap, width, dtype=torch.float32, device=rows.device)
    n_used = torch.empty(2, dtype=torch.int32, device=rows.device)  # [min(count, cap), true count]
    lib().pad_targets(rows.data_ptr(), rows.shape[0], width, B, cap, float(scale_xy[0]), float(scale_xy[1]), out.data_ptr(), n_used.data_ptr(),
                      ops.stream())
    if capturing:  # no read-back inside a hipGraph capture: graph.GraphedTrainStep reads this (static) word back after every replay
        _CAPTURED_COUNTS.append((n_used, cap))
    else:
        watch_target_count(n_used, cap)
    return out, n_used[:1]


_CAPTURED_COUNTS = []  # (n_used, cap) of the pad_targets launches recorded into the hipGraph being captured


def watch_target_count(n_used, cap):
    """queue an asynchronous read-back of n_used[1] (the true largest per-image box count) for check_target_overflow"""
    host = torch.empty(1, dtype=torch.int32).pin_memory()
    host.copy_(n_used[1:2], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    _OVERFLOW_PENDING.append((ev, host, cap))
    if len(_OVERFLOW_PENDING) > 64:  # a caller that never lets the stream drain: look now
        check_target_overflow(wait=True)


def _flatten_maps(feats):
    """list of (B, no, H, W) NHWC maps -> (B, A, no) fp32 (a free view per level + one cat)"""
    B, no = feats[0].shape[:2]
    return torch.cat([f.permute(0, 2, 3, 1).reshape(B, -1, no) for f in feats], 1).float()


def _loss3d_set(cfg, gt, n_used, calib, mean_sizes, map_ptrs, psw, grad_ptrs, gsw, Hs, Ws, B, dtype, dev):
    """one head set on the HIP kernels: assignment + six loss items + gradient rows written at grad_ptrs (pixel stride gsw).
    -> (items[6], fg (B, A) uint8, gt_idx (B, A) int32, target_scores (B, A, nc))"""
    L = lib()
    strides, nc, topk, alpha, beta, gamma, w = cfg[:7]
    mode = cfg[7] if len(cfg) > 7 else 11  # assigner mode bits (include/y3d.h: y3d_tal3d_assign); default tal_2d | tal_3d | constrain_anchors, l1
    dt, st, nl = ops.code(dtype), ops.stream(), len(map_ptrs)
    A = sum(h * w_ for h, w_ in zip(Hs, Ws))
    n = gt.shape[1]
    PV = ctypes.c_void_p * nl
    c_maps, c_grads = PV(*map_ptrs), PV(*grad_ptrs)
    c_psw, c_gsw = (ctypes.c_int64 * nl)(*psw), (ctypes.c_int64 * nl)(*gsw)
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
                   alpha, beta, gamma, int(mode), scratch.data_ptr(), fg.data_ptr(), gi.data_ptr(), ts.data_ptr(), scal.data_ptr(),
                   n_used.data_ptr() if n_used is not None else None, st)
    nblk = (B * A + 255) // 256
    part = torch.empty(nblk * 6, dtype=torch.float32, device=dev)
    items = torch.empty(6, dtype=torch.float32, device=dev)
    L.loss3d(dt, nl, c_maps, c_psw, c_grads, c_gsw, c_H, c_W, c_st, B, nc, gt.data_ptr(), n, fg.data_ptr(), gi.data_ptr(), ts.data_ptr(),
             scal.data_ptr(), w[0], w[1], w[2], w[3], w[4], w[5], 1.0, part.data_ptr(), items.data_ptr(), st)
    return items, fg, gi, ts


class Loss3dFn(torch.autograd.Function):
    """One head set: task-aligned assignment (no grad) + the six 3D loss terms + d(sum of terms)/d(head maps), on the fused HIP
    kernels.  apply(cfg, gt(B,n,17), n_used, calib, mean_sizes, *maps) -> (sum_of_items, items[6], fg_mask, target_gt_idx, target_scores);
    n_used: device int32 from pad_targets (or None: walk all n rows)."""

    @staticmethod
    def forward(ctx, cfg, gt, n_used, calib, mean_sizes, *maps):
        dtype, dev = maps[0].dtype, maps[0].device
        for m in maps:
            if not (m.is_cuda and ops.px_dense(m)):
                raise Y3DError("Loss3dFn: head maps must be pixel-dense NHWC tensors on a HIP device")
        B, no = maps[0].shape[:2]
        Hs, Ws = [m.shape[2] for m in maps], [m.shape[3] for m in maps]
        grads = [ops.nhwc_empty(B, no, h, w_, dtype, dev) for h, w_ in zip(Hs, Ws)]
        items, fg, gi, ts = _loss3d_set(cfg, gt.float().contiguous(), n_used, calib.float().contiguous(), mean_sizes.float().contiguous(),
                                        [m.data_ptr() for m in maps], [m.stride(3) for m in maps], [g.data_ptr() for g in grads],
                                        [no] * len(maps), Hs, Ws, B, dtype, dev)
        ctx.save_for_backward(*grads)
        ctx.set_materialize_grads(False)  # no zero-filled gradients for the assignment outputs
        total = items.sum()
        ctx.mark_non_differentiable(items, fg, gi, ts)
        return total, items, fg, gi, ts

    @staticmethod
    def backward(ctx, d_total, *unused):
        grads = ctx.saved_tensors
        if d_total is None:
            return (None,) * (5 + len(grads))
        return (None, None, None, None, None, *[g * d_total.to(g.dtype) for g in grads])


class DualLoss3dFn(torch.autograd.Function):
    """Both head sets of a step on the head's own (B, 2*no, H, W) maps ([one-to-one | one-to-many] channels, as
    v10Detect3d.forward_train_fused writes them): the gradient rows of both sets go into ONE NHWC tensor per level and the backward
    is one broadcast multiply per level.  With the two sets as separate autograd inputs (channel slices of that map) autograd
    zero-fills a full map per slice, copies, adds and re-lays the sum out: 7 elementwise launches per level instead of 1.
    apply(cfg_o2o, cfg_o2m, gt, n_used, calib, mean_sizes, *maps) -> (total_o2o, items_o2o, total_o2m, items_o2m, fg1, gi1, ts1, fgm, gim, tsm)"""

    @staticmethod
    def forward(ctx, cfg1, cfgm, gt, n_used, calib, mean_sizes, *maps):
        dtype, dev = maps[0].dtype, maps[0].device
        for m in maps:
            if not (m.is_cuda and ops.px_dense(m)):
                raise Y3DError("DualLoss3dFn: head maps must be pixel-dense NHWC tensors on a HIP device")
        B, no2 = maps[0].shape[:2]
        no, esz = no2 // 2, maps[0].element_size()
        Hs, Ws = [m.shape[2] for m in maps], [m.shape[3] for m in maps]
        grads = [ops.nhwc_empty(B, no2, h, w_, dtype, dev) for h, w_ in zip(Hs, Ws)]
        gt, calib, mean_sizes = gt.float().contiguous(), calib.float().contiguous(), mean_sizes.float().contiguous()
        outs = []
        for cfg, off in ((cfg1, 0), (cfgm, no)):
            outs.append(_loss3d_set(cfg, gt, n_used, calib, mean_sizes, [m.data_ptr() + off * esz for m in maps], [m.stride(3) for m in maps],
                                    [g.data_ptr() + off * esz for g in grads], [no2] * len(maps), Hs, Ws, B, dtype, dev))
        ctx.save_for_backward(*grads)
        ctx.no = no
        ctx.set_materialize_grads(False)
        (i1, fg1, gi1, ts1), (im, fgm, gim, tsm) = outs
        ctx.mark_non_differentiable(i1, im, fg1, gi1, ts1, fgm, gim, tsm)
        return i1.sum(), i1, im.sum(), im, fg1, gi1, ts1, fgm, gim, tsm

    @staticmethod
    def backward(ctx, d1, _i1, dm, *unused):
        grads = ctx.saved_tensors
        if d1 is None and dm is None:
            return (None,) * (6 + len(grads))
        no = ctx.no
        z = torch.zeros((), dtype=torch.float32, device=grads[0].device)
        scale = torch.cat(((d1 if d1 is not None else z).float().reshape(1).expand(no), (dm if dm is not None else z).float().reshape(1).expand(no)))
        scale = scale.to(grads[0].dtype).view(1, 2 * no, 1, 1)
        return (None, None, None, None, None, None, *[g * scale for g in grads])


class DDDetectionLoss:
    """utils/loss.py:774-963"""

    def __init__(self, model, tal_topk=10):
        h = model.args
        m = model.model[-1]
        self.hyp = h
        self.stride = [float(s) for s in m.stride]
        self.nc, self.no = m.nc, m.no
        # assigner modes of cfg/default.yaml:116-119 (utils/tal.py:465-497), as the mode bits of y3d_tal3d_assign
        if not (h.tal_2d or h.tal_3d):
            raise RuntimeError("Either 2D or 3D assignment or both has to be selected!")  # the reference's message, tal.py:484
        if h.kps_dist_metric not in ("l1", "l2"):
            raise ValueError(f"kps_dist_metric {h.kps_dist_metric!r}: 'l1' or 'l2' (utils/tal.py:465-470)")
        self.mode = (1 if h.tal_2d else 0) | (2 if h.tal_3d else 0) | (4 if h.kps_dist_metric == "l2" else 0) | (8 if h.constrain_anchors else 0)
        self.topk = tal_topk
        if getattr(h, "distillation", False):
            raise NotImplementedError("distillation needs the DINOv2 teacher (network); pinned off (SURVEY §0.5)")

    GT_KEYS = ("batch_idx", "cls", "bboxes", "center_2d", "size_2d", "center_3d", "size_3d", "depth", "heading_bin", "heading_res")

    def targets(self, batch, B, H, W, dev):
        """padded ground truth of one step (loss.py:848-856): (gt (B, cap, 17), n_used) or None when the batch has no box at all"""
        rows = torch.cat([batch[k].to(dev).float().view(batch[k].shape[0], -1) for k in self.GT_KEYS], 1)
        if rows.shape[0] == 0:
            return None
        return pad_targets(rows, B, 17, (W * self.stride[0], H * self.stride[0]))

    def __call__(self, preds, batch, embeddings=None, targets=None):
        """`targets`: the result of `self.targets(...)` when the caller shares it between the two head sets of a step"""
        feats = preds[1] if isinstance(preds, tuple) else preds
        dev = feats[0].device
        if not feats[0].is_cuda:
            raise Y3DError("the 3D loss runs on the HIP kernels of tal_loss3d.hip: head maps must live on a HIP device (no CPU fallback)")
        B = feats[0].shape[0]
        H, W = feats[0].shape[2:]
        if targets is None:
            targets = self.targets(batch, B, H, W, dev)
        if targets is None:
            loss = torch.zeros(6, device=dev)
            return loss.sum() * B, loss  # reference: graph-less zeros (loss.py:873-877); callers skip the step
        g, n_used = targets
        maps = [f if f.dtype == ops.compute_dtype() else f.to(ops.compute_dtype()) for f in feats]
        total, items, fg, gt_idx, t_sc = Loss3dFn.apply(self.cfg(len(feats)), g, n_used, batch["calib"].to(dev), batch["mean_sizes"].to(dev), *maps)
        self._assignment = (fg, gt_idx, t_sc)
        return total * B, items

    def cfg(self, nl):
        h = self.hyp
        return (self.stride[:nl], self.nc, self.topk, float(h.tal_alpha), float(h.tal_beta), float(h.tal_gamma),
                (float(h.loss2d), float(h.cls), float(h.depth), float(h.offset3d), float(h.size3d), float(h.heading)), self.mode)

    @property
    def last_assignment(self):
        """(fg_mask bool, target_gt_idx int64, target_scores) of the last call, in the reference's dtypes (converted on access)"""
        fg, gi, ts = self._assignment
        return fg.bool(), gi.long(), ts


class DetectLoss3d:
    """utils/loss.py:740-771"""

    def __init__(self, model):
        self.one2many = DDDetectionLoss(model, tal_topk=model.args.tal_topk)
        self.one2one = DDDetectionLoss(model, tal_topk=1)

    @staticmethod
    def _shared_maps(maps, o2o, o2m, no):
        """`maps`: the head's own (B, 2*no, H, W) tensors (preds["_y3d_maps"], written by v10Detect3d.forward_train_fused), accepted
        only when the two head sets really are their channel halves [one-to-one | one-to-many]; else None"""
        if not maps or len(maps) != len(o2o) or len(o2o) != len(o2m):
            return None
        for base, a, b in zip(maps, o2o, o2m):
            if (base.dim() != 4 or base.shape[1] != 2 * no or a.shape[1] != no or b.shape[1] != no or base.dtype != ops.compute_dtype()
                    or not ops.px_dense(base) or a.stride() != base.stride() or b.stride() != base.stride()
                    or a.data_ptr() != base.data_ptr() or b.data_ptr() != base.data_ptr() + no * base.element_size()):
                return None
        return list(maps)

    def __call__(self, preds, batch):
        o2o = preds["one2one"][1] if isinstance(preds["one2one"], tuple) else preds["one2one"]
        dev = o2o[0].device
        B, (H, W) = o2o[0].shape[0], o2o[0].shape[2:]
        tg = self.one2one.targets(batch, B, H, W, dev) if o2o[0].is_cuda else None  # padded once per step
        kw = {"targets": tg} if tg is not None else {}
        o2m = preds.get("one2many", None)
        if o2m and tg is not None:
            o2m = o2m[1] if isinstance(o2m, tuple) else o2m
            bases = self._shared_maps(preds.get("_y3d_maps"), o2o, o2m, self.one2one.no)
            if bases is not None:
                g, n_used = tg
                nl = len(bases)
                t1, i1, tm, im, fg1, gi1, ts1, fgm, gim, tsm = DualLoss3dFn.apply(self.one2one.cfg(nl), self.one2many.cfg(nl), g, n_used, batch["calib"].to(dev),
                                                                                  batch["mean_sizes"].to(dev), *bases)
                self.one2one._assignment, self.one2many._assignment = (fg1, gi1, ts1), (fgm, gim, tsm)
                return tm * B + t1 * B, torch.cat((im, i1))
        l1, i1 = self.one2one(preds["one2one"], batch, embeddings=preds.get("o2o_embs"), **kw)
        if o2m:
            lm, im = self.one2many(preds["one2many"], batch, embeddings=preds.get("o2m_embs"), **kw)
            return lm + l1, torch.cat((im, i1))
        return torch.zeros(1), i1


class Loss2dFn(torch.autograd.Function):
    """2D head set: assignment + (box, cls, dfl) + gradient wrt the head maps on the fused HIP kernels of tal_loss2d.hip.
    apply(cfg, gt(B,n,5), n_used, *maps) -> (sum_of_items, items[3], fg_mask, target_gt_idx, target_scores)"""

    @staticmethod
    def forward(ctx, cfg, gt, n_used, *maps):
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
                       gi.data_ptr(), ts.data_ptr(), scal.data_ptr(), n_used.data_ptr() if n_used is not None else None, st)
        nblk = (B * A + 255) // 256
        part = torch.empty(nblk * 3, dtype=torch.float32, device=dev)
        items = torch.empty(3, dtype=torch.float32, device=dev)
        L.loss2d(dt, nl, c_maps, c_psw, c_grads, c_gsw, c_H, c_W, c_st, B, nc, gt.data_ptr(), n, fg.data_ptr(), gi.data_ptr(), ts.data_ptr(),
                 scal.data_ptr(), w[0], w[1], w[2], 1.0, part.data_ptr(), items.data_ptr(), st)
        ctx.save_for_backward(*grads)
        ctx.set_materialize_grads(False)  # no zero-filled gradients for the assignment outputs
        total = items.sum()
        ctx.mark_non_differentiable(items, fg, gi, ts)
        return total, items, fg, gi, ts

    @staticmethod
    def backward(ctx, d_total, *unused):
        if d_total is None:
            return (None,) * (3 + len(ctx.saved_tensors))
        return (None, None, None, *[g * d_total.to(g.dtype) for g in ctx.saved_tensors])


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

    def targets(self, batch, B, H, W, dev):
        """padded ground truth (loss.py:223-226): (gt (B, cap, 5), n_used) or None when the batch has no box at all"""
        rows = torch.cat((batch["batch_idx"].view(-1, 1), batch["cls"].view(-1, 1), batch["bboxes"]), 1).to(dev).float()
        if rows.shape[0] == 0:
            return None
        return pad_targets(rows, B, 5, (W * self.stride[0], H * self.stride[0]))

    def __call__(self, preds, batch, targets=None):
        feats = preds[1] if isinstance(preds, tuple) else preds
        dev = feats[0].device
        if not feats[0].is_cuda:
            raise Y3DError("the 2D loss runs on the HIP kernels of tal_loss2d.hip: head maps must live on a HIP device (no CPU fallback)")
        B = feats[0].shape[0]
        H, W = feats[0].shape[2:]
        if targets is None:
            targets = self.targets(batch, B, H, W, dev)
        if targets is None:
            # no boxes at all: background-only classification loss (loss.py:244), dense BCE against zero targets
            sc = _flatten_maps(feats)[..., self.reg_max * 4:]
            l_cls = F.binary_cross_entropy_with_logits(sc, torch.zeros_like(sc), reduction="none").sum() * self.hyp.cls
            loss = torch.stack((torch.zeros((), device=dev), l_cls, torch.zeros((), device=dev)))
            return loss.sum() * B, loss.detach()
        g, n_used = targets
        h = self.hyp
        cfg = (self.stride[: len(feats)], self.nc, self.topk, 0.5, 6.0, (float(h.box), float(h.cls), float(h.dfl)))
        maps = [f if f.dtype == ops.compute_dtype() else f.to(ops.compute_dtype()) for f in feats]
        total, items, fg, gt_idx, t_sc = Loss2dFn.apply(cfg, g, n_used, *maps)
        self.last_assignment = (fg.bool(), gt_idx.long(), t_sc)
        return total * B, items


class v10DetectLoss:
    """utils/loss.py:727-737"""

    def __init__(self, model):
        self.one2many = v8DetectionLoss(model, tal_topk=10)
        self.one2one = v8DetectionLoss(model, tal_topk=1)

    def __call__(self, preds, batch):
        f0 = preds["one2many"][1] if isinstance(preds["one2many"], tuple) else preds["one2many"]
        tg = self.one2many.targets(batch, f0[0].shape[0], f0[0].shape[2], f0[0].shape[3], f0[0].device)  # padded once per step
        kw = {"targets": tg} if tg is not None else {}
        lm, im = self.one2many(preds["one2many"], batch, **kw)
        l1, i1 = self.one2one(preds["one2one"], batch, **kw)
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
