"""KITTI decode of the post-processed detections on the device — the eval tail behind `v10_3Dpostprocess`.

Mirrors `KITTIDataset.decode_preds` / `decode_preds_eval` (data/datasets/kitti.py:515-576), which the reference's validator calls
from `_prepare_preds` (models/yolov10_3D/val.py:210-214) on CPU with a python loop and `.item()` per detection: heading-bin argmax +
residual -> alpha, size residual + class mean size, centre back-projection through the (inverse-affine'd) calibration, rotation_y,
score = sigmoid(cls) * exp(-depth log-variance), threshold.  Here it is one launch over all B*K rows (`y3d_kitti_decode`).
"""
from __future__ import annotations

import torch

from . import ops
from ._lib import Y3DError, lib

# (h, w, l) per class, kitti.py:38-41
CLS_MEAN_SIZE = ((1.52563191462, 1.62856739989, 3.88311640418), (1.76255119, 0.66068622, 0.84422524), (1.73698127, 0.59706367, 1.76282397))


def _calib_rows(calibs, B, dev):
    if torch.is_tensor(calibs):
        c = calibs.to(dev, torch.float64).reshape(B, 6)
    else:  # the reference's Calibration objects (kitti_utils.py:178-196) or anything with the same six attributes
        c = torch.tensor([[float(k.cu), float(k.cv), float(k.fu), float(k.fv), float(k.tx), float(k.ty)] for k in calibs], dtype=torch.float64, device=dev)
    if c.shape != (B, 6):
        raise Y3DError(f"decode_preds: {B} images but calibration rows of shape {tuple(c.shape)}")
    return c.contiguous()


def decode_preds_device(preds, calibs, ratio_pad, inv_trans, undo_augment=True, threshold=0.001, cls_mean_size=CLS_MEAN_SIZE,
                        use_camera_dis=False):
    """preds (B, K, 37) on the device -> rows (B, K, 14) float64 [cls, alpha, x1, y1, x2, y2, h, w, l, x, y, z, ry, score] and
    keep (B, K) bool, both on the device, no host synchronisation.  ratio_pad: (B, 2, 2) as collated (kitti.py:404, 431) or (B, 2)."""
    if preds.dim() != 3 or preds.shape[-1] != 37:
        raise Y3DError(f"decode_preds: expected (B, K, 37) predictions, got {tuple(preds.shape)}")
    dev = preds.device
    if dev.type != "cuda":
        raise Y3DError("decode_preds needs the predictions on a HIP device")
    B, K, _ = preds.shape
    p = preds.detach().float().contiguous()
    calib = _calib_rows(calibs, B, dev)
    rp = torch.as_tensor(ratio_pad).to(dev, torch.float64)
    ratio = (rp[:, 0] if rp.dim() == 3 else rp).reshape(B, 2).contiguous()
    inv = None
    if undo_augment:
        inv = torch.stack([torch.as_tensor(t, dtype=torch.float64).reshape(2, 3) for t in inv_trans]).to(dev).contiguous()
        if inv.shape[0] != B:
            raise Y3DError(f"decode_preds: {B} images but {inv.shape[0]} inverse transforms")
    ms = torch.as_tensor(cls_mean_size, dtype=torch.float64).reshape(-1, 3).to(dev).contiguous()
    out = torch.empty(B, K, 14, dtype=torch.float64, device=dev)
    keep = torch.empty(B, K, dtype=torch.uint8, device=dev)
    lib().kitti_decode(p.data_ptr(), B, K, calib.data_ptr(), ratio.data_ptr(), inv.data_ptr() if inv is not None else None, ms.data_ptr(),
                       ms.shape[0], int(bool(use_camera_dis)), float(threshold), out.data_ptr(), keep.data_ptr(), ops.stream())
    return out, keep.bool()


def decode_preds(preds, calibs, im_files, ratio_pad, inv_trans, undo_augment=True, threshold=0.001, cls_mean_size=CLS_MEAN_SIZE,
                 use_camera_dis=False):
    """kitti.py:519-576: {im_file: [[cls, alpha, x1, y1, x2, y2, h, w, l, x, y, z, ry, score], ...]} (one device->host copy)"""
    rows, keep = decode_preds_device(preds, calibs, ratio_pad, inv_trans, undo_augment, threshold, cls_mean_size, use_camera_dis)
    rows, keep = rows.cpu(), keep.cpu()
    return {f: rows[i][keep[i]].tolist() for i, f in enumerate(im_files)}


def decode_preds_eval(preds, calibs, im_files, ratio_pad, inv_trans, undo_augment=True, threshold=0.001, **kw):
    """kitti.py:515-517"""
    return decode_preds(preds, calibs, im_files, ratio_pad, inv_trans, undo_augment=undo_augment, threshold=threshold, **kw)
