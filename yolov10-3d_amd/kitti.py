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


# ------------------------------------------------------------------------------------------------------------------------------
# f2: one-to-many depth fusion of the validator (models/yolov10_3D/val.py:78-102)
# ------------------------------------------------------------------------------------------------------------------------------
def aggregate_o2m_preds(predsO, predsM, thres=0.1, iou_thres=0.9, nprop=500):
    """`YOLOv10_3DDetectionValidator.aggregate_o2m_preds`: predsO (B, K, 37) / predsM (B, KM, 37) post-processed rows (regression |
    score | label, val.py:47-54) on the device -> predsO with every depth replaced by the mode of the weighted kernel density of the
    matching one-to-many depths.  One HIP launch (`y3d_kde_depth_fusion`, one block per detection); the reference loops over the
    detections in Python and fits a scikit-learn KernelDensity per detection on the host."""
    if predsO.dim() != 3 or predsM.dim() != 3 or predsO.shape[0] != predsM.shape[0] or predsO.shape[2] != predsM.shape[2]:
        raise Y3DError(f"aggregate_o2m_preds: expected (B, K, C) and (B, KM, C), got {tuple(predsO.shape)} / {tuple(predsM.shape)}")
    if predsO.device.type != "cuda":
        raise Y3DError("aggregate_o2m_preds needs the predictions on a HIP device")
    O, M = predsO.detach().float().contiguous(), predsM.detach().float().contiguous()
    out = torch.empty_like(O)
    B, K, C = O.shape
    lib().kde_depth_fusion(O.data_ptr(), B, K, M.data_ptr(), M.shape[1], C, float(thres), float(iou_thres), int(nprop), out.data_ptr(), ops.stream())
    return out


# ------------------------------------------------------------------------------------------------------------------------------
# f3: image side of the input pipeline (data/datasets/kitti.py:132-206) on the device
# ------------------------------------------------------------------------------------------------------------------------------
def get_affine_transform(center, scale, output_size, inv=False):
    """kitti_utils.py:423-464 (rot = 0, shift = 0, as kitti.py:192 calls it): the crop's affine map, host side (six numbers per image).
    center (2,), scale = crop size (2,) or scalar, output_size (W, H) -> trans (2, 3) float64 [, trans_inv]"""
    import numpy as np
    center = np.asarray(center, np.float64)
    scale = np.asarray(scale, np.float64) if np.ndim(scale) else np.array([scale, scale], np.float64)
    src, dst = np.zeros((3, 2), np.float32), np.zeros((3, 2), np.float32)
    src[0] = center
    src[1] = center + np.array([0, scale[0] * -0.5])
    dst[0] = [output_size[0] * 0.5, output_size[1] * 0.5]
    dst[1] = np.array([output_size[0] * 0.5, output_size[1] * 0.5], np.float32) + np.array([0, output_size[0] * -0.5], np.float32)
    for p in (src, dst):  # third point: the right-angle companion (kitti_utils.py get_3rd_point)
        d = p[0] - p[1]
        p[2] = p[1] + np.array([-d[1], d[0]], np.float32)

    def solve(a, b):  # cv2.getAffineTransform: the exact map through three point pairs
        return np.linalg.solve(np.hstack((a.astype(np.float64), np.ones((3, 1)))), b.astype(np.float64)).T.copy()

    trans = solve(src, dst)
    return (trans, solve(dst, src)) if inv else trans


def affine_transform(pt, t):
    """kitti_utils.py:467-470: a point through the 2x3 map (labels: box corners, projected 3D centre)"""
    import numpy as np
    return np.dot(t, np.array([pt[0], pt[1], 1.0], dtype=np.float32).T)[:2]


def augment_images(imgs, partners, flips, trans_inv, out_wh, mode="float"):
    """The image work of `KITTIDataset.__getitem__` for a batch, one HIP launch: `imgs[b]` (H, W, 3) uint8 RGB device tensors as
    decoded, `partners[b]` the mixup partner or None (kitti.py:160-189), `flips[b]` bool (:147-149), `trans_inv[b]` the (2, 3) crop
    matrix (:192), `out_wh` = the dataset's resolution (W, H).
    mode "float": (B, 3, H, W) float32 in [0, 1] — the reference's `img` tensor, bit for bit; mode "uint8": (B, H, W, 3) uint8, which
    the stem consumes directly (the /255 and the layout change happen in `y3d_stem_im2col_u8`)."""
    import numpy as np
    B = len(imgs)
    if B == 0 or any((not t.is_cuda) or t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3 for t in imgs):
        raise Y3DError("augment_images: images must be (H, W, 3) uint8 tensors on a HIP device")
    dev = imgs[0].device
    imgs = [t.contiguous() for t in imgs]
    parts = [None if q is None else q.contiguous() for q in partners]
    for t, q in zip(imgs, parts):
        if q is not None and (q.shape != t.shape or q.dtype != torch.uint8 or not q.is_cuda):
            raise Y3DError("augment_images: a mixup partner must match its image (the reference mixes equal-sized frames only, kitti.py:176)")
    W, H = int(out_wh[0]), int(out_wh[1])
    src = torch.tensor([t.data_ptr() for t in imgs], dtype=torch.int64).to(dev)
    src2 = torch.tensor([0 if q is None else q.data_ptr() for q in parts], dtype=torch.int64).to(dev) if any(q is not None for q in parts) else None
    hw = torch.tensor([[t.shape[0], t.shape[1]] for t in imgs], dtype=torch.int32).to(dev)
    fl = torch.tensor([int(bool(f)) for f in flips], dtype=torch.int32).to(dev)
    ti = torch.tensor(np.stack([np.asarray(t, np.float64).reshape(6) for t in trans_inv]), dtype=torch.float64).to(dev)
    if mode == "float":
        out = torch.empty(B, 3, H, W, dtype=torch.float32, device=dev)
    elif mode == "uint8":
        out = torch.empty(B, H, W, 3, dtype=torch.uint8, device=dev)
    else:
        raise ValueError("mode must be 'float' or 'uint8'")
    lib().kitti_image_aug(src.data_ptr(), src2.data_ptr() if src2 is not None else None, hw.data_ptr(), fl.data_ptr(), ti.data_ptr(), B, H, W,
                          0 if mode == "float" else 1, out.data_ptr(), ops.stream())
    return out
