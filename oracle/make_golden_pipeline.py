"""TEST INFRASTRUCTURE — golden vectors for the two callers either side of the hot path, minted from the
REFERENCE in the build container:

    python -m oracle.make_golden_pipeline      # writes tests/golden/kde_fusion.npz, tests/golden/kitti_aug.npz

* kde_fusion: `YOLOv10_3DDetectionValidator.aggregate_o2m_preds` (models/yolov10_3D/val.py:78-102) on synthetic post-processed
  one-to-one / one-to-many detections (scikit-learn's KernelDensity is the third-party arithmetic; version recorded in the file).
* kitti_aug:  `KITTIDataset.__getitem__` (data/datasets/kitti.py:116-442) on a SYNTHETIC three-image KITTI directory written to a
  temp dir: random flip / crop / mixup, PIL's FLIP_LEFT_RIGHT + Image.blend + AFFINE/BILINEAR transform (Pillow is the third-party
  arithmetic; version recorded), `/255`, CHW.  The dataset's 1280x384 output resolution is lowered to keep the fixture small.
  OpenCV is absent from the image: `cv2.getAffineTransform` (kitti_utils.py:459-463, the exact affine map through three point
  pairs) is supplied as a float64 linear solve for the duration of the run.
Fixtures are data only (inputs, the reference's outputs, the random decisions taken): no reference source text is stored.
"""
from __future__ import annotations

import os
import shutil
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_shim as R  # noqa: E402
from oracle.make_golden import save  # noqa: E402


def affine_from_points(src, dst):
    """cv2.getAffineTransform: the 2x3 matrix M with M @ [x, y, 1] = dst for three point pairs (float64)"""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    A = np.hstack((src, np.ones((3, 1))))
    return np.linalg.solve(A, dst).T.copy()


def kde_fixture():
    import sklearn
    from ultralytics.models.yolov10_3D.val import YOLOv10_3DDetectionValidator as V

    g = torch.Generator().manual_seed(21)
    B, K, KM = 2, 50, 250
    O = torch.zeros(B, K, 37)
    xy = torch.rand(B, K, 2, generator=g) * torch.tensor([1100.0, 300.0])
    wh = 30 + 120 * torch.rand(B, K, 2, generator=g)
    O[..., 0:2], O[..., 2:4] = xy, xy + wh
    O[..., 4:33] = torch.randn(B, K, 29, generator=g)
    O[..., -4] = 5 + 55 * torch.rand(B, K, generator=g)              # depth
    O[..., -3] = -1.5 + 4.5 * torch.rand(B, K, generator=g)          # depth log-variance: exp(-u) > 0.1  <=>  u < 2.30
    O[..., -2] = torch.rand(B, K, generator=g)                       # score
    O[..., -1] = torch.randint(0, 3, (B, K), generator=g).float()    # label
    # one-to-many detections: 0..8 near-duplicates of every one-to-one box (IoU > 0.9 for most), some with another label,
    # some with a large uncertainty, the rest unrelated boxes
    M = torch.zeros(B, KM, 37)
    for b in range(B):
        rows = []
        for j in range(K):
            n = int(torch.randint(0, 9, (1,), generator=g))
            for _ in range(n):
                r = O[b, j].clone()
                r[0:4] += torch.randn(4, generator=g) * 0.012 * wh[b, j].repeat(2)
                r[-4] += torch.randn((), generator=g) * 1.5
                r[-3] = -1.5 + 4.5 * torch.rand((), generator=g)
                if torch.rand((), generator=g) < 0.15:
                    r[-1] = (r[-1] + 1) % 3
                rows.append(r)
        rows = rows[:KM]
        while len(rows) < KM:
            r = torch.zeros(37)
            p = torch.rand(2, generator=g) * torch.tensor([1100.0, 300.0])
            r[0:2], r[2:4] = p, p + 20 + 100 * torch.rand(2, generator=g)
            r[-4], r[-3], r[-1] = 5 + 55 * torch.rand((), generator=g), torch.rand((), generator=g), float(torch.randint(0, 3, (1,), generator=g))
            rows.append(r)
        perm = torch.randperm(KM, generator=g)
        M[b] = torch.stack(rows)[perm]
    out = V.aggregate_o2m_preds(None, O.clone(), M.clone())
    changed = int((out[..., -4] != O[..., -4]).sum())
    print(f"kde_fusion: {changed} of {B * K} depths fused; scikit-learn {sklearn.__version__}, numpy {np.__version__}")
    save("kde_fusion", predsO=O, predsM=M, fused=out, versions=np.array([sklearn.__version__, np.__version__]))


P2 = "7.215377e+02 0.000000e+00 6.095593e+02 4.485728e+01 0.000000e+00 7.215377e+02 1.728540e+02 2.163791e-01 0.000000e+00 0.000000e+00 1.000000e+00 2.745884e-03"
R0 = "9.999239e-01 9.837760e-03 -7.445048e-03 -9.869795e-03 9.999421e-01 -4.278459e-03 7.402527e-03 4.351614e-03 9.999631e-01"
TR = "7.533745e-03 -9.999714e-01 -6.166020e-04 -4.069766e-03 1.480249e-02 7.280733e-04 -9.998902e-01 -7.631618e-02 9.998621e-01 7.523790e-03 1.480755e-02 -2.717806e-01"


def write_synthetic_kitti(root, n=3, W=320, H=96):
    """three random RGB images with calibration files and labels in KITTI's directory / text formats"""
    g = np.random.RandomState(5)
    for sub in ("training/image_2", "training/calib", "training/label_2", "ImageSets"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    from PIL import Image
    imgs = []
    for i in range(n):
        yy, xx = np.mgrid[0:H, 0:W]
        base = np.stack([127 + 100 * np.sin(xx / (17.0 + 5 * i) + c) * np.cos(yy / (11.0 + 3 * c)) for c in range(3)], -1)
        img = np.clip(base + g.randint(-40, 41, size=(H, W, 3)), 0, 255).astype(np.uint8)
        Image.fromarray(img, "RGB").save(os.path.join(root, "training/image_2", f"{i:06d}.png"))
        imgs.append(img)
        with open(os.path.join(root, "training/calib", f"{i:06d}.txt"), "w") as f:
            # the projection is scaled to the small synthetic image (fu, cu, ... of KITTI's 1242-wide frames divided by 3.88)
            p2 = [float(v) for v in P2.split()]
            for q in (0, 2, 3, 5, 6, 7):
                p2[q] /= 3.88
            ps = " ".join(f"{v:e}" for v in p2)
            f.write(f"P0: {ps}\nP1: {ps}\nP2: {ps}\nP3: {ps}\nR0_rect: {R0}\nTr_velo_to_cam: {TR}\nTr_imu_to_velo: {TR}\n")
        with open(os.path.join(root, "training/label_2", f"{i:06d}.txt"), "w") as f:
            for k in range(2 + i):
                z = 8.0 + 9.0 * k + i
                x = -3.0 + 2.5 * k
                cls = ["Car", "Pedestrian", "Cyclist"][(k + i) % 3]
                h, w, l = [(1.5, 1.6, 3.9), (1.75, 0.65, 0.85), (1.7, 0.6, 1.75)][(k + i) % 3]
                u = 185.96 * x / z + 157.1
                v = 185.96 * 1.65 / z + 44.55
                bw, bh = 185.96 * l / z * 0.6, 185.96 * h / z
                f.write(f"{cls} 0.00 {k % 2} -1.57 {u - bw / 2:.2f} {v - bh:.2f} {u + bw / 2:.2f} {v:.2f} {h:.2f} {w:.2f} {l:.2f} {x:.2f} 1.65 {z:.2f} {-1.2 + 0.7 * k:.2f}\n")
    with open(os.path.join(root, "ImageSets", "train.txt"), "w") as f:
        f.write("\n".join(f"{i:06d}" for i in range(n)) + "\n")
    return imgs


def kitti_fixture():
    import PIL
    from PIL import Image
    cv2 = sys.modules["cv2"]
    cv2.getAffineTransform = affine_from_points
    from ultralytics.data.datasets import kitti as K
    K.cv2 = cv2
    from ultralytics.data.datasets import kitti_utils as KU
    KU.cv2 = cv2
    root = tempfile.mkdtemp(prefix="y3d_kitti_")
    try:
        imgs = write_synthetic_kitti(root)
        args = R.model_args(seed=0, load_depth_maps=False, cam_dis=False, fliplr=0.5, random_crop=0.6, mixup=True)
        ds = K.KITTIDataset(os.path.join(root, "ImageSets", "train.txt"), "train", args)
        ds.resolution = np.array([224, 64])  # W * H: the algorithm does not depend on it; 1280 x 384 would be a 6 MB fixture per sample
        arrs = {"src": np.stack(imgs), "resolution": ds.resolution.copy()}
        # record the random decisions: which images were opened, whether the image was mirrored
        opened, flips = [], []
        orig_get, orig_tr = ds.get_image, Image.Image.transpose
        ds.get_image = lambda idx: (opened.append(int(idx)), orig_get(idx))[1]

        def tr(self, method):
            flips.append(int(method))
            return orig_tr(self, method)

        Image.Image.transpose = tr
        kinds = set()
        n = 0
        for seed in range(40):
            np.random.seed(seed)
            opened.clear()
            flips.clear()
            s = ds[seed % 3]
            mixed, flipped = int(s["mixed"]), int(len(flips) > 0)
            tinv = np.asarray(s["info"]["trans_inv"], np.float64)
            cropped = int(not np.allclose(tinv, affine_from_points([[112, 32], [112, -80], [224, -80]], [[160, 48], [160, -112], [320, -112]])))
            kind = (mixed, flipped, cropped)
            if kind in kinds:
                continue
            kinds.add(kind)
            partner = opened[-1] if mixed else -1
            img8 = np.round(s["img"].numpy() * 255.0).astype(np.uint8)
            assert np.array_equal(img8.astype(np.float32) / 255.0, s["img"].numpy()), "the sample is not k/255 valued"
            arrs[f"s{n}/index"] = np.array([opened[0], partner, flipped, mixed])
            arrs[f"s{n}/trans_inv"] = tinv
            arrs[f"s{n}/img8"] = img8          # (3, H, W): the reference's float tensor is exactly img8 / 255
            for k in ("bboxes", "center_2d", "center_3d", "size_2d", "depth", "calib"):
                arrs[f"s{n}/{k}"] = np.asarray(s[k], np.float64)
            n += 1
            if len(kinds) >= 8:
                break
        Image.Image.transpose = orig_tr
        arrs["n"] = np.array(n)
        arrs["versions"] = np.array([PIL.__version__, np.__version__])
        print(f"kitti_aug: {n} samples, (mixed, flipped, cropped) combinations {sorted(kinds)}; Pillow {PIL.__version__}")
        save("kitti_aug", **arrs)
    finally:
        shutil.rmtree(root, ignore_errors=True)


def main():
    R.import_reference()
    kde_fixture()
    kitti_fixture()


if __name__ == "__main__":
    main()
