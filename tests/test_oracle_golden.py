"""CPU: the oracle restatement (oracle/restate.py) against golden vectors minted from the reference
(oracle/make_golden.py).  This is what pins the oracle (task §③)."""
import copy
import os

import pytest
import torch
import yaml

from conftest import load_golden, ROOT
from oracle import restate as RS

TOL = dict(rtol=1e-4, atol=2e-5)


def close(a, b, **kw):
    t = dict(TOL)
    t.update(kw)
    torch.testing.assert_close(a.float(), b.float(), **t)


def close_rel(a, b, tol=1e-4):
    """max-norm relative: |a-b|_inf <= tol * max(|b|_inf, 1e-6) — for gradients whose scale is arbitrary."""
    a, b = a.float(), b.float()
    err = (a - b).abs().max().item()
    ref = max(b.abs().max().item(), 1e-6)
    assert err <= tol * ref, f"max err {err:.3e} vs scale {ref:.3e} (tol {tol})"


def run_module(fn, g, need_grad=True):
    st = {k: v.clone() for k, v in g["state"].items()}
    params = [k for k, v in st.items() if v.is_floating_point() and "running" not in k]
    for k in params:
        st[k].requires_grad_(True)
    x = g["x"].clone().requires_grad_(True)
    y = fn(RS.Ctx(st, True), "model.0", x)
    close(y, g["y_train"])
    (y * g["r"]).sum().backward()
    close(x.grad, g["dx"])
    for k, gv in g["grads"].items():
        close(st[k].grad, gv, rtol=2e-4, atol=1e-4)
    for k, v in g["state_after"].items():
        close(st[k].detach(), v)
    st2 = {k: v.clone() for k, v in g["state"].items()}
    st2.update({k: v.clone() for k, v in g["state_after"].items()})  # the fixture's eval pass ran after the train step
    with torch.no_grad():
        ye = fn(RS.Ctx(st2, False), "model.0", g["x"].clone())
    close(ye, g["y_eval"])


CONVS = {
    "conv_k1": dict(k=1), "conv_k3s1": dict(k=3), "conv_k3s2": dict(k=3, s=2), "conv_k3s2_odd": dict(k=3, s=2),
    "conv_stem": dict(k=3, s=2), "conv_dw3": dict(k=3, g=16), "conv_dw3s2": dict(k=3, s=2, g=16, act=False),
    "conv_dw7": dict(k=7, g=16, act=False), "conv_k1_noact": dict(k=1, act=False),
}


@pytest.mark.parametrize("name", sorted(CONVS))
def test_conv(name):
    kw = CONVS[name]
    run_module(lambda c, p, x: RS.conv_bn_act(c, p, x, **kw), load_golden(name))


def test_blocks():
    run_module(lambda c, p, x: RS.c2f(c, p, x, 2, True), load_golden("c2f_shortcut"))
    run_module(lambda c, p, x: RS.c2f(c, p, x, 1, False), load_golden("c2f_neck"))
    run_module(lambda c, p, x: RS.c2f(c, p, x, 1, True, cib_lk=True), load_golden("c2fcib_lk"))
    run_module(lambda c, p, x: RS.c2f(c, p, x, 1, True, cib_lk=False), load_golden("c2fcib"))
    run_module(lambda c, p, x: RS.scdown(c, p, x, 3, 2), load_golden("scdown"))
    run_module(lambda c, p, x: RS.sppf(c, p, x, 5), load_golden("sppf"))


@pytest.mark.parametrize("name", ["psa_1head", "psa_2head"])
def test_psa(name):
    run_module(RS.psa, load_golden(name))


def test_fold_bn():
    g = load_golden("fold_bn")
    w, b = RS.fold_conv_bn(g["w"], g["gamma"], g["beta"], g["mean"], g["var"])
    close(w, g["w_folded"])
    close(b, g["b_folded"])


@pytest.mark.parametrize("tag", ["k33", "k31"])
def test_head3d_train(tag):
    g = load_golden(f"head3d_train_{tag}")
    k1, k2, nl = [int(v) for v in g["meta"]]
    L = dict(nc=3, nl=nl, k1=k1, k2=k2)
    st = {k: v.clone() for k, v in g["state"].items()}
    for k, v in st.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    xs = [x.clone().requires_grad_(True) for x in g["x"]]
    out = RS.head3d(RS.Ctx(st, True), "model.0", xs, L, [8.0, 16.0, 32.0][:nl])
    for a, b in zip(out["one2many"], g["o2m"]):
        close(a, b)
    for a, b in zip(out["one2one"], g["o2o"]):
        close(a, b)
    for a, b in zip(out["o2m_embs"], g["o2m_embs"]):
        close(a, b)
    sum((t * r).sum() for t, r in zip(out["one2many"] + out["one2one"], g["r"])).backward()
    for a, b in zip(xs[:nl], g["dx"]):
        close(a.grad, b, rtol=2e-4, atol=1e-4)
    for k, gv in g["grads"].items():
        close(st[k].grad, gv, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("tag", ["k33", "k31"])
def test_head3d_eval(tag):
    g = load_golden(f"head3d_eval_{tag}")
    k1, k2, nl = [int(v) for v in g["meta"]]
    L = dict(nc=3, nl=nl, k1=k1, k2=k2)
    st = {k: v.clone() for k, v in g["state"].items()}
    with torch.no_grad():
        out = RS.head3d(RS.Ctx(st, False), "model.0", [x.clone() for x in g["x"]], L, [8.0, 16.0, 32.0][:nl])
    y, maps = out["one2one"]
    for a, b in zip(maps, g["maps"]):
        close(a, b)
    close(y, g["y"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("name", ["dsconv", "half", "ds_half", "pred", "pred_half"])
def test_head3d_constructor_options_train(name):
    """the non-default v10Detect3d switches (head.py:554-650, 727-737) against fixtures minted from the reference with each switch on
    (oracle/make_golden_headopts.py): head maps, 'dep' embeddings, input gradients, parameter gradients, BatchNorm running statistics"""
    g = load_golden(f"head3d_opt_{name}_train")
    ds, half, common, pred = [int(v) for v in g["meta"]]
    L = dict(nc=3, nl=2, k1=3, k2=3, dsconv=ds, pred=pred)
    st = {k: v.clone() for k, v in g["state"].items()}
    for k, v in st.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    xs = [x.clone().requires_grad_(True) for x in g["x"]]
    out = RS.head3d(RS.Ctx(st, True), "model.0", xs, L, [8.0, 16.0])
    for a, b in zip(out["one2many"] + out["one2one"], g["o2m"] + g["o2o"]):
        close(a, b)
    for a, b in zip(out["o2m_embs"] + out["o2o_embs"], g["o2m_embs"] + g["o2o_embs"]):
        close(a, b)
    sum((t * r).sum() for t, r in zip(out["one2many"] + out["one2one"], g["r"])).backward()
    for a, b in zip(xs, g["dx"]):
        close(a.grad, b, rtol=2e-4, atol=1e-4)
    for k, gv in g["grads"].items():
        close(st[k].grad, gv, rtol=2e-4, atol=2e-4)
    for k, v in g["state_after"].items():
        close(st[k].detach().float(), v.float(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["dsconv", "half", "ds_half"])
def test_head3d_constructor_options_eval(name):
    g = load_golden(f"head3d_opt_{name}_eval")
    ds, half, common, pred = [int(v) for v in g["meta"]]
    L = dict(nc=3, nl=2, k1=3, k2=3, dsconv=ds, pred=pred)
    st = {k: v.clone() for k, v in g["state"].items()}
    with torch.no_grad():
        y, maps = RS.head3d(RS.Ctx(st, False), "model.0", [x.clone() for x in g["x"]], L, [8.0, 16.0])["one2one"]
    for a, b in zip(maps, g["maps"]):
        close(a, b)
    close(y, g["y"], rtol=1e-4, atol=1e-4)


def test_head3d_predecessors_have_no_eval_path():
    """the reference's patch path raises on `use_predecessors` (channel mismatch, printed by make_golden_headopts.py); so does the oracle"""
    g = load_golden("head3d_opt_pred_train")
    L = dict(nc=3, nl=2, k1=3, k2=3, pred=1)
    with pytest.raises(RuntimeError):
        RS.head3d(RS.Ctx(dict(g["state"]), False), "model.0", [torch.zeros(1, 8, 8, 8), torch.zeros(1, 16, 8, 8)], L, [8.0, 16.0])


@pytest.mark.parametrize("topk", [8, 1])
def test_tal3d(topk):
    g = load_golden(f"tal3d_topk{topk}")
    gts = g["gt"].split((1, 4, 2, 2, 2, 3, 1, 1, 1), 2)
    targets, fg, gi = RS.tal3d(g["pd_scores"], g["pd_bboxes"], g["pd_3d"], g["anc"] * g["stride"], gts, g["mask_gt"],
                               g["stride"], g["calib"], g["mean_sizes"], topk, 3)
    # integer outputs: bit exact
    assert torch.equal(fg, g["fg_mask"].bool())
    assert torch.equal(gi, g["target_gt_idx"].long())
    assert torch.equal(targets[0], g["targets"][0].long())
    assert int(fg.sum()) > 0
    for a, b in zip(targets[1:], g["targets"][1:]):
        close(a, b, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("mode", ["box_only", "kps_only_l2", "both_l2", "both_unconstrained", "kps_only_unconstrained_top1"])
def test_tal3d_non_default_modes(mode):
    """cfg/default.yaml:116-119 (`tal_2d`, `tal_3d`, `kps_dist_metric`, `constrain_anchors`): the oracle against the reference's
    TaskAlignedAssigner3d in its other modes (tests/golden/tal3d_modes.npz, oracle/make_golden_modes.py), inputs of tal3d_topk8"""
    g = load_golden("tal3d_topk8")
    m = load_golden("tal3d_modes")[mode]
    u2, u3, l2, con, topk = [int(v) for v in m["mode"]]
    gts = g["gt"].split((1, 4, 2, 2, 2, 3, 1, 1, 1), 2)
    targets, fg, gi = RS.tal3d(g["pd_scores"], g["pd_bboxes"], g["pd_3d"], g["anc"] * g["stride"], gts, g["mask_gt"],
                               g["stride"], g["calib"], g["mean_sizes"], topk, 3, use_2d=bool(u2), use_3d=bool(u3),
                               kps_dist="l2" if l2 else "l1", constrain=bool(con))
    assert torch.equal(fg, m["fg_mask"].bool())
    assert torch.equal(gi, m["target_gt_idx"].long())
    assert torch.equal(targets[0], m["target_labels"].long())
    close(targets[1], m["target_scores"], rtol=1e-4, atol=1e-6)
    assert int(fg.sum()) > 0


def test_keypoints():
    g = load_golden("tal3d_topk8")
    gts = g["gt"].split((1, 4, 2, 2, 2, 3, 1, 1, 1), 2)
    lab = gts[0].squeeze(-1).long()
    kp = RS.keypoints_3d(gts[4], gts[6], g["mean_sizes"][lab] + gts[5], gts[7], gts[8], g["calib"])
    close(kp, g["gt_kps"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("topk", [10, 1])
def test_tal2d(topk):
    g = load_golden(f"tal2d_topk{topk}")
    gl, gb = g["gt"].split((1, 4), 2)
    tl, tb, ts, fg, gi = RS.tal2d(g["pd_scores"], g["pd_bboxes"], g["anc"] * g["stride"], gl, gb, g["mask_gt"], topk, 80)
    assert torch.equal(fg, g["fg_mask"].bool())
    assert torch.equal(gi, g["target_gt_idx"].long())
    assert torch.equal(tl, g["target_labels"].long())
    close(tb, g["target_bboxes"])
    close(ts, g["target_scores"], rtol=1e-4, atol=1e-6)


def test_loss3d():
    g = load_golden("loss3d")
    o2m = [t.clone().requires_grad_(True) for t in g["o2m"]]
    o2o = [t.clone().requires_grad_(True) for t in g["o2o"]]
    loss, items, _ = RS.loss3d({"one2many": o2m, "one2one": o2o}, g["batch"], [float(s) for s in g["strides"]], 3)
    close(loss, g["loss"].squeeze(), rtol=1e-5, atol=1e-4)
    close(items, g["items"], rtol=1e-5, atol=1e-5)
    loss.backward()
    for a, b in zip(o2m + o2o, g["g_o2m"] + g["g_o2o"]):
        close(a.grad, b, rtol=1e-4, atol=1e-6)


def test_loss2d():
    g = load_golden("loss2d")
    o2m = [t.clone().requires_grad_(True) for t in g["o2m"]]
    o2o = [t.clone().requires_grad_(True) for t in g["o2o"]]
    loss, items, _ = RS.loss2d({"one2many": o2m, "one2one": o2o}, g["batch"], [float(s) for s in g["strides"]], 80)
    close(loss, g["loss"].squeeze(), rtol=1e-5, atol=1e-4)
    close(items, g["items"], rtol=1e-5, atol=1e-5)
    loss.backward()
    for a, b in zip(o2m + o2o, g["g_o2m"] + g["g_o2o"]):
        close(a.grad, b, rtol=1e-4, atol=1e-6)


def test_postprocess():
    g = load_golden("post3d")
    reg, sc, lab = RS.postprocess3d(g["preds"], 50, 3)
    assert torch.equal(lab, g["labels"].long())
    close(reg, g["reg"])
    close(sc, g["scores"])
    g = load_golden("post2d")
    bx, sc, lab = RS.postprocess2d(g["preds"], 300, 80)
    assert torch.equal(lab, g["labels"].long())
    close(bx, g["boxes"])
    close(sc, g["scores"])


def test_kitti_decode():
    """decode_preds (kitti.py:519-576): with the inverse affine, without it (fixed 1242/1280, 375/384 rescale), camera-distance mode"""
    g = load_golden("kitti_decode")
    for tag, inv, cam in (("aug", g["inv_trans"], False), ("noaug", None, False), ("camdis", g["inv_trans"], True)):
        rows, keep = RS.kitti_decode(g["preds"], g["calib"], g["ratio"], inv, use_camera_dis=cam)
        cnt = g[f"count_{tag}"].long()
        assert keep.sum(1).tolist() == cnt.tolist(), "detections under the 0.001 score threshold are dropped"
        assert 0 < int(cnt.min()) and int(cnt.max()) < rows.shape[1]
        for i in range(rows.shape[0]):
            mine = torch.from_numpy(rows[i][keep[i]])
            ref = g[f"rows_{tag}"][i, :int(cnt[i])].double()
            torch.testing.assert_close(mine, ref, rtol=1e-5, atol=1e-5)  # the fixture is float32-accurate (numpy 2 scalar rules)


def tiny_cfg(name, **over):
    with open(os.path.join(ROOT, "yolov10-3d_amd", "cfg", "models", name)) as f:
        d = yaml.safe_load(f)
    d.update(over)
    return d


TINY = dict(scales={"n": [0.33, 0.125, 1024]}, scale="n",
            channels={k + "_c": 16 for k in ("cls", "o2d", "s2d", "o3d", "s3d", "hd", "dep", "dep_un")})


@pytest.mark.parametrize("tag,over", [("e2e_tiny3d_s", dict(kernel_size_1=3, kernel_size_2=3, num_scales=3)),
                                      ("e2e_tiny3d_m", dict(kernel_size_1=3, kernel_size_2=1, num_scales=2))])
def test_e2e_tiny3d(tag, over):
    g = load_golden(tag)
    cfg = tiny_cfg("v10-3D/yolov10s_3D.yaml" if tag.endswith("_s") else "v10-3D/yolov10m_3D.yaml", **TINY, **over)
    spec = RS.build_spec(cfg)
    strides = RS.model_strides(spec)
    assert strides == [float(s) for s in g["strides"]]
    st = {k: v.clone() for k, v in g["state"].items()}
    ref_keys = set(st)
    mine = set(RS.init_state(spec).keys())
    assert mine == ref_keys, (sorted(mine - ref_keys)[:5], sorted(ref_keys - mine)[:5])
    for k, v in st.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    preds = RS.forward(spec, st, g["img"], True)
    loss, items, aux = RS.loss3d(preds, g["batch"], strides, 3)
    close(loss, g["loss"].squeeze(), rtol=2e-5, atol=1e-3)
    close(items, g["items"], rtol=1e-4, atol=1e-5)
    loss.backward()
    for k, gv in g["grads"].items():
        close_rel(st[k].grad, gv, 2e-4)
    for k, v in g["state_after"].items():
        close(st[k].detach(), v)
    st2 = {k: v.clone() for k, v in g["state"].items()}
    st2.update({k: v.clone() for k, v in g["state_eval"].items()})  # running stats the fixture's eval pass saw
    with torch.no_grad():
        y = RS.forward(spec, st2, g["img_eval"], False)["one2one"][0]
    close(y, g["y_eval"], rtol=1e-3, atol=1e-3)
    reg, sc, lab = RS.postprocess3d(y.permute(0, 2, 1), 50, 3)
    assert torch.equal(lab, g["post_labels"].long())
    close(sc, g["post_scores"], rtol=1e-3, atol=1e-4)


def test_e2e_tiny2d():
    g = load_golden("e2e_tiny2d")
    cfg = tiny_cfg("v10/yolov10n.yaml", nc=20, scales={"n": [0.33, 0.125, 1024]}, scale="n")
    spec = RS.build_spec(cfg)
    strides = RS.model_strides(spec)
    st = {k: v.clone() for k, v in g["state"].items()}
    mine = set(RS.init_state(spec).keys())
    assert mine == set(st), (sorted(mine - set(st))[:5], sorted(set(st) - mine)[:5])
    for k, v in st.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    preds = RS.forward(spec, st, g["img"], True)
    loss, items, _ = RS.loss2d(preds, g["batch"], strides, 20)
    close(loss, g["loss"].squeeze(), rtol=2e-5, atol=1e-3)
    close(items, g["items"], rtol=1e-4, atol=1e-5)
    loss.backward()
    for k, gv in g["grads"].items():
        close_rel(st[k].grad, gv, 2e-4)
    st2 = {k: v.clone() for k, v in g["state"].items()}
    st2.update({k: v.clone() for k, v in g["state_after"].items()})
    with torch.no_grad():
        out = RS.forward(spec, st2, g["img"], False)
    close(out["one2one"][0], g["y_eval_o2o"], rtol=1e-3, atol=1e-3)
    close(out["one2many"][0], g["y_eval_o2m"], rtol=1e-3, atol=1e-3)


def test_e2e_n2d_320_baseline_config0():
    """BASELINE.json configs[0]: the shipped YOLOv10-N 2D yaml (nc=80), 320x320, batch 2 — one training step + eval + postprocess of
    the reference's CPU path (oracle/make_golden_configs.py) against the restatement"""
    g = load_golden("e2e_n2d_320")
    import yaml
    with open(os.path.join(ROOT, "yolov10-3d_amd", "cfg", "models", "v10", "yolov10n.yaml")) as f:
        cfg = yaml.safe_load(f)
    cfg["scale"] = "n"
    spec = RS.build_spec(cfg)
    strides = RS.model_strides(spec)
    assert [float(s) for s in strides] == [float(s) for s in g["strides"]]
    st = {k: v.clone() for k, v in g["state"].items()}
    assert set(RS.init_state(spec).keys()) == set(st)
    for k, v in st.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    img = g["img8"].float() / 255
    preds = RS.forward(spec, st, img, True)
    loss, items, _ = RS.loss2d(preds, g["batch"], strides, 80)
    close(loss, g["loss"].squeeze(), rtol=2e-5, atol=1e-3)
    close(items, g["items"], rtol=1e-4, atol=1e-5)
    loss.backward()
    for k, gv in g["grads"].items():
        close_rel(st[k].grad, gv, 5e-4)
    big = max(float(v) for v in g["grad_norms"].values())
    for k, nv in g["grad_norms"].items():  # every parameter's gradient norm (exact-zero-in-theory ones on the scale of the largest)
        assert abs(float(st[k].grad.norm()) - float(nv)) <= 1e-3 * max(float(nv), 1e-3 * big), k
    st2 = {k: v.clone() for k, v in g["state"].items()}
    st2.update({k: v.clone() for k, v in g["state_after"].items()})
    with torch.no_grad():
        out = RS.forward(spec, st2, img, False)
    close(out["one2one"][0], g["y_eval_o2o"], rtol=1e-3, atol=1e-3)
    close(out["one2many"][0], g["y_eval_o2m"], rtol=1e-3, atol=1e-3)
    bx, sc, lab = RS.postprocess2d(g["y_eval_o2o"].permute(0, 2, 1), 300, 80)
    assert torch.equal(lab.long(), g["post_labels"].long())
    close(sc, g["post_scores"], rtol=1e-6, atol=1e-7)
    close(bx, g["post_boxes"], rtol=1e-6, atol=1e-5)


def test_kde_depth_fusion_restatement_matches_reference():
    """f2: oracle restatement of the validator's one-to-many depth fusion against the reference's own output (bit for bit)"""
    import numpy as np
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "kde_fusion.npz"))  # (carries a string array: not through load_golden)
    O, M, F = (torch.from_numpy(g[k]) for k in ("predsO", "predsM", "fused"))
    out = RS.kde_fuse_depth(O, M)
    assert int((F[..., -4] != O[..., -4]).sum()) >= 50, "the fixture must exercise the fusion"
    assert torch.equal(out, F)


def test_kitti_image_aug_restatement_matches_reference():
    """f3: mirror / mixup blend / affine crop of KITTIDataset.__getitem__ (Pillow arithmetic) restated in numpy, against the
    reference's own samples: every pixel of every sample equal; plus the crop matrix from (centre, crop size)"""
    import numpy as np
    import os
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "kitti_aug.npz"))
    src, (W, H) = z["src"], z["resolution"]
    kinds = set()
    for i in range(int(z["n"])):
        i0, i1, flip, mixed = [int(v) for v in z[f"s{i}/index"]]
        out = RS.kitti_image_aug(src[i0], src[i1] if mixed else None, flip, z[f"s{i}/trans_inv"], (W, H))
        assert np.array_equal(out, z[f"s{i}/img8"].transpose(1, 2, 0)), f"sample {i}"
        kinds.add((flip, mixed))
    assert len(kinds) == 4
    # un-cropped samples: centre = image centre, crop = image size
    t, tinv = RS.get_affine_transform(np.array([160.0, 48.0]), np.array([320.0, 96.0]), (W, H), inv=True)
    assert np.allclose(tinv, z["s0/trans_inv"], atol=1e-9) and np.allclose(t @ np.vstack((tinv, [0, 0, 1])), np.eye(3)[:2], atol=1e-9)


def test_fp8_conv_mode_restatement_invariants():
    """the oracle's restatement of the fp8 MFMA convolution mode (no reference counterpart: BASELINE configs[4] is this build's own format) -
    its defining properties, on the CPU: (1) an MX block scale is the smallest power of two that brings the block's amax to <= 448, so no
    code saturates and the dequantised value is within half an e4m3 step (2^-4 relative) of the input, blocks of zeros stay zeros;
    (2) values that are e4m3-representable at their block's scale come back exactly; (3) the convolution's gradients are those of the
    UNquantised convolution (straight-through): dx sees w, dW sees x, not their quantised forms."""
    torch.manual_seed(0)
    x = torch.randn(2, 64, 5, 7) * torch.exp2(torch.randint(-6, 6, (2, 2, 1, 5, 7)).float()).repeat_interleave(32, 2).reshape(2, 64, 5, 7)
    x[0, :32, 0, 0] = 0
    q, s, xe = RS.mx_quantize_act(x)
    blk = x.reshape(2, 2, 32, 5, 7)
    amax = blk.abs().amax(2)
    scale = torch.exp2(s.float() - 127)
    assert bool((amax <= 448 * scale).all()) and bool(((amax > 224 * scale) | (amax == 0)).all()), "not the smallest power of two"
    assert int(s[0, 0, 0, 0]) == 127 and bool((xe[0, :32, 0, 0] == 0).all())
    err = (xe - x).abs().reshape(2, 2, 32, 5, 7)
    assert bool((err <= 2.0 ** -4 * blk.abs() + 2.0 ** -10 * scale.unsqueeze(2)).all())  # half a step: 3 mantissa bits; subnormals: 2^-9 * scale steps
    exact = torch.randint(-7, 8, (1, 32, 3, 3)).float() * 0.25
    assert torch.equal(RS.mx_quantize_act(exact)[2], exact)
    # straight-through gradients
    w = (torch.randn(16, 128, 3, 3) * 0.1).requires_grad_(True)
    xin = torch.randn(1, 128, 8, 8).requires_grad_(True)
    y = RS._Fp8ConvSTE.apply(xin, w, 1)
    dy = torch.randn_like(y)
    y.backward(dy)
    assert torch.allclose(xin.grad, torch.nn.grad.conv2d_input(xin.shape, w.detach(), dy, 1, 1, 1, 1), atol=1e-5)
    assert torch.allclose(w.grad, torch.nn.grad.conv2d_weight(xin.detach(), w.shape, dy, 1, 1, 1, 1), atol=1e-4)
    assert not torch.allclose(y, torch.nn.functional.conv2d(xin, w, None, 1, 1), atol=1e-4), "the forward must see the quantised input"
    assert RS.fp8_conv_served(torch.zeros(1, 128, 8, 8), w, 1, 1, 1) and not RS.fp8_conv_served(torch.zeros(1, 128, 8, 8), w, 2, 1, 1)
