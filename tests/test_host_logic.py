"""CPU: host-side logic of the package (yaml -> model, state_dict layout) and the test-only torch formulations of the assigners
(tests/torch_assigners.py, the comparators of the HIP kernels at bench scale) against the reference's golden vectors."""
import pytest
import torch

from conftest import load_golden

import yolov10_3d_amd as y3d
from yolov10_3d_amd import loss as PL
from oracle import restate as RS
import torch_assigners as TA


def close(a, b, rtol=1e-4, atol=2e-5):
    torch.testing.assert_close(a.float(), b.float(), rtol=rtol, atol=atol)


@pytest.mark.parametrize("name,params_m,nsd", [("yolov10s_3D.yaml", 30.07, 1386), ("yolov10m_3D.yaml", 20.38, None),
                                               ("yolov10n.yaml", 2.78, None), ("yolov10x_3D.yaml", 63.89, None)])
def test_model_tables_build(name, params_m, nsd):
    m = y3d.DetectionModel(name)
    n = sum(p.numel() for p in m.parameters()) / 1e6
    assert abs(n - params_m) < 0.01, n  # SURVEY §8d measured parameter counts
    if nsd:
        assert len(m.state_dict()) == nsd  # incl. the aliased head keys


def test_state_dict_keys_match_reference_fixture():
    g = load_golden("e2e_tiny3d_s")
    cfg = y3d.yaml_model_load("yolov10s_3D.yaml")
    cfg.update(scales={"n": [0.33, 0.125, 1024]}, scale="n",
               channels={k + "_c": 16 for k in ("cls", "o2d", "s2d", "o3d", "s3d", "hd", "dep", "dep_un")})
    m = y3d.YOLOv10_3DDetectionModel(cfg)
    sd = m.state_dict()
    for k, v in g["state"].items():
        assert k in sd and sd[k].shape == v.shape, k
    assert [float(s) for s in m.stride] == [float(s) for s in g["strides"]]
    # aliases are the same tensors (reference head.py:627-629)
    assert sd["model.23.cls.0.0.conv.weight"].data_ptr() == sd["model.23.o2o_heads.0.0.0.conv.weight"].data_ptr()


@pytest.mark.parametrize("topk", [8, 1])
def test_assigner3d_cpu(topk):
    g = load_golden(f"tal3d_topk{topk}")
    gts = g["gt"].split((1, 4, 2, 2, 2, 3, 1, 1, 1), 2)
    asg = TA.TaskAlignedAssigner3d(topk=topk, num_classes=3, alpha=0.5, beta=1.0, gamma=1.0)
    targets, fg, gi, pk, gk = asg(g["pd_scores"], g["pd_bboxes"], g["pd_3d"], g["anc"] * g["stride"], gts, g["mask_gt"], g["stride"],
                                  g["calib"], g["mean_sizes"])
    assert torch.equal(fg, g["fg_mask"].bool()) and torch.equal(gi, g["target_gt_idx"].long())
    for a, b in zip(targets[1:], g["targets"][1:]):
        close(a, b, atol=1e-6)
    close(pk, g["pd_kps"], atol=1e-4)


@pytest.mark.parametrize("topk", [10, 1])
def test_assigner2d_cpu(topk):
    g = load_golden(f"tal2d_topk{topk}")
    gl, gb = g["gt"].split((1, 4), 2)
    asg = TA.TaskAlignedAssigner(topk=topk, num_classes=80, alpha=0.5, beta=6.0)
    tl, tb, ts, fg, gi = asg(g["pd_scores"], g["pd_bboxes"], g["anc"] * g["stride"], gl, gb, g["mask_gt"])
    assert torch.equal(fg, g["fg_mask"].bool()) and torch.equal(gi, g["target_gt_idx"].long())
    close(ts, g["target_scores"], atol=1e-6)


class _FakeModel:
    def __init__(self, head, args):
        self.model = [head]
        self.args = args


def test_loss3d_refuses_cpu_tensors():
    """the 3D loss is a HIP kernel path: CPU tensors must raise, never fall back"""
    from types import SimpleNamespace
    g = load_golden("loss3d")
    head = SimpleNamespace(stride=g["strides"], nc=3, no=38)
    crit = PL.DetectLoss3d(_FakeModel(head, SimpleNamespace(**y3d.tasks.DEFAULT_HYP)))
    with pytest.raises(y3d.Y3DError):
        crit({"one2many": g["o2m"], "one2one": g["o2o"]}, g["batch"])


def test_postprocess_refuses_cpu_tensors():
    g = load_golden("post3d")
    with pytest.raises(y3d.Y3DError):
        PL.v10_3Dpostprocess(g["preds"], 50, 3)


def test_kitti_decode_and_optimizer_side_refuse_cpu_tensors():
    """the eval tail and the optimizer-side kernels are HIP paths too: no CPU fallback"""
    from yolov10_3d_amd import kitti, optim
    g = load_golden("kitti_decode")
    with pytest.raises(y3d.Y3DError):
        kitti.decode_preds_eval(g["preds"], g["calib"], ["a", "b", "c"], g["ratio"], g["inv_trans"])
    with pytest.raises(y3d.Y3DError):
        kitti.decode_preds_device(torch.zeros(2, 5, 36), g["calib"][:2], g["ratio"][:2], None, undo_augment=False)  # not 37 columns
    lin = torch.nn.Linear(4, 4)
    lin.weight.grad, lin.bias.grad = torch.zeros_like(lin.weight), torch.zeros_like(lin.bias)
    for opt in (optim.FusedSGD(list(lin.parameters())), optim.FusedAdamW(list(lin.parameters()))):
        with pytest.raises(y3d.Y3DError):
            opt.step()
    with pytest.raises(y3d.Y3DError):
        optim.ModelEMA(lin).update(lin)


def test_build_optimizer_groups_follow_reference():
    """engine/trainer.py:766-790: biases (no decay) | weights (decay) | normalisation weights (no decay); SGD nesterov or AdamW"""
    from yolov10_3d_amd import optim
    m = y3d.YOLOv10_3DDetectionModel("yolov10n_3D.yaml")
    for name, cls in (("SGD", optim.FusedSGD), ("AdamW", optim.FusedAdamW)):
        o = optim.build_optimizer(m, lr=0.01, momentum=0.9, decay=5e-4, name=name)
        assert type(o) is cls and [g["weight_decay"] for g in o.param_groups] == [0.0, 5e-4, 0.0]
        assert all(p.dim() == 1 for p in o.param_groups[0]["params"]) and all(p.dim() == 4 for p in o.param_groups[1]["params"])
        assert sum(len(g["params"]) for g in o.param_groups) == len({id(p) for p in m.parameters()})
    with pytest.raises(NotImplementedError):
        optim.build_optimizer(m, name="RMSProp")


def test_no_positive_zero_ties_in_fixtures():
    """The top-k tie rule (lowest index first) only differs from the reference's library-dependent order when an
    in-box anchor with a metric of exactly 0 is selected; the golden fixtures contain no such case."""
    for topk in (8, 1):
        g = load_golden(f"tal3d_topk{topk}")
        gts = g["gt"].split((1, 4, 2, 2, 2, 3, 1, 1, 1), 2)
        targets, fg, gi = RS.tal3d(g["pd_scores"], g["pd_bboxes"], g["pd_3d"], g["anc"] * g["stride"], gts, g["mask_gt"], g["stride"],
                                   g["calib"], g["mean_sizes"], topk, 3)
        assert (targets[1].sum(-1)[fg] > 0).all()


def test_fold_conv_bn_matches_reference_fixture():
    g = load_golden("fold_bn")
    w, b = y3d.tasks.fuse_conv_and_bn(g["w"], g["gamma"], g["beta"], g["mean"], g["var"])
    close(w, g["w_folded"])
    close(b, g["b_folded"])


# ---------------------------------------------------------------------------------------------------------
# initialisation constants against the reference-minted fixture (oracle/make_golden_init.py)
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nl", [3, 2])
def test_bias_init_3d_matches_reference_fixture(nl):
    from yolov10_3d_amd import modules as M
    g = load_golden("bias_init")[f"head3d_nl{nl}"]
    chan = {k + "_c": 16 for k in ("cls", "o2d", "s2d", "o3d", "s3d", "hd", "dep", "dep_un")}
    torch.manual_seed(0)
    hd = M.v10Detect3d(3, (16, 32, 64), False, chan, False, False, False, False, nl, False, False, 3, 3)
    hd.stride = torch.tensor([8.0, 16.0, 32.0][:nl])
    hd.bias_init()
    for name in ("cls", "o2d", "s2d", "o3d", "s3d", "dep"):
        for i in range(nl):
            assert torch.equal(getattr(hd, name)[i][-1].bias.detach(), g[f"{name}/{i}/bias"]), f"{name}[{i}] bias"
    deps, ranges = M.v10Detect3d.DEPTH_PRIOR[nl]
    for i in range(nl):
        # re-drawn projection weights: same distribution as the reference's draw (its min / max / mean / std are in the fixture)
        w = hd.dep[i][-1].weight.detach()
        lo, hi = ranges[i]
        ref = g[f"dep/{i}/weight_stats"]
        assert lo <= float(w.min()) and float(w.max()) <= hi and lo <= float(ref[0]) and float(ref[1]) <= hi
        assert 0.6 < float(w.std()) / float(ref[3]) < 1.6
        w = hd.s3d[i][-1].weight.detach()
        ref = g[f"s3d/{i}/weight_stats"]
        assert 0.6 < float(w.std()) / float(ref[3]) < 1.6 and abs(float(w.mean())) < 0.03
    # the one-to-one set aliases the named branches; the one-to-many set restarts as its copy (head.py:869-870)
    assert int(g["o2m_equals_o2o"]) == 1
    assert hd.o2o_heads[0] is hd.cls and hd.o2o_heads[6] is hd.dep
    for a, b in zip(hd.o2o_heads.state_dict().values(), hd.o2m_heads.state_dict().values()):
        assert torch.equal(a, b)
    assert hd.o2m_heads[0][0][0].conv.weight.data_ptr() != hd.cls[0][0].conv.weight.data_ptr()


def test_bias_init_2d_and_initialize_weights_match_reference_fixture():
    from yolov10_3d_amd import modules as M
    g = load_golden("bias_init")
    torch.manual_seed(0)
    h2 = M.v10Detect(80, (16, 32, 64))
    h2.stride = torch.tensor([8.0, 16.0, 32.0])
    h2.bias_init()
    for i in range(3):
        for name in ("cv2", "cv3", "one2one_cv2", "one2one_cv3"):
            assert torch.equal(getattr(h2, name)[i][-1].bias.detach(), g["head2d"][f"{name}/{i}/bias"]), f"{name}[{i}]"
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8), torch.nn.SiLU())
    y3d.tasks.initialize_weights(m)
    eps, mom = [float(v) for v in g["initialize_weights"]["bn_eps_momentum"]]
    assert (m[1].eps, m[1].momentum) == (eps, mom) and int(m[2].inplace) == int(g["initialize_weights"]["silu_inplace"])
    # a built model carries them on every BatchNorm (tasks.DetectionModel.__init__ calls initialize_weights)
    cfg = y3d.yaml_model_load("yolov10s_3D.yaml")
    cfg.update(scales={"n": [0.33, 0.125, 1024]}, scale="n",
               channels={k + "_c": 16 for k in ("cls", "o2d", "s2d", "o3d", "s3d", "hd", "dep", "dep_un")})
    model = y3d.YOLOv10_3DDetectionModel(cfg)
    bns = [b for b in model.modules() if isinstance(b, torch.nn.BatchNorm2d)]
    assert bns and all((b.eps, b.momentum) == (eps, mom) for b in bns)
    # ... and its head went through bias_init with the model's strides (tasks.py: DetectionModel.__init__)
    g3 = g["head3d_nl3"]
    head = model.model[-1]
    for i in range(3):
        assert torch.equal(head.cls[i][-1].bias.detach()[:3], g3[f"cls/{i}/bias"]) and torch.equal(head.dep[i][-1].bias.detach(), g3[f"dep/{i}/bias"])


# ---------------------------------------------------------------------------------------------------------
# drop-in boundary (INTEGRATION.md form A)
# ---------------------------------------------------------------------------------------------------------
def _reference_walk(model, x):
    """the reference's BaseModel._predict_once (nn/tasks.py:117-146) as the trainer would run it over rebound modules: a plain walk of
    the rows, no placement, no private state"""
    y = []
    for m in model.model:
        if m.f != -1:
            x = y[m.f] if isinstance(m.f, int) else [x if j == -1 else y[j] for j in m.f]
        x = m(x)
        y.append(x if m.i in model.save else None)
    return x


@pytest.mark.parametrize("name,want", [("yolov10s_3D.yaml", [8.0, 16.0, 32.0]), ("yolov10m_3D.yaml", [8.0, 16.0]), ("yolov10n.yaml", [8.0, 16.0, 32.0]),
                                       ("yolov10x.yaml", [8.0, 16.0, 32.0])])
def test_host_tensor_is_answered_with_shapes_only(name, want):
    """the reference's constructor probes the strides with forward(torch.zeros(1, ch, 256, 256)) on the HOST (nn/tasks.py:300-310):
    the modules answer a host tensor with meta tensors of the right shape, so the probe's strides equal the table's - and nothing
    is computed"""
    m = y3d.DetectionModel(name)
    out = _reference_walk(m.train(), torch.zeros(1, 3, 256, 256))
    maps = out["one2many"]
    assert all(t.device.type == "meta" for t in maps)
    assert [256 / t.shape[-2] for t in maps] == want == [float(s) for s in m.stride]
    head = m.model[-1]
    assert all(t.shape[1] == head.no for t in maps)
    ev = _reference_walk(m.eval(), torch.zeros(2, 3, 256, 256))
    y = ev["one2one"][0]
    A = sum(t.shape[2] * t.shape[3] for t in maps)
    assert y.device.type == "meta" and tuple(y.shape) == (2, head.no if hasattr(head, "dep") else 4 + head.nc, A)


def test_dropin_against_the_reference():
    """INTEGRATION.md form A applied verbatim to the imported reference: all twelve shipped yamls build through the reference's own
    constructors with identical keys / strides / bias_init / folded weights (oracle/check_dropin.py; build container only - the
    reference does not travel).  Run in a child process: the block rebinds names inside the reference AND torch.nn.Upsample."""
    import json
    import os
    import subprocess
    import sys
    from oracle import ref_shim
    if not ref_shim.available():
        pytest.skip("the reference is not present on this machine")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "oracle.check_dropin"], cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    last = json.loads(r.stdout.strip().splitlines()[-1])
    assert last == {"yamls": 12, "failed": 0}


def test_bench_crash_guard_prints_the_measured_line(tmp_path):
    """tools/crash_line.c (bench.py, N > 1 hipGraph leg): a fatal signal after arming prints the preformatted line from the handler and
    exits with the given code; disarmed, the default action is back"""
    import os
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "libcrash_line.so")
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-Wall", "-Werror", os.path.join(root, "tools", "crash_line.c"), "-o", so], check=True)
    prog = textwrap.dedent(f"""
        import ctypes, sys
        L = ctypes.CDLL({so!r})
        assert L.crash_line_arm(b'{{"metric": "m", "value": 1.5}}', 0) == 0
        if sys.argv[1] == "disarm":
            L.crash_line_disarm()
        sys.stdout.flush()
        ctypes.string_at(0)
    """)
    r = subprocess.run([sys.executable, "-c", prog, "armed"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout == '{"metric": "m", "value": 1.5}\n'
    r = subprocess.run([sys.executable, "-c", prog, "disarm"], capture_output=True, text=True)
    assert r.returncode != 0 and r.stdout == ""
