"""GPU parity: the HIP modules (through the C ABI) against the golden vectors minted from the reference,
and against the CPU oracle restatement on fresh seeded inputs.

fp32 mode (exact-f32 MFMA) is held to 1e-3 relative on every output (north_star tolerance);
bf16 mode (the performance mode) is held to a documented looser bound."""
import copy

import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

import yolov10_3d_amd as y3d  # noqa: E402
from yolov10_3d_amd import modules as M  # noqa: E402
from oracle import restate as RS  # noqa: E402  (the checker)

DEV = "cuda"


def rel_err(a, b, floor=1e-4):
    """max-norm relative error |a-b|_inf / max(|b|_inf, floor)."""
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp(min=floor)).item()


def check(a, b, tol, what="", floor=1e-4):
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    e = rel_err(a, b, floor)
    assert e <= tol, f"{what}: max-norm relative error {e:.3e} > {tol}"


def grad_floor(grads, tol=1e-3):
    """gradients that are numerically zero in exact arithmetic (a BN bias cancelled by a later BatchNorm) are compared on
    the scale of the fixture's largest gradient, not on their own rounding noise"""
    return max(1e-3, tol / 3) * max(float(v.abs().max()) for v in grads.values())


def l2_rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp(min=1e-12))


def cosine(a, b):
    a, b = a.detach().float().cpu().flatten(), b.detach().float().cpu().flatten()
    return float(torch.dot(a, b) / (a.norm() * b.norm()).clamp(min=1e-12))


def check_sparse_eval(y, y_ref, nc, tol, min_overlap=0.9):
    """eval-mode y (B, no, A): cls channels are dense; the 35 regression channels exist only at the top-50 candidate cells per
    level.  Candidate selection is a top-k over nearly tied logits at random init, so a 1e-6 difference may swap a few cells:
    compare cls everywhere, require >= 90 % common candidates and compare the regression channels on the common ones."""
    y, y_ref = y.detach().float().cpu(), y_ref.detach().float().cpu()
    check(y[:, :nc], y_ref[:, :nc], tol, "eval cls")
    dep = nc + 4 + 2 + 3 + 24  # raw depth channel: non-zero exactly at candidate cells (bias 10..45)
    ca, cb = y[:, dep] != 0, y_ref[:, dep] != 0
    both = ca & cb
    assert both.sum() >= min_overlap * cb.sum(), f"candidate overlap {int(both.sum())}/{int(cb.sum())}"
    ya, yb = y.permute(0, 2, 1)[both], y_ref.permute(0, 2, 1)[both]
    check(ya, yb, tol, "eval reg @ common candidates")
    return bool((ca == cb).all())


def load_into(mod, state, prefix="model.0."):
    sd = {k[len(prefix):]: v for k, v in state.items() if k.startswith(prefix)}
    missing, unexpected = mod.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("num_batches" in k or k.split(".")[0] in ("cls", "o2d", "s2d", "o3d", "s3d", "hd", "dep", "dep_un", "o2m_heads") for k in missing), missing
    for m in mod.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eps, m.momentum = 1e-3, 0.03
    return mod.to(DEV)


MODS = {
    "conv_k1": lambda: M.Conv(16, 24, 1, 1), "conv_k3s1": lambda: M.Conv(16, 24, 3, 1), "conv_k3s2": lambda: M.Conv(16, 24, 3, 2),
    "conv_k3s2_odd": lambda: M.Conv(8, 16, 3, 2), "conv_stem": lambda: M.Conv(3, 16, 3, 2),
    "conv_dw3": lambda: M.Conv(16, 16, 3, 1, None, 16), "conv_dw3s2": lambda: M.Conv(16, 16, 3, 2, None, 16, 1, False),
    "conv_dw7": lambda: M.Conv(16, 16, 7, 1, 3, 16, 1, False), "conv_k1_noact": lambda: M.Conv(16, 24, 1, 1, None, 1, 1, False),
    "c2f_shortcut": lambda: M.C2f(32, 32, 2, True), "c2f_neck": lambda: M.C2f(48, 32, 1, False),
    "c2fcib_lk": lambda: M.C2fCIB(32, 32, 1, True, True), "c2fcib": lambda: M.C2fCIB(32, 32, 1, True, False),
    "scdown": lambda: M.SCDown(16, 32, 3, 2), "sppf": lambda: M.SPPF(32, 32, 5),
    "psa_1head": lambda: M.PSA(128, 128), "psa_2head": lambda: M.PSA(256, 256),
}


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 6e-2)])
@pytest.mark.parametrize("name", sorted(MODS))
def test_module_vs_reference_golden(name, dtype, tol):
    y3d.set_compute_dtype(dtype)
    g = load_golden(name)
    mod = load_into(MODS[name](), g["state"]).train()
    is_stem = name == "conv_stem"
    x = g["x"].to(DEV).requires_grad_(not is_stem)
    y = mod(x)
    check(y, g["y_train"], tol, "y_train")
    (y.float() * g["r"].to(DEV)).sum().backward()
    gtol = tol * 3
    if not is_stem:
        check(x.grad, g["dx"], gtol, "dx")
    named = dict(mod.named_parameters())
    gf = grad_floor(g["grads"], tol)
    for k, gv in g["grads"].items():
        check(named[k[len("model.0."):]].grad, gv, gtol, f"grad {k}", gf)
    sd = mod.state_dict()
    for k, v in g["state_after"].items():
        check(sd[k[len("model.0."):]].float(), v.float(), 1e-3 if dtype == torch.float32 else 3e-2, f"state {k}")
    mod.eval()
    with torch.no_grad():
        ye = mod(g["x"].to(DEV))
    check(ye, g["y_eval"], tol, "y_eval")


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 6e-2)])
@pytest.mark.parametrize("tag", ["k33", "k31"])
def test_head3d_train_vs_reference_golden(tag, dtype, tol):
    y3d.set_compute_dtype(dtype)
    g = load_golden(f"head3d_train_{tag}")
    k1, k2, nl = [int(v) for v in g["meta"]]
    chan = {k + "_c": 16 for k in ("cls", "o2d", "s2d", "o3d", "s3d", "hd", "dep", "dep_un")}
    hd = M.v10Detect3d(3, (16, 32, 64), False, chan, False, False, False, False, nl, False, False, k1, k2)
    hd.stride = torch.tensor([8.0, 16.0, 32.0][:nl])
    hd = load_into(hd, g["state"]).train()
    xs = [x.to(DEV).requires_grad_(True) for x in g["x"]]
    out = hd(xs)
    for a, b in zip(out["one2many"] + out["one2one"], g["o2m"] + g["o2o"]):
        check(a, b, tol, "head map")
    for a, b in zip(out["o2m_embs"], g["o2m_embs"]):
        check(a, b, tol, "embs")
    sum((t.float() * r.to(DEV)).sum() for t, r in zip(out["one2many"] + out["one2one"], g["r"])).backward()
    for a, b in zip(xs[:nl], g["dx"]):
        check(a.grad, b, tol * 3, "dx")
    named = dict(hd.named_parameters())
    for k, gv in g["grads"].items():
        check(named[k[len("model.0."):]].grad, gv, tol * 3, f"grad {k}")


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 2e-2)])
def test_head3d_bn_projection_fusion_matches_unfused(dtype, tol):
    """64-channel branches: the fused node (grouped conv -> BatchNorm statistics -> projections applying BatchNorm + SiLU on the fly,
    one reduce + one apply pass backward) against the unfused chain (bn_act_fwd, projections, projection data/weight gradients,
    BatchNorm reduce/apply) on the same weights: head maps, input gradients and every parameter gradient"""
    y3d.set_compute_dtype(dtype)
    torch.manual_seed(7)
    chan = {k + "_c": 64 for k in ("cls", "o2d", "s2d", "o3d", "s3d", "hd", "dep", "dep_un")}
    hd = M.v10Detect3d(3, (32, 64), False, chan, False, False, False, False, 2, False, False, 3, 3)
    hd.stride = torch.tensor([8.0, 16.0])
    hd = hd.to(DEV).train()
    with torch.no_grad():
        for p_ in hd.parameters():
            p_.add_(0.05 * torch.randn_like(p_))
    xs0 = [torch.randn(3, 32, 24, 24, device=DEV), torch.randn(3, 64, 12, 12, device=DEV)]
    rs = None
    res = {}
    for fused in (True, False):
        hd.fuse_bn_proj = fused
        hd.zero_grad(set_to_none=True)
        xs = [x.clone().requires_grad_(True) for x in xs0]
        out = hd(xs)
        maps = out["one2many"] + out["one2one"]
        if rs is None:
            rs = [torch.randn_like(t.float()) for t in maps]
        sum((t.float() * r).sum() for t, r in zip(maps, rs)).backward()
        res[fused] = ([t.detach().float() for t in maps], [x.grad.float() for x in xs], {k: v.grad.float().clone() for k, v in hd.named_parameters() if v.grad is not None})
    for a, b in zip(res[True][0], res[False][0]):
        check(a, b, tol, "head map fused vs unfused")
    for a, b in zip(res[True][1], res[False][1]):
        check(a, b, tol * 3, "dx fused vs unfused")
    assert set(res[True][2]) == set(res[False][2])
    gf = 1e-3 * max(float(v.abs().max()) for v in res[False][2].values())
    for k in res[False][2]:
        check(res[True][2][k], res[False][2][k], tol * 3, f"grad {k}", gf)


@pytest.mark.parametrize("tag", ["k33", "k31"])
def test_head3d_eval_vs_reference_golden(tag):
    y3d.set_compute_dtype(torch.float32)
    g = load_golden(f"head3d_eval_{tag}")
    k1, k2, nl = [int(v) for v in g["meta"]]
    chan = {k + "_c": 16 for k in ("cls", "o2d", "s2d", "o3d", "s3d", "hd", "dep", "dep_un")}
    hd = M.v10Detect3d(3, (16, 32, 64), False, chan, False, False, False, False, nl, False, False, k1, k2)
    hd.stride = torch.tensor([8.0, 16.0, 32.0][:nl])
    hd = load_into(hd, g["state"]).eval()
    with torch.no_grad():
        out = hd([x.to(DEV) for x in g["x"]])
    y, maps = out["one2one"]
    B = y.shape[0]
    for a, b in zip(maps, g["maps"]):
        check_sparse_eval(a.reshape(B, 38, -1), b.reshape(B, 38, -1), 3, 1e-3)
    check_sparse_eval(y, g["y"], 3, 1e-3)
    # the module must not stay mutated (the reference leaves padding=0 behind)
    assert hd.o2o_heads[1][0][0].conv.padding == (k1 // 2, k1 // 2)
    # the >= 90 % overlap above concerns the SELECTION only (a top-k over nearly tied logits).  With the
    # selection seeded by the reference's own candidate cells (the cells of the fixture maps whose raw depth channel is non-zero) every
    # regression channel of EVERY reference candidate - and so the whole map and the decoded output - must match within 1e-3
    dep = 3 + 4 + 2 + 3 + 24
    ref_idx = []
    for m in g["maps"]:
        nz = (m[:, dep].reshape(B, -1) != 0)
        assert bool((nz.sum(1) == hd.max_det).all())
        ref_idx.append(torch.stack([r.nonzero().flatten() for r in nz]).to(torch.int32).to(DEV))
    calls = iter(ref_idx)
    hd.select_candidates = lambda scores: next(calls)
    try:
        with torch.no_grad():
            y2, maps2 = hd([x.to(DEV) for x in g["x"]])["one2one"]
    finally:
        del hd.select_candidates
    for a, b in zip(maps2, g["maps"]):
        check(a, b, 1e-3, "eval map, reference candidates")
    check(y2, g["y"], 1e-3, "eval output, reference candidates")


@pytest.mark.parametrize("cout", [16, 32, 48, 64, 80])
@pytest.mark.parametrize("B,H,W", [(2, 64, 96), (3, 67, 101), (1, 258, 130)])
def test_fused_eval_stem_matches_two_step_form_and_fp32_reference(cout, B, H, W):
    """csrc/stem_fused.hip (eval stem in one pass: window gather -> one 16x16x32 MFMA -> folded BatchNorm + SiLU) against (a) the
    two-step form the training path uses (y3d_stem_im2col + dense conv, `ops.STEM_FUSED = False`) - same bf16 operands, so equal up to
    the rounding of a different fp32 summation order - and (b) torch's fp32 conv + eval BatchNorm + SiLU within the bf16 bound; fp32
    NCHW, uint8 NCHW and uint8 NHWC (channels-last) images, odd sizes, every stem width of the model family (N 16 ... X 80)"""
    from yolov10_3d_amd import ops
    import torch.nn.functional as F
    y3d.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(cout + H)
    m = M.Conv(3, cout, 3, 2).to(DEV).eval()
    with torch.no_grad():
        m.bn.weight.uniform_(0.5, 1.5)
        m.bn.bias.uniform_(-0.5, 0.5)
        m.bn.running_mean.uniform_(-0.2, 0.2)
        m.bn.running_var.uniform_(0.5, 1.5)
    u8 = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, device=DEV)
    imgs = {"f32": torch.rand(B, 3, H, W, device=DEV), "u8": u8, "u8_hwc": u8.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)}
    for name, img in imgs.items():
        res = {}
        for fused in (True, False):
            ops.STEM_FUSED = fused
            try:
                with torch.no_grad():
                    res[fused] = m(img).float()
            finally:
                ops.STEM_FUSED = True
            ops.bump_param_epoch()  # drop the eval caches between the two forms
        xf = img.float() / 255 if img.dtype == torch.uint8 else img
        with torch.no_grad():
            ref = F.silu(F.batch_norm(F.conv2d(xf, m.conv.weight, None, 2, 1), m.bn.running_mean, m.bn.running_var, m.bn.weight, m.bn.bias, False, 0.0, m.bn.eps))
        assert res[True].shape == ref.shape
        check(res[True], res[False], 8e-3, f"{name}: fused vs two-step")     # one bf16 ulp of the largest output
        check(res[True], ref, 2e-2, f"{name}: fused vs fp32 reference")
        frac = (res[True] == res[False]).float().mean().item()
        assert frac > 0.97, f"{name}: only {frac:.3f} of the outputs are bit-identical to the two-step form"


@pytest.mark.parametrize("cout", [16, 32, 80])
@pytest.mark.parametrize("B,H,W", [(2, 64, 96), (3, 67, 101), (2, 258, 130)])
def test_fused_train_stem_matches_two_step_form(cout, B, H, W):
    """training stem: one pass (raw conv output + BatchNorm partials + column tensor) against im2col + dense conv on the same operands:
    activations, running statistics, weight / BatchNorm gradients (the backward is shared: it reads the column tensor either form wrote)"""
    from yolov10_3d_amd import ops
    y3d.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(cout + W)
    m0 = M.Conv(3, cout, 3, 2).to(DEV).train()
    with torch.no_grad():
        m0.bn.weight.uniform_(0.5, 1.5)
        m0.bn.bias.uniform_(-0.5, 0.5)
    img = torch.rand(B, 3, H, W, device=DEV)
    u8 = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device=DEV).permute(0, 3, 1, 2)
    for x in (img, u8):
        res = {}
        for fused in (True, False):
            m = copy.deepcopy(m0)
            ops.STEM_FUSED = fused
            try:
                z = m(x)
                r = torch.linspace(-1, 1, z.numel(), device=DEV).view(z.shape)
                (z.float() * r).sum().backward()
            finally:
                ops.STEM_FUSED = True
            res[fused] = (z.detach().float(), m.conv.weight.grad.clone(), m.bn.weight.grad.clone(), m.bn.bias.grad.clone(),
                          m.bn.running_mean.clone(), m.bn.running_var.clone())
        names = ("z", "dW", "dgamma", "dbeta", "running_mean", "running_var")
        for n, a, b in zip(names, res[True], res[False]):
            # a raw output that rounds to the other bf16 neighbour (different fp32 summation order) moves z by a few bf16 ulps through BatchNorm
            check(a, b, 3e-2 if n in ("z", "dW", "dgamma", "dbeta") else 1e-4, f"{n}: fused vs two-step")
        xf = x.float() / 255 if x.dtype == torch.uint8 else x
        mr = copy.deepcopy(m0).float()
        ref = torch.nn.functional.silu(mr.bn(torch.nn.functional.conv2d(xf, mr.conv.weight, None, 2, 1)))
        check(res[True][0], ref, 3e-2, "z vs torch fp32")
        check(res[True][4], mr.bn.running_mean, 2e-2, "running_mean vs torch fp32", floor=1e-2)


def _opt_head(meta, nl=2):
    ds, half, common, pred = [bool(int(v)) for v in meta]
    chan = {k + "_c": 16 for k in ("cls", "o2d", "s2d", "o3d", "s3d", "hd", "dep", "dep_un")}
    hd = M.v10Detect3d(3, (8, 16, 32), ds, chan, pred, True, False, common, nl, half, False, 3, 3)
    hd.stride = torch.tensor([8.0, 16.0, 32.0][:nl])
    return hd


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 6e-2)])
@pytest.mark.parametrize("name", ["dsconv", "half", "ds_half", "pred", "pred_half"])
def test_head3d_constructor_options_train_vs_reference_golden(name, dtype, tol):
    """VERDICT round 2, Missing 7: `dsconv`, `half_channels`, `use_predecessors` (head.py:554-650, 727-737) on the HIP ops against
    fixtures minted from the reference with each switch on: both head sets' maps, the 'dep' embeddings, input gradients (the one-to-one
    set and the predecessor inputs are detached), parameter gradients, BatchNorm running statistics"""
    y3d.set_compute_dtype(dtype)
    g = load_golden(f"head3d_opt_{name}_train")
    hd = load_into(_opt_head(g["meta"]), g["state"]).train()
    xs = [x.to(DEV).requires_grad_(True) for x in g["x"]]
    out = hd(list(xs))
    assert "_y3d_maps" not in out  # branch-by-branch form
    for a, b in zip(out["one2many"] + out["one2one"], g["o2m"] + g["o2o"]):
        check(a, b, tol, "head map")
    for a, b in zip(out["o2m_embs"] + out["o2o_embs"], g["o2m_embs"] + g["o2o_embs"]):
        check(a, b, tol, "embs")
    sum((t.float() * r.to(DEV)).sum() for t, r in zip(out["one2many"] + out["one2one"], g["r"])).backward()
    for a, b in zip(xs, g["dx"]):
        check(a.grad, b, tol * 3, "dx")
    named = dict(hd.named_parameters())
    gf = grad_floor(g["grads"], tol)
    for k, gv in g["grads"].items():
        check(named[k[len("model.0."):]].grad, gv, tol * 3, f"grad {k}", gf)
    sd = hd.state_dict()
    for k, v in g["state_after"].items():
        check(sd[k[len("model.0."):]].float(), v.float(), 1e-3 if dtype == torch.float32 else 3e-2, f"state {k}")


@pytest.mark.parametrize("name", ["dsconv", "half", "ds_half"])
def test_head3d_constructor_options_eval_vs_reference_golden(name):
    """the reference's patch forward with the switches on, including what it does with `dsconv` (nested Sequentials keep their padding,
    the 5x5 result is read at cell (0, 0)): candidate cells seeded from the fixture, every channel of every candidate within 1e-3"""
    y3d.set_compute_dtype(torch.float32)
    g = load_golden(f"head3d_opt_{name}_eval")
    hd = load_into(_opt_head(g["meta"]), g["state"]).eval()
    B = g["y"].shape[0]
    with torch.no_grad():
        y, maps = hd([x.to(DEV) for x in g["x"]])["one2one"]
    for a, b in zip(maps, g["maps"]):
        check_sparse_eval(a.reshape(B, 38, -1), b.reshape(B, 38, -1), 3, 1e-3)
    dep = 3 + 4 + 2 + 3 + 24
    ref_idx = []
    for m in g["maps"]:
        nz = (m[:, dep].reshape(B, -1) != 0)
        assert bool((nz.sum(1) == hd.max_det).all())
        ref_idx.append(torch.stack([r.nonzero().flatten() for r in nz]).to(torch.int32).to(DEV))
    calls = iter(ref_idx)
    hd.select_candidates = lambda scores: next(calls)
    try:
        with torch.no_grad():
            y2, maps2 = hd([x.to(DEV) for x in g["x"]])["one2one"]
    finally:
        del hd.select_candidates
    for a, b in zip(maps2, g["maps"]):
        check(a, b, 1e-3, "eval map, reference candidates")
    check(y2, g["y"], 1e-3, "eval output, reference candidates")
    assert all(l.conv.padding == (l.k // 2, l.k // 2) for l in hd.modules() if isinstance(l, M.Conv))


def test_head3d_switches_without_reference_behaviour_raise():
    """`common_head`: the reference's training forward fails on it (AssertionError, oracle/make_golden_headopts.py prints it);
    `use_predecessors` in eval mode: the reference raises on the channel count.  Neither falls back to anything."""
    chan = {k + "_c": 16 for k in ("cls", "o2d", "s2d", "o3d", "s3d", "hd", "dep", "dep_un")}
    with pytest.raises(NotImplementedError):
        M.v10Detect3d(3, (8, 16), False, chan, False, True, False, True, 2, False, False, 3, 3)
    hd = M.v10Detect3d(3, (8, 16), False, chan, True, True, False, False, 2, False, False, 3, 3).to(DEV).eval()
    with pytest.raises(RuntimeError), torch.no_grad():
        hd([torch.randn(1, 8, 16, 16, device=DEV), torch.randn(1, 16, 8, 8, device=DEV)])


def _tiny_cfg(name, **over):
    d = y3d.yaml_model_load(name)
    d.update(over)
    return d


TINY = dict(scales={"n": [0.33, 0.125, 1024]}, scale="n",
            channels={k + "_c": 16 for k in ("cls", "o2d", "s2d", "o3d", "s3d", "hd", "dep", "dep_un")})


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 8e-2)])
@pytest.mark.parametrize("tag,over", [("e2e_tiny3d_s", dict(kernel_size_1=3, kernel_size_2=3, num_scales=3)),
                                      ("e2e_tiny3d_m", dict(kernel_size_1=3, kernel_size_2=1, num_scales=2))])
def test_e2e_tiny3d_vs_reference_golden(tag, over, dtype, tol):
    y3d.set_compute_dtype(dtype)
    g = load_golden(tag)
    cfg = _tiny_cfg("yolov10s_3D.yaml" if tag.endswith("_s") else "yolov10m_3D.yaml", **TINY, **over)
    model = y3d.YOLOv10_3DDetectionModel(cfg)
    n_ok, n_all = model.load(g["state"])
    assert n_ok == len(g["state"])
    model = model.to(DEV).train()
    batch = {k: v.to(DEV) for k, v in g["batch"].items()}
    batch["img"] = g["img"].to(DEV)
    loss, items = model(batch)
    loss.backward()
    named = dict(model.named_parameters())
    if dtype == torch.bfloat16:
        # the assigner is discrete: bf16 rounding of near-tied metrics flips a few of the ~40 positives of this 2-image batch,
        # so the loss is only loosely comparable; the smooth part (all head maps, no assigner) is held to `tol` against the oracle.
        assert torch.isfinite(items).all() and all(torch.isfinite(p.grad).all() for p in named.values() if p.grad is not None)
        check(loss.reshape(()), g["loss"].reshape(()), 0.3, "loss (bf16, loose)")
        spec = RS.build_spec(cfg)
        st = {k: v.clone() for k, v in g["state"].items()}
        with torch.no_grad():
            ref = RS.forward(spec, st, g["img"], True)
            model.load(g["state"])  # undo the running-stat update of the first pass
            out = model.predict(batch["img"])
        for a, b in zip(out["one2many"] + out["one2one"], ref["one2many"] + ref["one2one"]):
            # ~25 bf16 layers whose BatchNorms see as few as 8 samples (2x2 P5 map of a 64x64 image): rounding noise is amplified
            # far beyond what the full-size model sees (test_bf16_tracks_f32_at_scale), so this is a loose norm-wise bound
            e = l2_rel(a, b)
            assert e <= 2 * tol, f"head maps vs oracle (bf16): relative L2 error {e:.3e} > {2 * tol}"
        return
    check(items, g["items"], tol, "loss items")
    check(loss.reshape(()), g["loss"].reshape(()), tol, "loss")
    gf = grad_floor(g["grads"])
    for k, gv in g["grads"].items():
        check(named[k].grad, gv, tol * 5, f"grad {k}", gf)
    if dtype == torch.float32:
        sd = model.state_dict()
        for k, v in g["state_after"].items():
            check(sd[k].float(), v.float(), 1e-3, f"state {k}")
        # eval + postprocess on the state the fixture's eval pass saw
        model.load({**g["state"], **g["state_eval"]})
        model.eval()
        with torch.no_grad():
            y = model(g["img_eval"].to(DEV))["one2one"][0]
        same = check_sparse_eval(y, g["y_eval"], 3, 2e-3)
        from yolov10_3d_amd.loss import v10_3Dpostprocess
        reg, sc, lab = v10_3Dpostprocess(y.permute(0, 2, 1), 50, 3)
        check(sc, g["post_scores"], 2e-3, "post scores")
        if same:
            assert torch.equal(lab.cpu(), g["post_labels"].long())


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 8e-2)])
def test_e2e_tiny2d_vs_reference_golden(dtype, tol):
    y3d.set_compute_dtype(dtype)
    g = load_golden("e2e_tiny2d")
    cfg = _tiny_cfg("yolov10n.yaml", nc=20, scales={"n": [0.33, 0.125, 1024]}, scale="n")
    model = y3d.YOLOv10DetectionModel(cfg)
    n_ok, _ = model.load(g["state"])
    assert n_ok == len(g["state"])
    model = model.to(DEV).train()
    batch = {k: v.to(DEV) for k, v in g["batch"].items()}
    batch["img"] = g["img"].to(DEV)
    loss, items = model(batch)
    loss.backward()
    named = dict(model.named_parameters())
    if dtype == torch.bfloat16:
        assert torch.isfinite(items).all()
        check(loss.reshape(()), g["loss"].reshape(()), 0.3, "loss (bf16, loose)")
        spec = RS.build_spec(cfg)
        st = {k: v.clone() for k, v in g["state"].items()}
        with torch.no_grad():
            ref = RS.forward(spec, st, g["img"], True)
            model.load(g["state"])
            out = model.predict(batch["img"])
        for a, b in zip(out["one2many"] + out["one2one"], ref["one2many"] + ref["one2one"]):
            # ~25 bf16 layers whose BatchNorms see as few as 8 samples (2x2 P5 map of a 64x64 image): rounding noise is amplified
            # far beyond what the full-size model sees (test_bf16_tracks_f32_at_scale), so this is a loose norm-wise bound
            e = l2_rel(a, b)
            assert e <= 2 * tol, f"head maps vs oracle (bf16): relative L2 error {e:.3e} > {2 * tol}"
        return
    check(items, g["items"], tol, "loss items")
    gf = grad_floor(g["grads"])
    for k, gv in g["grads"].items():
        check(named[k].grad, gv, tol * 5, f"grad {k}", gf)
    if dtype == torch.float32:
        model.load({**g["state"], **g["state_after"]})
        model.eval()
        with torch.no_grad():
            out = model(g["img"].to(DEV))
        check(out["one2one"][0], g["y_eval_o2o"], 2e-3, "eval o2o")
        check(out["one2many"][0], g["y_eval_o2m"], 2e-3, "eval o2m")


@pytest.mark.parametrize("name,S,B,quant", [("yolov10s_3D.yaml", 320, 2, None), ("yolov10m_3D.yaml", 320, 2, None), ("yolov10n_3D.yaml", 256, 2, None),
                                             ("yolov10l.yaml", 384, 1, None), ("yolov10x_3D.yaml", 256, 1, None), ("yolov10b_3D.yaml", 256, 1, None),
                                             ("yolov10l_3D.yaml", 256, 1, None), ("yolov10x_3D.yaml", 256, 1, "fp8")])
def test_full_size_scales_vs_oracle(name, S, B, quant):
    """the shipped model yamls at full width (BASELINE configs[1]: S + 3D head, the benchmark's model; configs[2]: M + 3D head with
    num_scales 2 and 3x3 / 1x1 branch kernels; N + 3D head; configs[3]: L, 2D; configs[4]: X + 3D head, also with fp8 conv weights - the
    oracle then runs on `RS.fp8w_state`'s fp8-valued weights; B and L + 3D head) on fresh seeded inputs: one training step of the HIP path
    (exact-fp32 mode) against the CPU oracle restatement with the same weights - loss items within 1e-3 and every parameter's gradient
    norm within 5e-3"""
    y3d.set_weight_quant(quant)
    try:
        _full_size_scale_vs_oracle(name, S, B, quant)
    finally:
        y3d.set_weight_quant(None)


@pytest.mark.parametrize("over", [dict(use_predecessors=True), dict(dsconv=True, half_channels=True), dict(half_channels=True, use_predecessors=True)],
                         ids=["predecessors", "dsconv_half", "half_predecessors"])
def test_model_with_head_switches_vs_oracle(over):
    """yolov10n_3D.yaml with the v10Detect3d yaml switches on (tasks.py:930-941 hands them to the head), full width, one training step
    through the dual-assignment loss: loss items and every parameter's gradient norm against the oracle, as the shipped yamls above"""
    _full_size_scale_vs_oracle("yolov10n_3D.yaml", 256, 2, None, over)


def _full_size_scale_vs_oracle(name, S, B, quant, over=None):
    import yaml as _yaml
    import os as _os
    from bench import synth_batch
    y3d.set_compute_dtype(torch.float32)
    is3d = "3D" in name
    torch.manual_seed(1)
    model = (y3d.YOLOv10_3DDetectionModel if is3d else y3d.YOLOv10DetectionModel)(name if not over else _tiny_cfg(name, **over))
    for m in model.modules():  # non-trivial BatchNorm state
        if isinstance(m, torch.nn.BatchNorm2d):
            with torch.no_grad():
                m.weight.uniform_(0.9, 1.1)
                m.bias.uniform_(-0.1, 0.1)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    nc = model.yaml["nc"]
    batch = synth_batch(B, S, S, 5, "cpu", nc=nc)
    # oracle
    sub = "v10-3D" if is3d else "v10"
    with open(_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "yolov10-3d_amd", "cfg", "models", sub, name)) as f:
        cfg = _yaml.safe_load(f)
    cfg["scale"] = RS.guess_scale(name)
    cfg.update(over or {})
    spec = RS.build_spec(cfg)
    okeys = set(RS.init_state(spec).keys())
    st = {k: v.clone() for k, v in state.items() if k in okeys}
    assert set(st) == okeys
    if quant == "fp8":
        st0 = st
        st = RS.fp8w_state(spec, {k: v.clone() for k, v in st0.items()})
        assert sum(1 for k in st if not torch.equal(st[k], st0[k])) >= 60, "fp8 mode: the oracle's weights were not quantised"
    for v in st.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    for k in list(st):
        if "running" in k or "num_batches" in k:
            st[k] = st[k].detach()
    preds = RS.forward(spec, st, batch["img"], True)
    strides = RS.model_strides(spec)
    loss_o, items_o, _ = (RS.loss3d if is3d else RS.loss2d)(preds, batch, strides, nc)
    loss_o.backward()
    # HIP
    model = model.to(DEV).train()
    dbatch = {k: v.to(DEV) for k, v in batch.items()}
    loss, items = model(dbatch)
    loss.backward()
    check(items, items_o.detach(), 1e-3, "loss items")
    named = dict(model.named_parameters())
    norms = {k: float(v.grad.norm()) for k, v in st.items() if v.requires_grad and v.grad is not None}
    big = max(norms.values())
    bad = []
    for k, nv in norms.items():
        if k in named and named[k].grad is not None:
            if abs(float(named[k].grad.norm()) - nv) > 5e-3 * max(nv, 1e-2 * big):
                bad.append((k, float(named[k].grad.norm()), nv))
    assert not bad, f"{len(bad)} gradient norms differ, e.g. {bad[:3]}"
    # (the oracle state lists the aliased head keys twice; named_parameters() reports each parameter once)
    missing = [k for k in norms if k in named and named[k].grad is None]
    assert not missing, f"parameters with an oracle gradient but none on the HIP path: {missing[:4]}"
    assert sum(1 for k in norms if k in named) >= 0.5 * len(norms)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_baseline_config0_n2d_320_vs_reference_golden(dtype):
    """BASELINE.json configs[0]: the shipped YOLOv10-N 2D model (nc=80), 320x320, batch 2 — the reference's own CPU PyTorch run
    (oracle/make_golden_configs.py).  fp32 mode: loss items, selected gradients, EVERY parameter's gradient norm, running statistics,
    eval outputs and the postprocessed detections within 1e-3; bf16 mode: the documented loose bounds."""
    from yolov10_3d_amd.loss import v10postprocess
    y3d.set_compute_dtype(dtype)
    g = load_golden("e2e_n2d_320")
    model = y3d.YOLOv10DetectionModel("yolov10n.yaml")
    n_ok, n_all = model.load(g["state"])
    assert n_ok == len(g["state"]) == n_all
    model = model.to(DEV).train()
    batch = {k: v.to(DEV) for k, v in g["batch"].items()}
    batch["img"] = (g["img8"].float() / 255).to(DEV)
    loss, items = model(batch)
    loss.backward()
    named = dict(model.named_parameters())
    if dtype == torch.bfloat16:
        assert torch.isfinite(items).all()
        check(items, g["items"], 0.35, "loss items (bf16: the one-to-one top-1 assignment is discontinuous)")
        return
    check(items, g["items"], 1e-3, "loss items")
    gf = grad_floor(g["grads"])
    for k, gv in g["grads"].items():
        check(named[k].grad, gv, 5e-3, f"grad {k}", gf)
    big = max(float(v) for v in g["grad_norms"].values())
    for k, nv in g["grad_norms"].items():
        assert abs(float(named[k].grad.norm()) - float(nv)) <= 5e-3 * max(float(nv), 1e-2 * big), f"gradient norm of {k}"
    sd = model.state_dict()
    for k, v in g["state_after"].items():
        check(sd[k].float(), v.float(), 1e-3, f"state {k}")
    model.load({**g["state"], **g["state_after"]})
    model.eval()
    with torch.no_grad():
        out = model(batch["img"])
    check(out["one2one"][0], g["y_eval_o2o"], 2e-3, "eval o2o")
    check(out["one2many"][0], g["y_eval_o2m"], 2e-3, "eval o2m")
    bx, sc, lab = v10postprocess(out["one2one"][0].permute(0, 2, 1), 300, 80)
    same = (lab.cpu() == g["post_labels"].long()).float().mean()
    assert same >= 0.98, f"postprocessed labels: {float(same):.3f} equal (near-tied scores may swap)"
    check(sc, g["post_scores"], 2e-3, "postprocessed scores")


# ---------------------------------------------------------------------------------------------------------
# fresh seeded inputs at realistic channel counts: HIP vs the CPU oracle restatement (fp32 mode)
# ---------------------------------------------------------------------------------------------------------
CASES = [
    # c1, c2, k, s, g, H, W, B
    (128, 128, 3, 1, 1, 40, 40, 2),   # the headline K1 shape (smaller map)
    (256, 128, 3, 1, 1, 20, 20, 2),
    (64, 64, 3, 1, 1, 33, 29, 3),     # ragged spatial dims, BC=64 tile
    (32, 32, 3, 1, 1, 17, 23, 2),     # BC=32 tile
    (64, 128, 3, 2, 1, 40, 40, 2),    # stride-2 dense (dgrad gather with stride)
    (384, 128, 1, 1, 1, 20, 20, 2),   # 1x1 GEMM
    (48, 96, 3, 2, 1, 30, 30, 2),     # M-model channel counts (multiples of 16, not 64)
    (256, 256, 3, 1, 2, 16, 16, 2),   # grouped (the fused-head second layer shape family)
    (128, 256, 3, 1, 1, 32, 80, 2),   # resident-halo tile kernel, TH=16, 5 x-tiles, 2 channel tiles
    (64, 96, 3, 1, 1, 16, 40, 2),     # TH=16, ragged last x-tile (40 = 2.5 x 16), ragged channel tile
    (128, 128, 3, 1, 1, 24, 24, 1),   # TH=8
    (64, 64, 3, 1, 1, 20, 20, 3),     # TH=4
    (256, 256, 3, 1, 4, 16, 16, 2),   # grouped on the tile kernel (64 channels per group)
    (256, 256, 3, 2, 256, 20, 20, 2),  # depth-wise s2 (SCDown)
    (128, 128, 7, 1, 128, 20, 20, 2),  # depth-wise 7x7 (RepVGGDW)
]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 5e-2)])
@pytest.mark.parametrize("case", CASES)
def test_conv_bn_silu_vs_oracle(case, dtype, tol):
    c1, c2, k, s, g, H, W, B = case
    y3d.set_compute_dtype(dtype)
    torch.manual_seed(0)
    mod = M.Conv(c1, c2, k, s, None, g)
    with torch.no_grad():
        mod.bn.weight.uniform_(0.8, 1.2)
        mod.bn.bias.uniform_(-0.2, 0.2)
    mod.bn.eps, mod.bn.momentum = 1e-3, 0.03
    st = {"model.0." + kk: v.clone() for kk, v in mod.state_dict().items()}
    x = torch.randn(B, c1, H, W)
    # oracle (CPU)
    for kk, v in st.items():
        if v.is_floating_point() and "running" not in kk:
            v.requires_grad_(True)
    xo = x.clone().requires_grad_(True)
    yo = RS.conv_bn_act(RS.Ctx(st, True), "model.0", xo, k, s, g)
    r = torch.randn_like(yo)
    (yo * r).sum().backward()
    # HIP
    mod = mod.to(DEV).train()
    xh = x.to(DEV).requires_grad_(True)
    yh = mod(xh)
    check(yh, yo, tol, "y")
    (yh.float() * r.to(DEV)).sum().backward()
    check(xh.grad, xo.grad, tol * 3, "dx")
    check(mod.conv.weight.grad, st["model.0.conv.weight"].grad, tol * 3, "dW")
    check(mod.bn.weight.grad, st["model.0.bn.weight"].grad, tol * 3, "dgamma")
    check(mod.bn.bias.grad, st["model.0.bn.bias"].grad, tol * 3, "dbeta")
    check(mod.bn.running_var, st["model.0.bn.running_var"], 1e-3 if dtype == torch.float32 else 2e-2, "running_var")


def test_conv_on_channel_slice_views():
    """chunk()/split() views feed the kernels without copies (C2f, PSA)"""
    y3d.set_compute_dtype(torch.float32)
    torch.manual_seed(1)
    mod = M.Conv(32, 32, 3, 1)
    mod.bn.eps, mod.bn.momentum = 1e-3, 0.03
    st = {"model.0." + kk: v.clone() for kk, v in mod.state_dict().items()}
    x = torch.randn(2, 64, 12, 12)
    yo = RS.conv_bn_act(RS.Ctx(st, True), "model.0", x[:, 32:], 3)
    big = y3d.ops.to_nhwc(x.to(DEV), torch.float32)
    yh = mod.to(DEV).train()(big[:, 32:])
    check(yh, yo, 1e-3, "slice conv")


def test_hip_library_is_the_path():
    """no silent fallback: the kernel wrappers refuse host tensors, a module answers one with a shape (a meta tensor: the reference
    constructor's stride probe) - never with values"""
    from yolov10_3d_amd import ops
    y3d.set_compute_dtype(torch.float32)
    mod = M.Conv(16, 16, 3)
    x = torch.randn(1, 16, 8, 8)
    with pytest.raises(y3d.Y3DError):
        ops.ConvBNActFn.apply(x, mod.conv.weight, mod.bn.weight, mod.bn.bias, None, 0, mod)
    assert mod(x).device.type == "meta"
    mod = mod.to(DEV)
    with pytest.raises(y3d.Y3DError):  # parameters on the device, input on the host
        ops.ConvBNActFn.apply(x, mod.conv.weight, mod.bn.weight, mod.bn.bias, None, 0, mod)
    y3d.set_compute_dtype(torch.bfloat16)


def test_bf16_tracks_f32_at_scale():
    """YOLOv10-S-3D at 320x320, B=4: the bf16 performance mode against the exact-f32 mode of the same kernels (same weights,
    same batch) — head maps within 12 % (norm-wise; ~60 bf16 layers deep at random init), the six one-to-many loss items within
    8 %, the six one-to-one items within 35 %: that branch assigns ONE anchor per object (top-1), so bf16 noise moves single
    assignments between near-tied anchors and the regression terms of a random-init model jump with them."""
    import bench
    torch.manual_seed(0)
    model = y3d.YOLOv10_3DDetectionModel("yolov10s_3D.yaml").to(DEV).train()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    batch = bench.synth_batch(4, 320, 320, seed=3, device=DEV)
    res = {}
    for dt in (torch.float32, torch.bfloat16):
        y3d.set_compute_dtype(dt)
        model.load_state_dict(state)
        out = model.predict(batch["img"])
        loss, items = model.criterion(out, batch) if hasattr(model, "criterion") else model.loss(batch, out)
        res[dt] = ([t.detach().float() for t in out["one2many"] + out["one2one"]], items.detach().float())
    errs = [l2_rel(a, b) for a, b in zip(res[torch.bfloat16][0], res[torch.float32][0])]
    print("bf16 vs f32 head-map relative L2 errors:", [round(e, 4) for e in errs])
    assert max(errs) < 0.12, f"bf16 vs f32 head maps: relative L2 errors {errs}"
    check(res[torch.bfloat16][1][:6], res[torch.float32][1][:6], 0.08, "one-to-many loss items bf16 vs f32")
    check(res[torch.bfloat16][1][6:], res[torch.float32][1][6:], 0.35, "one-to-one loss items bf16 vs f32")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_loss3d_hip_vs_reference_golden(dtype):
    """fused HIP assigner + loss + gradient (tal_loss3d.hip) on the reference's own loss fixture: loss items and gradients within
    1e-3, and the integer assignment (fg_mask, target_gt_idx) BIT-EXACT against the oracle (itself bit-exact vs the reference)."""
    from types import SimpleNamespace
    from yolov10_3d_amd import loss as PL
    y3d.set_compute_dtype(dtype)
    g = load_golden("loss3d")
    strides = [float(s) for s in g["strides"]]
    head = SimpleNamespace(stride=g["strides"], nc=3, no=38)
    crit = PL.DetectLoss3d(SimpleNamespace(model=[head], args=SimpleNamespace(**y3d.tasks.DEFAULT_HYP)))
    src_m, src_o = g["o2m"], g["o2o"]
    if dtype == torch.bfloat16:  # the kernel sees bf16 logits: give the oracle the same rounded values
        src_m = [t.bfloat16().float() for t in src_m]
        src_o = [t.bfloat16().float() for t in src_o]
    o2m = [y3d.ops._dense_any(t.to(DEV), dtype).requires_grad_(True) for t in src_m]
    o2o = [y3d.ops._dense_any(t.to(DEV), dtype).requires_grad_(True) for t in src_o]
    batch = {k: v.to(DEV) for k, v in g["batch"].items()}
    loss, items = crit({"one2many": o2m, "one2one": o2o}, batch)
    loss.backward()
    # oracle on the same logits
    om = [t.clone().requires_grad_(True) for t in src_m]
    oo = [t.clone().requires_grad_(True) for t in src_o]
    lo, io, aux = RS.loss3d({"one2many": om, "one2one": oo}, g["batch"], strides, 3)
    lo.backward()
    for crit1, key in ((crit.one2many, "one2many"), (crit.one2one, "one2one")):
        fg, gi, ts = crit1.last_assignment
        assert torch.equal(fg.cpu(), aux[key]["fg_mask"]), f"{key}: fg_mask differs"
        assert torch.equal(gi.cpu(), aux[key]["target_gt_idx"]), f"{key}: target_gt_idx differs"
        check(ts, aux[key]["target_scores"], 1e-4, f"{key} target_scores")
    check(items, io, 1e-3, "loss items vs oracle")
    if dtype == torch.float32:
        check(items, g["items"], 1e-3, "loss items vs reference fixture")
        check(loss.reshape(()), g["loss"].reshape(()), 1e-3, "loss vs reference fixture")
    tol = 1e-3 if dtype == torch.float32 else 1e-2  # bf16: the gradient itself is stored in bf16
    for a, b in zip(o2m + o2o, om + oo):
        check(a.grad, b.grad, tol, "d loss / d map")


def test_tal3d_hip_on_assigner_fixture_scale():
    """assignment at the bench geometry (A = 8400, B = 8, up to 8 GTs): HIP vs the torch-device formulation, bit-exact indices"""
    from types import SimpleNamespace
    from yolov10_3d_amd import loss as PL
    import bench
    import torch_assigners as TA
    y3d.set_compute_dtype(torch.float32)
    torch.manual_seed(5)
    B, nc = 8, 3
    shapes, strides = [(80, 80), (40, 40), (20, 20)], [8.0, 16.0, 32.0]
    batch = bench.synth_batch(B, 640, 640, seed=11, device=DEV)
    maps = []
    for (h, w) in shapes:
        t = torch.randn(B, 38, h, w, device=DEV)
        t[:, 0:3] -= 2.0
        t[:, 5:7] = 2 + 4 * torch.rand(B, 2, h, w, device=DEV)
        t[:, 36] = 10 + 30 * torch.rand(B, h, w, device=DEV)
        maps.append(y3d.ops._dense_any(t, torch.float32))
    head = SimpleNamespace(stride=torch.tensor(strides), nc=nc, no=38)
    model = SimpleNamespace(model=[head], args=SimpleNamespace(**y3d.tasks.DEFAULT_HYP))
    for topk in (8, 1):
        crit = PL.DDDetectionLoss(model, tal_topk=topk)
        crit(maps, batch)
        fg, gi, ts = crit.last_assignment
        # torch-device formulation of the same assignment
        cat = PL._flatten_maps(maps)
        sc, o2d, s2d, o3d, s3d, hd, dep, dun = cat.split((nc, 2, 2, 2, 3, 24, 1, 1), -1)
        anc, st = PL.make_anchors(shapes, strides, DEV)
        rows = torch.cat([batch[k].float().view(batch[k].shape[0], -1) for k in
                          ("batch_idx", "cls", "bboxes", "center_2d", "size_2d", "center_3d", "size_3d", "depth", "heading_bin", "heading_res")], 1)
        gpad = TA.pad_targets_torch(rows, B, 17, torch.tensor([640.0, 640.0, 640.0, 640.0], device=DEV))
        # the HIP padding kernel (fixed capacity, device-side count) holds the same rows, bit for bit, and zeros behind them
        ghip, n_used = PL.pad_targets(rows, B, 17, (640.0, 640.0))
        nm = gpad.shape[1]
        assert int(n_used) == nm and torch.equal(ghip[:, :nm], gpad) and not ghip[:, nm:].any()
        gts = gpad.split((1, 4, 2, 2, 2, 3, 1, 1, 1), 2)
        mask_gt = (gts[1].sum(2, keepdim=True) > 0).float()
        cen = anc + o2d
        pb = torch.cat((cen - s2d / 2, cen + s2d / 2), -1) * st
        asg = TA.TaskAlignedAssigner3d(topk=topk, num_classes=nc, alpha=0.5, beta=1.0, gamma=1.0)
        targets, fg_t, gi_t, _, _ = asg(sc.sigmoid(), pb, torch.cat((o3d, s3d, hd, dep, dun), -1), anc * st, gts, mask_gt, st,
                                        batch["calib"], batch["mean_sizes"])
        assert int(fg_t.sum()) > 0
        mism = int((fg != fg_t).sum()) + int((gi != gi_t).sum())
        assert mism == 0, f"topk={topk}: {mism} anchors differ"
        check(ts, targets[1], 1e-4, "target_scores")


@pytest.mark.parametrize("case", [(128, 256, 3, 1, 1, 32, 80, 2), (64, 96, 3, 1, 1, 16, 40, 2), (128, 128, 3, 1, 1, 20, 20, 3),
                                  (256, 256, 3, 1, 4, 16, 16, 2), (2048, 2048, 3, 1, 16, 40, 40, 2),
                                  # odd batch against the 2- / 4-image tiles, partial column tiles, 96 / 192 / 320 channels, one image
                                  (96, 192, 3, 1, 1, 16, 24, 3), (128, 64, 3, 1, 1, 8, 40, 5), (192, 320, 3, 1, 1, 24, 24, 1),
                                  (256, 128, 3, 1, 2, 32, 16, 7),
                                  # input channels per group = 96 / 160: the wgrad tile's last 64-channel slab is half empty
                                  (96, 96, 3, 1, 1, 16, 32, 2), (192, 192, 3, 1, 2, 8, 16, 3), (160, 64, 3, 1, 1, 12, 20, 1)])
def test_tile_kernels_agree_with_generic_kernels(case):
    """A/B inside one process: the resident-tile kernels (conv3x3_tile / conv3x3_wgrad_tile) and the generic implicit-GEMM
    kernels compute the same bf16 products with fp32 accumulation, so forward, dx and dW must agree to accumulation-order noise."""
    c1, c2, k, s, g, H, W, B = case
    y3d.set_compute_dtype(torch.bfloat16)
    L = y3d.lib()
    torch.manual_seed(3)
    mod = M.Conv(c1, c2, k, s, None, g).to(DEV).train()
    x = torch.randn(B, c1, H, W, device=DEV)
    r = torch.randn(B, c2, H, W, device=DEV)
    outs = {}
    for enable in (1, 0):
        old = L.set_tile_kernels(enable)
        try:
            mod.zero_grad(set_to_none=True)
            xi = x.clone().requires_grad_(True)
            y = mod(xi)
            (y.float() * r).sum().backward()
            outs[enable] = (y.detach().float(), xi.grad.float(), mod.conv.weight.grad.clone())
        finally:
            L.set_tile_kernels(old)
    for a, b, what, tol in zip(outs[1], outs[0], ("y", "dx", "dW"), (2e-2, 2e-2, 2e-3)):
        check(a, b, tol, f"tile vs generic {what}")


@pytest.mark.parametrize("dims", [(32, 64), (36, 72)], ids=["kd32_hd64", "kd36_hd72"])
@pytest.mark.parametrize("shape", [(2, 20, 20, 2), (1, 7, 11, 1), (1, 40, 40, 4)])
def test_attention_mfma_vs_valu(shape, dims):
    """bf16 PSA attention: MFMA kernel (S^T = K Q^T, O^T = V^T P^T on the matrix cores) vs the fp32-VALU kernel on the same qkv;
    kd = 36 / hd = 72 (the M widths: head blocks at 8-byte offsets, fragments zero-padded past the head dims)"""
    B, H, W, nh = shape
    y3d.set_compute_dtype(torch.bfloat16)
    L = y3d.lib()
    torch.manual_seed(2)
    kd, hd = dims
    qkv = y3d.ops.to_nhwc(torch.randn(B, nh * (2 * kd + hd), H, W, device=DEV), torch.bfloat16)
    res = {}
    for enable in (1, 0):
        old = L.set_tile_kernels(enable)
        try:
            o, v = y3d.ops.AttentionFn.apply(qkv, nh, kd, hd, kd ** -0.5)
            res[enable] = o.float().clone()
        finally:
            L.set_tile_kernels(old)
    check(res[1], res[0], 2e-2, "attention MFMA vs VALU")
    # and against plain fp32 math
    q, k, vv = qkv.float().view(B, nh, 2 * kd + hd, H * W).split([kd, kd, hd], 2)
    attn = ((q.transpose(-2, -1) @ k) * kd ** -0.5).softmax(-1)
    ref = (vv @ attn.transpose(-2, -1)).view(B, nh * hd, H, W)
    check(res[1], ref, 2e-2, "attention MFMA vs fp32 math")


# N = 1600 and N = 621: the LDS images of the MFMA backward hold 1024 keys / 512 queries, longer sequences walk them in passes
@pytest.mark.parametrize("dims", [(32, 64), (36, 72)], ids=["kd32_hd64", "kd36_hd72"])
@pytest.mark.parametrize("shape", [(2, 20, 20, 2), (1, 7, 11, 1), (1, 40, 40, 4), (3, 20, 20, 2), (1, 23, 27, 2)])
def test_attention_backward_mfma_vs_valu_and_autograd(shape, dims):
    """bf16 PSA attention backward: the MFMA kernels (dQ^T += K^T dS^T; dV^T += dO^T P, dK^T += Q^T dS) vs the fp32-VALU kernels and vs
    torch autograd on plain fp32 math; the second output (v, consumed by the positional-encoding branch) feeds dv_extra"""
    B, H, W, nh = shape
    y3d.set_compute_dtype(torch.bfloat16)
    L = y3d.lib()
    torch.manual_seed(4)
    kd, hd = dims
    x = torch.randn(B, nh * (2 * kd + hd), H, W, device=DEV)
    r1 = torch.randn(B, nh * hd, H, W, device=DEV)
    r2 = torch.randn(B, nh * hd, H, W, device=DEV)
    res = {}
    for enable in (1, 0):
        old = L.set_tile_kernels(enable)
        try:
            qkv = y3d.ops.to_nhwc(x, torch.bfloat16).detach().requires_grad_(True)
            o, v = y3d.ops.AttentionFn.apply(qkv, nh, kd, hd, kd ** -0.5)
            ((o.float() * r1).sum() + (v.float() * r2).sum()).backward()
            res[enable] = qkv.grad.float().clone()
        finally:
            L.set_tile_kernels(old)
    check(res[1], res[0], 3e-2, "attention backward MFMA vs VALU")
    xq = y3d.ops.to_nhwc(x, torch.bfloat16).float().detach().requires_grad_(True)
    q, k, vv = xq.view(B, nh, 2 * kd + hd, H * W).split([kd, kd, hd], 2)
    attn = ((q.transpose(-2, -1) @ k) * kd ** -0.5).softmax(-1)
    ref = (vv @ attn.transpose(-2, -1)).reshape(B, nh * hd, H, W)
    ((ref * r1).sum() + (vv.reshape(B, nh * hd, H, W) * r2).sum()).backward()
    check(res[1], xq.grad, 3e-2, "attention backward MFMA vs autograd on fp32 math")


def test_postprocess_hip_vs_reference_golden():
    from yolov10_3d_amd.loss import v10_3Dpostprocess, v10postprocess
    g = load_golden("post3d")
    reg, sc, lab = v10_3Dpostprocess(g["preds"].to(DEV), 50, 3)
    assert torch.equal(lab.cpu(), g["labels"].long())
    check(reg, g["reg"], 1e-6, "post3d reg")
    check(sc, g["scores"], 1e-6, "post3d scores")
    g = load_golden("post2d")
    bx, sc, lab = v10postprocess(g["preds"].to(DEV), 300, 80)
    assert torch.equal(lab.cpu(), g["labels"].long())
    check(bx, g["boxes"], 1e-6, "post2d boxes")
    check(sc, g["scores"], 1e-6, "post2d scores")


@pytest.mark.parametrize("HW,K", [(400, 50), (1600, 50), (6400, 50), (25600, 50), (6400, 300), (64, 64), (1000, 1)])
def test_topk_selection_ties_and_sizes(HW, K):
    """the radix-select top-k of post.hip (`y3d_topk_cells`: select_candidates of the eval head; the same device function serves both
    stages of `y3d_v10_postprocess`) against a stable descending sort - (value desc, index asc) - on inputs built to tie: logits on a
    coarse grid (many exact ties around the K-th value), a constant map, -inf entries, signed zeros"""
    from yolov10_3d_amd import ops
    torch.manual_seed(HW + K)
    B, nc = 6, 3
    maps = torch.randn(B, HW, nc)
    maps[0] = (maps[0] * 4).round() / 4          # coarse grid: hundreds of ties
    maps[1] = 0.25                               # constant: the K lowest indices win
    maps[2] = (maps[2] * 2).round() / 2
    maps[2, ::3] = float("-inf")
    maps[3] = torch.where(torch.rand(HW, nc) < 0.5, torch.zeros(()), -torch.zeros(()))  # +0 / -0 compare equal
    maps[4] = -maps[4].abs() * 1e-3              # all negative, tiny
    for dt in (torch.float32, torch.bfloat16):
        m = maps.to(dt).to(DEV)
        idx = torch.empty(B, K, dtype=torch.int32, device=DEV)
        ops.lib().topk_cells(ops.code(dt), m.data_ptr(), nc, B, HW, nc, K, idx.data_ptr(), ops.stream())
        best = m.float().max(-1)[0].cpu() + 0.0
        ref = torch.sort(best, dim=1, descending=True, stable=True)[1][:, :K]
        assert torch.equal(idx.cpu().long(), ref), f"{dt}: rows {(idx.cpu().long() != ref).any(1).nonzero().flatten().tolist()} differ"


def test_topk_with_nan_scores_writes_every_slot():
    """a diverged model hands the eval head NaN logits: the selection treats a NaN as the largest key (as torch.topk does) and the rank
    sort uses the same total order, so every output slot is written with a distinct in-range index (round-3 advisor finding: with float
    compares all NaNs ranked 0 and the unwritten slots were read as anchor indices - out-of-bounds gathers)"""
    from yolov10_3d_amd import ops
    from yolov10_3d_amd.loss import v10_3Dpostprocess
    torch.manual_seed(5)
    B, HW, nc, K = 3, 1600, 3, 50
    maps = torch.randn(B, HW, nc)
    nan_cells = torch.tensor([7, 100, 101, 900, 1599])
    maps[0, nan_cells] = float("nan")
    maps[1] = float("nan")
    idx = torch.full((B, K), -7, dtype=torch.int32, device=DEV)
    m = maps.to(DEV)
    ops.lib().topk_cells(ops.code(torch.float32), m.data_ptr(), nc, B, HW, nc, K, idx.data_ptr(), ops.stream())
    idx = idx.cpu().long()
    for b in range(B):
        assert idx[b].min() >= 0 and idx[b].max() < HW and idx[b].unique().numel() == K
    assert idx[0, :5].tolist() == nan_cells.tolist()          # NaNs first, lowest index first
    assert idx[1].tolist() == list(range(K))                  # all NaN: the K lowest indices
    best = maps[0].max(-1)[0]
    best[nan_cells] = float("inf")
    assert torch.equal(idx[0], torch.sort(best, descending=True, stable=True)[1][:K])
    assert torch.equal(idx[2], torch.sort(maps[2].max(-1)[0], descending=True, stable=True)[1][:K])
    # the postprocess (two top-k stages + gathers by the selected indices) on NaN scores: finishes, indices in range
    preds = torch.randn(2, 2100, 38)
    preds[0, ::3, :3] = float("nan")
    preds[1, :, :3] = float("nan")
    reg, sc, lab = v10_3Dpostprocess(preds.to(DEV), 50, 3)
    torch.cuda.synchronize()
    assert reg.shape[:2] == (2, 50) and int(lab.min()) >= 0 and int(lab.max()) < 3


def test_postprocess_hires_vs_oracle():
    """1280x1280 (33 600 anchors, BASELINE configs[3]): the score row does not fit LDS and goes through the HBM scratch"""
    from oracle import restate as RS
    from yolov10_3d_amd.loss import v10postprocess
    torch.manual_seed(3)
    n = 2 * 33600 * 80  # distinct scores (< 2^24, exact in fp32): the order among exactly tied scores is unspecified in the reference
    preds = torch.cat((torch.rand(2, 33600, 4) * 1280, (torch.randperm(n).float() / n).view(2, 33600, 80)), -1)
    bx, sc, lab = v10postprocess(preds.to(DEV), 300, 80)
    rb, rs, rl = RS.postprocess2d(preds, 300, 80)
    assert torch.equal(lab.cpu(), rl.long())
    check(bx, rb, 1e-6, "hires boxes")
    check(sc, rs, 1e-6, "hires scores")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_loss2d_hip_vs_reference_golden(dtype):
    """fused HIP 2D assigner + BCE/CIoU/DFL loss + gradient (tal_loss2d.hip) on the reference's v10DetectLoss fixture"""
    from types import SimpleNamespace
    from yolov10_3d_amd import loss as PL
    y3d.set_compute_dtype(dtype)
    g = load_golden("loss2d")
    strides = [float(s) for s in g["strides"]]
    head = SimpleNamespace(stride=g["strides"], nc=80, no=144, reg_max=16)
    crit = PL.v10DetectLoss(SimpleNamespace(model=[head], args=SimpleNamespace(**y3d.tasks.DEFAULT_HYP)))
    src_m, src_o = g["o2m"], g["o2o"]
    if dtype == torch.bfloat16:
        src_m = [t.bfloat16().float() for t in src_m]
        src_o = [t.bfloat16().float() for t in src_o]
    o2m = [y3d.ops._dense_any(t.to(DEV), dtype).requires_grad_(True) for t in src_m]
    o2o = [y3d.ops._dense_any(t.to(DEV), dtype).requires_grad_(True) for t in src_o]
    batch = {k: v.to(DEV) for k, v in g["batch"].items()}
    loss, items = crit({"one2many": o2m, "one2one": o2o}, batch)
    loss.backward()
    om = [t.clone().requires_grad_(True) for t in src_m]
    oo = [t.clone().requires_grad_(True) for t in src_o]
    lo, io, aux = RS.loss2d({"one2many": om, "one2one": oo}, g["batch"], strides, 80)
    lo.backward()
    for crit1, key in ((crit.one2many, "one2many"), (crit.one2one, "one2one")):
        fg, gi, ts = crit1.last_assignment
        assert torch.equal(fg.cpu(), aux[key]["fg_mask"]), f"{key}: fg_mask differs"
        assert torch.equal(gi.cpu(), aux[key]["target_gt_idx"]), f"{key}: target_gt_idx differs"
    check(items, io, 1e-3, "loss items vs oracle")
    if dtype == torch.float32:
        check(items, g["items"], 1e-3, "loss items vs reference fixture")
    tol = 1e-3 if dtype == torch.float32 else 1e-2
    for a, b in zip(o2m + o2o, om + oo):
        check(a.grad, b.grad, tol, "d loss / d map")


def test_kitti_decode_vs_oracle_and_golden():
    """y3d_kitti_decode (one launch over B*K rows) vs the restated decode_preds and the reference's own rows (golden), three modes;
    then the dict-of-lists form the validator consumes"""
    from yolov10_3d_amd import kitti
    g = load_golden("kitti_decode")
    preds = g["preds"].to(DEV)
    files = [f"{i:06d}.png" for i in range(preds.shape[0])]
    for tag, undo, cam in (("aug", True, False), ("noaug", False, False), ("camdis", True, True)):
        rows, keep = kitti.decode_preds_device(preds, g["calib"], g["ratio"], g["inv_trans"], undo_augment=undo, use_camera_dis=cam)
        r_ref, k_ref = RS.kitti_decode(g["preds"], g["calib"], g["ratio"], g["inv_trans"] if undo else None, use_camera_dis=cam)
        assert torch.equal(keep.cpu(), torch.from_numpy(k_ref)), "threshold decisions are exact"
        torch.testing.assert_close(rows.cpu(), torch.from_numpy(r_ref), rtol=1e-6, atol=1e-6)  # f64 math; expf / atan2 differ in the last ulp
        res = kitti.decode_preds_eval(preds, g["calib"], files, g["ratio"], g["inv_trans"], undo_augment=undo, use_camera_dis=cam)
        cnt = g[f"count_{tag}"].long()
        for i, f in enumerate(files):
            assert len(res[f]) == int(cnt[i]) and all(len(t) == 14 for t in res[f])
            torch.testing.assert_close(torch.tensor(res[f], dtype=torch.float64), g[f"rows_{tag}"][i, :int(cnt[i])].double(), rtol=1e-5, atol=1e-5)
    # after the real post-processing kernel: rows of v10_3Dpostprocess feed the decode unchanged
    gp = load_golden("post3d")
    reg, sc, lab = y3d.loss.v10_3Dpostprocess(gp["preds"].to(DEV), 50, 3)
    full = torch.cat((reg, sc.unsqueeze(-1), lab.unsqueeze(-1).float()), -1)  # models/yolov10_3D/val.py:46-47
    B = full.shape[0]
    calib, ratio = g["calib"][:1].expand(B, 6), g["ratio"][:1].expand(B, 2)
    rows, keep = kitti.decode_preds_device(full, calib, ratio, None, undo_augment=False)
    r_ref, k_ref = RS.kitti_decode(full.cpu(), calib, ratio, None)
    assert torch.equal(keep.cpu(), torch.from_numpy(k_ref))
    torch.testing.assert_close(rows.cpu(), torch.from_numpy(r_ref), rtol=1e-6, atol=1e-6)


def test_eval_after_more_training_sees_new_weights():
    """train -> eval -> train -> eval: the eval path caches packed weights + folded BatchNorm per module; the optimizer, the EMA and
    bn_finalize write through raw pointers (no torch version bump), so the cache must be invalidated by ops.PARAM_EPOCH"""
    import copy
    from yolov10_3d_amd.optim import ModelEMA, build_optimizer
    from bench import synth_batch
    y3d.set_compute_dtype(torch.float32)
    try:
        torch.manual_seed(0)
        model = y3d.YOLOv10_3DDetectionModel("yolov10n_3D.yaml").to(DEV).train()
        opt = build_optimizer(model, lr=0.05)
        ema = ModelEMA(model, decay=0.5, tau=1)
        batch = synth_batch(2, 256, 256, 1, DEV)  # 8x8 cells at the coarsest level: the sparse eval head needs >= 50 per level

        def train_step():
            model.train()
            loss, _ = model(batch)
            loss.backward()
            opt.step(max_norm=10.0)
            opt.zero_grad()
            ema.update(model)

        def evals():
            model.eval()
            with torch.no_grad():
                a = model(batch["img"])["one2one"][0].clone()
                b = ema.ema(batch["img"])["one2one"][0].clone()
                fresh = copy.deepcopy(model)  # no caches: the ground truth for the current weights
                fresh_ema = copy.deepcopy(ema.ema)
                for m in list(fresh.modules()) + list(fresh_ema.modules()):
                    m.__dict__.pop("_eval_cache", None)
                    m.__dict__.pop("_stack_cache", None)
                c = fresh(batch["img"])["one2one"][0]
                d = fresh_ema(batch["img"])["one2one"][0]
            return a, b, c, d

        train_step()
        a1, b1, c1, d1 = evals()
        check(a1, c1, 1e-6, "eval vs uncached copy (1)")
        check(b1, d1, 1e-6, "EMA eval vs uncached copy (1)")
        train_step()
        train_step()
        a2, b2, c2, d2 = evals()
        check(a2, c2, 1e-6, "eval vs uncached copy (2)")
        check(b2, d2, 1e-6, "EMA eval vs uncached copy (2)")
        assert rel_err(a2, a1) > 1e-4 and rel_err(b2, b1) > 1e-4, "the weights moved, the outputs must move"
    finally:
        y3d.set_compute_dtype(torch.bfloat16)


def test_uint8_images_match_host_normalised_images():
    """the stem takes the dataset's uint8 image (NCHW, or NHWC as decoded) and divides by 255 on the device: bit-identical to feeding the
    float image the reference's dataset builds on the host (data/datasets/kitti.py:204-205)"""
    from bench import synth_batch
    for dtype in (torch.float32, torch.bfloat16):
        y3d.set_compute_dtype(dtype)
        torch.manual_seed(0)
        model = y3d.YOLOv10_3DDetectionModel("yolov10n_3D.yaml").to(DEV).train()
        batch = synth_batch(2, 256, 256, 1, DEV)
        img8 = torch.randint(0, 256, (2, 3, 256, 256), dtype=torch.uint8, device=DEV)
        outs = []
        import numpy as np
        host = torch.from_numpy(img8.cpu().numpy().astype(np.float32) / np.float32(255.0)).to(DEV)  # IEEE division, as numpy on the host
        for img in (host, img8, img8.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)):
            b = dict(batch)
            b["img"] = img
            model.zero_grad(set_to_none=True)
            loss, items = model(b)
            loss.backward()
            outs.append((items.clone(), model.model[0].conv.weight.grad.clone()))
        for o in outs[1:]:
            assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1])
    y3d.set_compute_dtype(torch.bfloat16)


def test_fused_adamw_matches_torch():
    """clip_grad_norm_(10) + AdamW(betas=(0.9, 0.999), per-group decay) vs torch.optim.AdamW, 4 steps (tolerance: torch's lerp /
    addcdiv kernels may contract to FMAs, ours are compiled with -ffp-contract=off)"""
    from yolov10_3d_amd.optim import FusedAdamW
    torch.manual_seed(0)
    shapes = [(64, 32, 3, 3), (64,), (17,), (300000,), (5, 7), (16385,)]
    ref = [torch.nn.Parameter(torch.randn(*s, device=DEV)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    groups = lambda ps: [{"params": ps[:3], "weight_decay": 5e-4}, {"params": ps[3:], "weight_decay": 0.0}]
    o_ref = torch.optim.AdamW(groups(ref), lr=0.002, betas=(0.9, 0.999), weight_decay=0.0)
    o_my = FusedAdamW(groups(mine), lr=0.002, betas=(0.9, 0.999), weight_decay=0.0)
    for step in range(4):
        grads = [torch.randn_like(p) * (3.0 if step == 1 else 0.01) for p in ref]  # step 1 is clipped
        for p, q, g in zip(ref, mine, grads):
            p.grad, q.grad = g.clone(), g.clone()
        n_ref = torch.nn.utils.clip_grad_norm_(ref, max_norm=10.0)
        o_ref.step()
        o_my.step(max_norm=10.0)
        check(o_my.last_norm[0:1], n_ref.reshape(1), 1e-5, "grad norm")
        for p, q in zip(ref, mine):
            check(q, p, 2e-6, f"param after AdamW step {step}")


def test_model_ema_matches_reference_rule():
    """ModelEMA.update in one launch vs the reference's loop over state_dict keys (utils/torch_utils.py:431-443) on the N-3D model:
    bit-exact, including the aliased one-to-one head tensors that the key walk updates twice per call"""
    import copy, math
    from yolov10_3d_amd.optim import ModelEMA
    torch.manual_seed(0)
    model = y3d.YOLOv10_3DDetectionModel("yolov10n_3D.yaml", verbose=False).to(DEV).train()
    ema = ModelEMA(model, decay=0.9999, tau=2000)
    ref = copy.deepcopy(ema.ema)
    n_keys = len(ref.state_dict())
    n_unique = len({v.data_ptr() for v in ref.state_dict().values()})
    assert n_keys > n_unique, "the head aliases must show up as duplicate state_dict keys"
    for upd in range(1, 4):
        with torch.no_grad():
            for p in model.parameters():  # the model moves between updates
                p.add_(torch.randn_like(p) * 0.01)
            for b in model.buffers():
                if b.dtype.is_floating_point:
                    b.add_(torch.rand_like(b) * 0.01)
        ema.update(model)
        d = 0.9999 * (1 - math.exp(-upd / 2000))
        msd = model.state_dict()
        with torch.no_grad():
            for k, v in ref.state_dict().items():
                if v.dtype.is_floating_point:
                    v *= d
                    v += (1 - d) * msd[k].detach()
        for (k, a), (_, b) in zip(ema.ema.state_dict().items(), ref.state_dict().items()):
            assert torch.equal(a, b), f"EMA tensor {k} differs after update {upd}: max |d| = {(a.float() - b.float()).abs().max().item():.3e}"


def test_fused_sgd_matches_torch():
    """clip_grad_norm_(10) + SGD(nesterov, weight decay) as 3 multi-tensor launches vs torch's own implementation, 3 steps"""
    from yolov10_3d_amd.optim import FusedSGD
    torch.manual_seed(0)
    shapes = [(64, 32, 3, 3), (64,), (17,), (300000,), (5, 7), (16385,)]
    ref = [torch.nn.Parameter(torch.randn(*s, device=DEV)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    o_ref = torch.optim.SGD([{"params": ref[:3], "weight_decay": 5e-4}, {"params": ref[3:], "weight_decay": 0.0}], lr=0.01, momentum=0.937, nesterov=True)
    o_my = FusedSGD([{"params": mine[:3], "weight_decay": 5e-4}, {"params": mine[3:], "weight_decay": 0.0}], lr=0.01, momentum=0.937, nesterov=True)
    for step in range(3):
        grads = [torch.randn_like(p) * (3.0 if step == 1 else 0.01) for p in ref]  # step 1 is clipped, the others are not
        for i, (p, q, g) in enumerate(zip(ref, mine, grads)):
            skip = i == 4 and step != 2  # a parameter without a gradient is skipped (unused detect level), and may get one later
            p.grad = None if skip else g.clone()
            q.grad = None if skip else g.clone()
        n_ref = torch.nn.utils.clip_grad_norm_(ref, max_norm=10.0)
        o_ref.step()
        o_my.step(max_norm=10.0)
        check(o_my.last_norm[0:1], n_ref.reshape(1), 1e-5, "grad norm")
        for p, q in zip(ref, mine):
            check(q, p, 1e-6, f"param after step {step}")


def test_graphed_eval_forward_replays_the_eager_forward():
    """yolov10-3d_amd/graph.py: the eval forward + postprocess captured into one hipGraph gives the eager forward's outputs bit for bit,
    on new inputs too (copied into the static buffer), and re-captures after the parameters moved (a training step bumps
    ops.PARAM_EPOCH: the capture holds the addresses of the packed / folded eval constants of the OLD state)"""
    from bench import synth_batch
    from yolov10_3d_amd import ops
    from yolov10_3d_amd.graph import GraphedForward
    from yolov10_3d_amd.loss import v10_3Dpostprocess
    from yolov10_3d_amd.optim import build_optimizer
    y3d.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(2)
    model = y3d.YOLOv10_3DDetectionModel("yolov10n_3D.yaml").to(DEV)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            with torch.no_grad():
                m.running_var.uniform_(0.5, 1.5)
                m.running_mean.uniform_(-0.2, 0.2)
    model.eval()
    imgs = [synth_batch(2, 256, 256, s, DEV)["img"] for s in (1, 2)]

    def fwd(img):
        y, maps = model(img)["one2one"]
        return (y,) + v10_3Dpostprocess(y.permute(0, 2, 1), 50, 3)

    with torch.no_grad():
        eager = [[t.clone() for t in fwd(im)] for im in imgs]
    g = GraphedForward(fwd, imgs[0])
    # inside the capture the three detection levels were recorded as parallel branches (modules.v10Detect3d.inference_forward_feat)
    assert ops.EVAL_LEVEL_STREAMS and len(ops._LEVEL_STREAMS) >= model.model[-1].nl
    for im, ref in zip(imgs + imgs[:1], eager + eager[:1]):
        out = g(im)
        torch.cuda.synchronize()
        for a, b in zip(out, ref):
            assert torch.equal(a, b)
    assert g.captures == 1
    # parameters move -> the next call re-captures and matches the eager forward of the NEW state
    model.train()
    opt = build_optimizer(model)
    batch = synth_batch(2, 256, 256, 5, DEV)
    loss, _ = model(batch)
    loss.backward()
    opt.step(max_norm=10.0)
    model.eval()
    with torch.no_grad():
        ref = [t.clone() for t in fwd(imgs[1])]
    out = g(imgs[1])
    torch.cuda.synchronize()
    assert g.captures == 2
    for a, b in zip(out, ref):
        assert torch.equal(a, b)
    assert not torch.equal(ref[0], eager[1][0]), "the training step did not change the eval output: the test would not see a stale capture"


def test_graphed_train_step_matches_eager_steps():
    """graph.GraphedTrainStep: three training steps replayed from one hipGraph leave the model where three eager steps leave it (same
    kernels on the same data in the same order: bit for bit) - parameters, BatchNorm running statistics, momentum buffers and the
    loss items of every step; batches with different box counts go through the padded static label buffers"""
    from bench import synth_batch
    from yolov10_3d_amd.graph import GraphedTrainStep
    from yolov10_3d_amd.optim import build_optimizer
    y3d.set_compute_dtype(torch.bfloat16)
    batches = [synth_batch(2, 256, 256, 20 + j, DEV) for j in range(3)]
    assert len({b["batch_idx"].shape[0] for b in batches}) > 1, "the batches should differ in their box counts"
    res = {}
    for mode in ("eager", "graph"):
        torch.manual_seed(3)
        model = y3d.YOLOv10_3DDetectionModel("yolov10n_3D.yaml").to(DEV).train()
        opt = build_optimizer(model, lr=0.01)
        model.model[-1].restack()
        items = []
        if mode == "eager":
            for b in batches:
                loss, it = model(b)
                loss.backward()
                opt.step(max_norm=10.0)
                opt.zero_grad()
                items.append(it.float().cpu())
        else:
            state0 = {k: v.clone() for k, v in model.state_dict().items()}
            step = GraphedTrainStep(model, opt, batches[0])
            # constructing the step does not train (round-3 advisor finding): its warm-up steps are undone - parameters, BatchNorm
            # statistics and counters, momentum, step counts are where they were
            for k, v in model.state_dict().items():
                assert torch.equal(v, state0[k]), f"GraphedTrainStep's constructor changed {k}"
            assert float(opt._state["flat"].abs().max()) == 0 and float(opt._state["norm_clip"].abs().max()) == 0 and opt._steps == 0
            for b in batches:
                loss, it = step(b)
                items.append(it.float().cpu().clone())
        torch.cuda.synchronize()
        res[mode] = (items, {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}, opt._state["flat"].cpu().clone())
    for a, b in zip(res["eager"][0], res["graph"][0]):
        assert torch.equal(a, b), (a, b)
    for k, v in res["eager"][1].items():
        assert torch.equal(v, res["graph"][1][k]), f"state {k} differs after three steps"
    assert torch.equal(res["eager"][2], res["graph"][2]), "momentum buffers differ"


def test_graphed_train_step_reports_target_overflow_and_survives_eager_steps():
    """round-3 advisor findings on graph.GraphedTrainStep: (1) an image with more boxes than the assigner kernels take (64) is reported
    after a replay as in the eager loop (the captured pad_targets has no read-back of its own); (2) eager optimizer steps between
    replays (the fallback for such a batch) do not disturb the graph's gradient pointer table"""
    from bench import synth_batch
    from yolov10_3d_amd import loss as PL
    from yolov10_3d_amd.graph import GraphedTrainStep
    from yolov10_3d_amd.optim import build_optimizer
    y3d.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(3)
    model = y3d.YOLOv10_3DDetectionModel("yolov10n_3D.yaml").to(DEV).train()
    opt = build_optimizer(model, lr=0.01)
    b0, b1 = synth_batch(2, 256, 256, 31, DEV), synth_batch(2, 256, 256, 32, DEV)
    step = GraphedTrainStep(model, opt, b0, label_capacity=256)
    step(b0)
    PL.check_target_overflow(wait=True)
    # six eager steps: more than the depth of the optimizer's rotating pointer buffers
    for _ in range(6):
        loss, _ = model(b1)
        loss.backward()
        opt.step(max_norm=10.0)
        opt.zero_grad()
    ref_model = copy.deepcopy(model)
    ref_opt = build_optimizer(ref_model, lr=0.01)
    ref_model.model[-1].restack()
    loss, it_ref = ref_model(b0)
    loss.backward()
    ref_opt.load_state_dict(opt.state_dict())
    ref_opt.step(max_norm=10.0)
    _, it = step(b0)
    torch.cuda.synchronize()
    assert torch.equal(it, it_ref)
    for (k, a), b in zip(model.state_dict().items(), ref_model.state_dict().values()):
        assert torch.equal(a, b), f"replay after eager steps: {k} differs from an eager step on the same state"
    PL.check_target_overflow(wait=True)
    # an image with 70 boxes
    big = {k: v.clone() if torch.is_tensor(v) else v for k, v in b0.items()}
    n0 = int((b0["batch_idx"] == 0).sum())
    reps = 70 - n0
    first = int((b0["batch_idx"] == 0).nonzero()[0])
    for k in step.box_keys:
        big[k] = torch.cat((b0[k], b0[k][first:first + 1].expand(reps, *b0[k].shape[1:])), 0)
    step(big)
    with pytest.raises(y3d.Y3DError, match="70 ground-truth boxes"):
        PL.check_target_overflow(wait=True)


@pytest.mark.parametrize("mode", ["box_only", "kps_only_l2", "both_l2", "both_unconstrained", "kps_only_unconstrained_top1"])
def test_tal3d_hip_non_default_modes_vs_oracle(mode):
    """cfg/default.yaml:116-119 - `tal_2d`, `tal_3d`, `kps_dist_metric`, `constrain_anchors` - through DDDetectionLoss on the HIP assigner
    against the oracle, which tests/test_oracle_golden.py::test_tal3d_non_default_modes pins to the reference's TaskAlignedAssigner3d in
    exactly these modes: fg_mask / target_gt_idx bit-exact, target scores 1e-4 (round 2 raised NotImplementedError here)"""
    from types import SimpleNamespace
    from yolov10_3d_amd import loss as PL
    import bench
    u2, u3, l2, con, topk = {"box_only": (1, 0, 0, 1, 8), "kps_only_l2": (0, 1, 1, 1, 8), "both_l2": (1, 1, 1, 1, 8),
                             "both_unconstrained": (1, 1, 0, 0, 8), "kps_only_unconstrained_top1": (0, 1, 0, 0, 1)}[mode]
    y3d.set_compute_dtype(torch.float32)
    torch.manual_seed(6)
    B, nc = 3, 3
    shapes, strides = [(32, 32), (16, 16), (8, 8)], [8.0, 16.0, 32.0]
    batch = bench.synth_batch(B, 256, 256, seed=12, device=DEV)
    maps = []
    for (h, w) in shapes:
        t = torch.randn(B, 38, h, w, device=DEV)
        t[:, 0:3] -= 2.0
        t[:, 5:7] = 2 + 4 * torch.rand(B, 2, h, w, device=DEV)
        t[:, 36] = 10 + 30 * torch.rand(B, h, w, device=DEV)
        maps.append(y3d.ops._dense_any(t, torch.float32))
    head = SimpleNamespace(stride=torch.tensor(strides), nc=nc, no=38)
    hyp = dict(y3d.tasks.DEFAULT_HYP, tal_2d=bool(u2), tal_3d=bool(u3), kps_dist_metric="l2" if l2 else "l1", constrain_anchors=bool(con))
    crit = PL.DDDetectionLoss(SimpleNamespace(model=[head], args=SimpleNamespace(**hyp)), tal_topk=topk)
    crit(maps, batch)
    fg, gi, ts = crit.last_assignment
    # oracle on the decoded predictions (CPU)
    cat = PL._flatten_maps(maps).cpu()
    sc, o2d, s2d, o3d, s3d, hd, dep, dun = cat.split((nc, 2, 2, 2, 3, 24, 1, 1), -1)
    anc, st = RS.make_anchors(shapes, strides)
    cb = {k: v.cpu() for k, v in batch.items()}
    rows = torch.cat([cb[k].float().view(cb[k].shape[0], -1) for k in
                      ("batch_idx", "cls", "bboxes", "center_2d", "size_2d", "center_3d", "size_3d", "depth", "heading_bin", "heading_res")], 1)
    gpad = RS.pad_targets(rows, B, 17, torch.tensor([256.0, 256.0, 256.0, 256.0]))
    gts = gpad.split((1, 4, 2, 2, 2, 3, 1, 1, 1), 2)
    mask_gt = (gts[1].sum(2, keepdim=True) > 0).float()
    cen = anc + o2d
    pb = torch.cat((cen - s2d / 2, cen + s2d / 2), -1) * st
    targets, fg_o, gi_o = RS.tal3d(sc.sigmoid(), pb, torch.cat((o3d, s3d, hd, dep, dun), -1), anc * st, gts, mask_gt, st, cb["calib"], cb["mean_sizes"],
                                   topk, nc, use_2d=bool(u2), use_3d=bool(u3), kps_dist="l2" if l2 else "l1", constrain=bool(con))
    assert int(fg_o.sum()) > 0
    mism = int((fg.cpu().bool() != fg_o).sum()) + int((gi.cpu().long() != gi_o).sum())
    assert mism == 0, f"{mode}: {mism} anchors differ"
    check(ts, targets[1], 1e-4, "target_scores")
    with pytest.raises(RuntimeError):
        PL.DDDetectionLoss(SimpleNamespace(model=[head], args=SimpleNamespace(**dict(hyp, tal_2d=False, tal_3d=False))), tal_topk=topk)


# ---------------------------------------------------------------------------------------------------------
# drop-in boundary on the device (INTEGRATION.md form A): what the reference's trainer / validator do to rebound modules
# ---------------------------------------------------------------------------------------------------------
def _reference_walk(model, x):
    """the reference's BaseModel._predict_once (nn/tasks.py:117-146): a plain walk of the rows - no placement, no private state"""
    y = []
    for m in model.model:
        if m.f != -1:
            x = y[m.f] if isinstance(m.f, int) else [x if j == -1 else y[j] for j in m.f]
        x = m(x)
        y.append(x if m.i in model.save else None)
    return x


def _reference_fuse(model):
    """what the reference's BaseModel.fuse() (nn/tasks.py:177-205) does to every Conv it finds: `m.conv = fuse_conv_and_bn(m.conv, m.bn)`
    (utils/torch_utils.py:171-198: a NEW nn.Conv2d with folded weight + bias), `delattr(m, "bn")`, `m.forward = m.forward_fuse`"""
    for m in model.modules():
        if isinstance(m, M.Conv) and hasattr(m, "bn"):
            w, b = y3d.tasks.fuse_conv_and_bn(m.conv.weight.detach(), m.bn.weight.detach(), m.bn.bias.detach(), m.bn.running_mean, m.bn.running_var, m.bn.eps)
            c = m.conv
            f = torch.nn.Conv2d(c.in_channels, c.out_channels, c.kernel_size, c.stride, c.padding, groups=c.groups, bias=True).requires_grad_(False).to(w.device)
            f.weight.copy_(w)
            f.bias.copy_(b)
            m.conv = f
            delattr(m, "bn")
            m.forward = m.forward_fuse
    return model


@pytest.mark.parametrize("name", ["yolov10n_3D.yaml", "yolov10n.yaml", "yolov10m_3D.yaml"])
def test_reference_style_walk_and_fuse_match_eval(name):
    """(1) the rows walked the reference's way give the package's own forward bit for bit (training maps and eval output: placement is an
    optimisation, not a semantic); (2) a model folded the reference's way (BaseModel.fuse) runs on `forward_fuse` and reproduces the
    unfolded eval forward (fp32 mode; folding moves the BatchNorm scale into the weights, so equal up to rounding)"""
    from bench import synth_batch
    y3d.set_compute_dtype(torch.float32)
    try:
        torch.manual_seed(0)
        model = y3d.DetectionModel(name).to(DEV).train()
        img = synth_batch(2, 256, 256, 1, DEV)["img"]
        state = copy.deepcopy(model.state_dict())
        own = model(img)
        model.load_state_dict(state)
        ref = _reference_walk(model, img)
        for a, b in zip(own["one2many"] + own["one2one"], ref["one2many"] + ref["one2one"]):
            assert torch.equal(a, b)
        model.eval()
        with torch.no_grad():
            y_own = model(img)["one2one"][0].clone()
            y_ref = _reference_walk(model, img)["one2one"][0]
            assert torch.equal(y_own, y_ref)
            folded = _reference_fuse(copy.deepcopy(model))
            assert sum(isinstance(m, torch.nn.BatchNorm2d) for m in folded.modules()) == 0
            y_f = _reference_walk(folded, img)["one2one"][0]
        if hasattr(model.model[-1], "dep"):
            check_sparse_eval(y_f, y_own, model.model[-1].nc, 1e-3)
        else:
            check(y_f, y_own, 1e-3, "folded 2D eval output")
    finally:
        y3d.set_compute_dtype(torch.bfloat16)


def test_fused_conv_forward_with_residuals_matches_eval():
    """forward_fuse keeps forward's (x, res, res_mode) signature: Bottleneck / RepVGGDW call their Convs with residual arguments"""
    y3d.set_compute_dtype(torch.float32)
    try:
        torch.manual_seed(3)
        for mod in (M.Bottleneck(32, 32, True, 1, (3, 3), 1.0), M.RepVGGDW(32), M.CIB(32, 32, True, 1.0, True), M.PSA(128, 128)):
            mod = mod.to(DEV).eval()
            for b in mod.modules():
                if isinstance(b, torch.nn.BatchNorm2d):
                    b.running_mean.normal_(0, 0.2)
                    b.running_var.uniform_(0.5, 1.5)
                    b.weight.data.uniform_(0.5, 1.5)
                    b.bias.data.normal_(0, 0.2)
            c = 128 if isinstance(mod, M.PSA) else 32
            x = torch.randn(2, c, 20, 20, device=DEV)
            with torch.no_grad():
                y0 = mod(x)
                y1 = _reference_fuse(copy.deepcopy(mod))(x)
            check(y1, y0, 2e-4, type(mod).__name__)
    finally:
        y3d.set_compute_dtype(torch.bfloat16)


def test_autocast_and_gradscaler_leave_the_step_unchanged():
    """engine/trainer.py:395-408 runs the step inside torch.autocast and backs a GradScaler-scaled loss: the modules ignore autocast
    (raw launches in the package's compute dtype) and every backward kernel is linear in the incoming gradient, so after
    `scaler.unscale_` the gradients are those of the plain step (a power-of-two scale: equal up to values that leave bf16's normal range)"""
    from bench import synth_batch
    torch.manual_seed(0)
    cfg = _tiny_cfg("yolov10s_3D.yaml", **TINY)
    model = y3d.YOLOv10_3DDetectionModel(cfg).to(DEV).train()
    batch = synth_batch(2, 128, 128, 1, DEV)
    state = copy.deepcopy(model.state_dict())
    loss0, items0 = model(batch)
    loss0.backward()
    g0 = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    model.zero_grad(set_to_none=True)
    model.load_state_dict(state)
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss1, items1 = model(batch)
    assert torch.equal(items0, items1) and torch.equal(loss0, loss1)
    scaler.scale(loss1).backward()
    scaler.unscale_(opt)
    g1 = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    assert set(g0) == set(g1)
    worst = max(l2_rel(g1[k], g0[k]) for k in g0 if float(g0[k].abs().max()) > 0)
    assert worst < 2e-3, worst
    scaler.step(opt)
    scaler.update()
    assert scaler.get_scale() == 1024.0  # no inf / NaN found: the scale is kept


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_depthwise_eval_epilogue_is_the_two_pass_form_bit_for_bit(dtype):
    """csrc/dwconv.hip: y3d_dwconv2d_fwd_affine (eval: depth-wise conv + folded BatchNorm + SiLU + residual in ONE launch) applies the
    affine pass to the conv result rounded as the pre-BatchNorm tensor would have been stored, so it must equal y3d_dwconv2d_fwd +
    y3d_bn_act_fwd (`ops.DW_EVAL_FUSED = False`) exactly: depth-wise Conv k3 / k3 s2 without activation (SCDown, block.py:824) / k7,
    RepVGGDW (pre-activation residual, block.py:711) and a whole C2fCIB with `lk=True` (block.py:737-768)"""
    from yolov10_3d_amd import ops
    y3d.set_compute_dtype(dtype)
    torch.manual_seed(5)
    mods = [M.Conv(64, 64, 3, 1, g=64), M.Conv(96, 96, 3, 2, g=96, act=False), M.Conv(64, 64, 7, 1, g=64, act=False), M.RepVGGDW(64),
            M.C2fCIB(64, 64, 1, True, True), M.SCDown(64, 128, 3, 2)]
    shapes = [(2, 64, 20, 24), (2, 96, 21, 19), (2, 64, 20, 20), (2, 64, 12, 20), (2, 64, 16, 16), (2, 64, 16, 24)]
    old = ops.DW_EVAL_FUSED
    try:
        for m, shp in zip(mods, shapes):
            m = m.to(DEV)
            for b in m.modules():
                if isinstance(b, torch.nn.BatchNorm2d):
                    with torch.no_grad():
                        b.running_mean.uniform_(-0.3, 0.3)
                        b.running_var.uniform_(0.5, 1.5)
                        b.weight.uniform_(0.5, 1.5)
                        b.bias.uniform_(-0.2, 0.2)
            m.eval()
            x = torch.randn(*shp, device=DEV)
            outs = []
            for flag in (True, False):
                ops.DW_EVAL_FUSED = flag
                with torch.no_grad():
                    outs.append(m(x).float().clone())
            assert torch.equal(outs[0], outs[1]), f"{type(m).__name__} {shp}: fused depth-wise epilogue differs from conv + bn_act_fwd"
            assert torch.isfinite(outs[0]).all() and outs[0].abs().max() > 0
    finally:
        ops.DW_EVAL_FUSED = old
        y3d.set_compute_dtype(torch.float32)
