"""GPU parity of the BENCHMARK path: the bf16 kernels that carry bench.py's number, at the bench shapes,
through the C ABI.

* exact tests: with small-integer operands (x, w, dy in {0, +-1}, sparse) every product, every fp32 partial sum and every bf16
  store is exact, so the bf16 kernels (conv3x3_wide forward / data gradient, conv3x3_wgrad_tile, their BatchNorm partial sums)
  must reproduce an fp32 host convolution BIT FOR BIT — at B=32 80x80 128->2048 / 16 x (128->128), the other head levels, odd
  batches and channel counts that are not multiples of the 128-channel tile;
* the 16-bit yardstick: HIP-bf16's distance to the fp32 golden vectors is held to 1.5 x the distance of the REFERENCE's own
  autocast(bfloat16) run on the same weights and inputs (tests/golden/autocast_bf16.npz, oracle/make_golden_bf16.py);
* the stacked head sees torch-side weight writes (load_state_dict, torch.optim): ADVICE round 1, high;
* the assignment does not depend on the padded-row bound (`n_used`)."""
import copy

import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden

pytestmark = pytest.mark.gpu

import yolov10_3d_amd as y3d  # noqa: E402
from yolov10_3d_amd import modules as M  # noqa: E402
from yolov10_3d_amd import ops  # noqa: E402
from yolov10_3d_amd._lib import BF16  # noqa: E402

DEV = "cuda"


def _sparse_int(shape, gen, density=0.125):
    """values in {0, +-1}: P(+1) = P(-1) = density / 2"""
    u = torch.rand(shape, generator=gen)
    return (u < density / 2).float() - (u > 1 - density / 2).float()


def _conv_abi(x, w, g, dy):
    """bf16 3x3 s1 p1 conv through the C ABI exactly as ops._cba_forward / _conv_backward drive it.
    x: (B, Cin, H, W) fp32 CPU, w: (Cout, Cin/g, 3, 3) fp32 CPU, dy: (B, Cout, H, W) fp32 CPU -> y, bn partial sums, dx, dW (CPU)"""
    L, st, dt = y3d.lib(), ops.stream(), BF16
    k, s, p = 3, 1, 1
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    bf = torch.bfloat16
    xin = ops.nhwc_empty(B, Cin, H, W, bf, DEV)
    xin.copy_(x.to(DEV))
    dyd = ops.nhwc_empty(B, Cout, H, W, bf, DEV)
    dyd.copy_(dy.to(DEV))
    wd = w.to(DEV).contiguous()
    sb, sh, sw = ops.s3(xin)
    # forward + BatchNorm partial sums
    wp = torch.empty(Cout * 9 * (Cin // g), dtype=bf, device=DEV)
    L.pack_weight_fwd(dt, wd.data_ptr(), wp.data_ptr(), Cout, Cin // g, Cin // g, k, k, st)
    nblk = L.conv2d_stat_rows(dt, B, H, W, Cin, Cout, g, k, k, s, p)
    part = torch.full((nblk, Cout, 2), float("nan"), dtype=torch.float32, device=DEV)
    y = ops.nhwc_empty(B, Cout, H, W, bf, DEV)
    L.conv2d_fwd(dt, xin.data_ptr(), sb, sh, sw, B, H, W, Cin, wp.data_ptr(), None, y.data_ptr(), Cout, H, W, Cout, g, k, k, s, p, part.data_ptr(), st)
    # data gradient
    kp = L.conv_kpad(dt, 9 * (Cout // g))
    wpd = torch.empty(Cin * kp, dtype=bf, device=DEV)
    L.pack_weight_dgrad(dt, wd.data_ptr(), wpd.data_ptr(), Cout, Cin // g, g, k, k, st)
    dx = ops.nhwc_empty(B, Cin, H, W, bf, DEV)
    dsb, dsh, dsw = ops.s3(dyd)
    L.conv2d_bwd_data(dt, dyd.data_ptr(), dsb, dsh, dsw, B, H, W, Cout, wpd.data_ptr(), dx.data_ptr(), Cin, H, W, Cin, g, k, k, s, p, st)
    # weight gradient
    ns = L.conv2d_wgrad_plan(dt, B, H, W, Cin, Cout, g, k, k, s, p)
    slab = torch.empty(ns * Cout * 9 * (Cin // g), dtype=torch.float32, device=DEV)
    dW = torch.empty_like(wd)
    L.conv2d_bwd_weight(dt, xin.data_ptr(), sb, sh, sw, B, H, W, Cin, Cin, dyd.data_ptr(), Cout, H, W, Cout, g, k, k, s, p, slab.data_ptr(), ns,
                        dW.data_ptr(), 0, st)
    torch.cuda.synchronize()
    return y.float().cpu(), part.double().sum(0).cpu(), dx.float().cpu(), dW.cpu()


BENCH_SHAPES = [
    # Cin, Cout, groups, H, W, B
    (128, 2048, 1, 80, 80, 32),     # bench: fused head layer 1 at P3 (8 branches x 2 head sets, Cin -> 16 x 128)
    (2048, 2048, 16, 80, 80, 32),   # bench: fused head layer 2 at P3 = the roofline kernel's launch (16 groups of 128 -> 128)
    (256, 2048, 1, 40, 40, 32),     # P4 layer 1
    (2048, 2048, 16, 40, 40, 32),   # P4 layer 2
    (512, 2048, 1, 20, 20, 32),     # P5 layer 1 (20 % 8 != 0: not the persistent kernel; the tile kernel / generic path)
    (2048, 2048, 16, 20, 20, 5),    # P5 layer 2, odd batch
    (128, 2048, 1, 80, 80, 3),      # odd batch against the 2-image tile
    (192, 320, 1, 24, 24, 3),       # Cn % 128 = 64, Cg = 192 (X-model widths)
    (96, 80, 1, 16, 40, 5),         # Cn % 128 = 80 (multiple of 16 only), ragged x tiles
    (1152, 1152, 2, 40, 40, 2),     # groups of 576 -> 576 (M-3D fused layer widths)
    # (the P5 comment above predates the ragged-height path: 20 rows run as three 8-row tiles of the persistent kernel)
    # narrow body layers: conv3x3_small.hip (all nine taps' weights resident, forward + flipped-tap data gradient)
    (64, 64, 1, 80, 80, 4),         # the 64 -> 64 Bottleneck convs at 80x80
    (32, 32, 1, 160, 160, 2),       # 32 -> 32 at 160x160 (64-byte K slabs)
    (32, 64, 1, 21, 27, 3),         # ragged tiles both ways
    (64, 24, 1, 24, 40, 1),         # output channels not a multiple of 16
    (64, 48, 1, 16, 16, 2),         # three output-channel tiles; fewer tiles than partial-sum rows
    # round 3: input channel counts that are not a multiple of the 32-channel K slab (X widths) on the persistent kernel
    (80, 80, 1, 40, 40, 3),         # 80 -> 80: 2.5 slabs
    (72, 160, 1, 24, 24, 2),        # 72 = 2 slabs + 8 channels, two output-channel tiles
    (160, 160, 2, 16, 32, 2),       # two groups of 80 -> 80
    (48, 96, 1, 16, 16, 4),         # 1.5 slabs, Cout > 64 (the narrow kernel does not take it)
    # round 4: conv3x3_flat.hip (tiles of 512 consecutive positions of the padded flat space) - maps whose rows / columns the rectangular
    # tiles do not divide, with enough channel tiles for y3d_tile_height to pick it (the P4 / P5 rows above go through it too)
    (64, 2048, 1, 23, 37, 7),       # odd height and width, odd batch: tiles span rows and images, the last tile is ragged
    (96, 1536, 3, 20, 20, 9),       # three groups of 32 -> 512: one K slab + 0 (Cg = 32 is below the kernel's 40: stays on the generic path)
    (80, 1024, 1, 20, 20, 11),      # 2.5 K slabs on the flat kernel
    (256, 640, 2, 12, 44, 5),       # W + 2 = 46: row wraps inside a 16-position DMA piece; Cn = 320 = 2.5 channel tiles per group
]


@pytest.mark.parametrize("shape", BENCH_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_conv_bench_shapes_bit_exact_on_integer_operands(shape):
    Cin, Cout, g, H, W, B = shape
    gen = torch.Generator().manual_seed(Cin + Cout + H + B)
    x = _sparse_int((B, Cin, H, W), gen)
    w = _sparse_int((Cout, Cin // g, 3, 3), gen)
    dy = _sparse_int((B, Cout, H, W), gen)
    y3d.set_compute_dtype(torch.bfloat16)
    y, stats, dx, dW = _conv_abi(x, w, g, dy)
    # fp32 host reference (every value is a small integer: exact in any summation order)
    y_ref = F.conv2d(x, w, padding=1, groups=g)
    assert float(y_ref.abs().max()) <= 256, "operands too dense for exact bf16 stores"
    assert torch.equal(y, y_ref), f"forward: {int((y != y_ref).sum())} of {y.numel()} outputs differ"
    s_ref = torch.stack((y_ref.double().sum((0, 2, 3)), (y_ref.double() ** 2).sum((0, 2, 3))), 1)
    assert torch.equal(stats, s_ref), "BatchNorm partial sums (sum, sum of squares) differ"
    del y, y_ref
    dx_ref = torch.nn.grad.conv2d_input(x.shape, w, dy, padding=1, groups=g)
    assert float(dx_ref.abs().max()) <= 256
    assert torch.equal(dx, dx_ref), f"data gradient: {int((dx != dx_ref).sum())} of {dx.numel()} differ"
    del dx, dx_ref
    dW_ref = torch.nn.grad.conv2d_weight(x, w.shape, dy, padding=1, groups=g)
    assert float(dW_ref.abs().max()) < 2 ** 24
    assert torch.equal(dW, dW_ref), f"weight gradient: {int((dW != dW_ref).sum())} of {dW.numel()} differ"


def test_conv_roofline_shape_real_valued_within_derived_bound():
    """the integer tests above are exact only on {0, +-1} operands.  The roofline launch's geometry (16 groups
    of 128 -> 128 @80x80; B = 4 keeps the host reference affordable, the persistent kernels still run 800 tiles) with REAL-valued
    bf16 operands against an fp64 host convolution, each output held to a bound derived from the arithmetic: fp32 accumulation of K
    exact products (<= K * 2^-24 * sum|x||w|, doubled for the split partial sums) plus ONE bf16 rounding of the result (2^-8
    relative, round to nearest even); the weight gradient has no output rounding (fp32)."""
    Cin = Cout = 2048
    g, H, W, B = 16, 80, 80, 4
    gen = torch.Generator().manual_seed(11)
    bfr = lambda t: t.bfloat16().float()
    pre = torch.randn(B, Cin, H, W, generator=gen)
    x = bfr(pre * torch.sigmoid(pre))                       # a post-SiLU activation
    w = bfr(torch.randn(Cout, Cin // g, 3, 3, generator=gen) * 0.03)
    dy = bfr(torch.randn(B, Cout, H, W, generator=gen) * 0.1)
    y3d.set_compute_dtype(torch.bfloat16)
    y, _, dx, dW = _conv_abi(x, w, g, dy)
    K = 9 * (Cin // g)
    eps32 = 2.0 ** -24
    for name, got, ref, mag, kk, rounded in (
            ("forward", y, F.conv2d(x.double(), w.double(), padding=1, groups=g), F.conv2d(x.abs(), w.abs(), padding=1, groups=g), K, True),
            ("data gradient", dx, torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), padding=1, groups=g),
             torch.nn.grad.conv2d_input(x.shape, w.abs(), dy.abs(), padding=1, groups=g), K, True),
            ("weight gradient", dW, torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), padding=1, groups=g),
             torch.nn.grad.conv2d_weight(x.abs(), w.shape, dy.abs(), padding=1, groups=g), B * H * W, False)):
        bound = 2.0 * kk * eps32 * mag.double() + (2.0 ** -8 * ref.abs() if rounded else 0.0) + 1e-30
        excess = ((got.double() - ref).abs() / bound).max()
        assert float(excess) <= 1.0, f"{name}: an output is {float(excess):.2f} x its derived error bound away from the fp64 convolution"
        del ref, mag


# ---------------------------------------------------------------------------------------------------------
# bf16 against the reference's own 16-bit (autocast) path
# ---------------------------------------------------------------------------------------------------------
def _l2(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp(min=1e-12))


AC_MODS = {
    "conv_k1": lambda: M.Conv(16, 24, 1, 1), "conv_k3s1": lambda: M.Conv(16, 24, 3, 1), "conv_k3s2": lambda: M.Conv(16, 24, 3, 2),
    "conv_dw3": lambda: M.Conv(16, 16, 3, 1, None, 16), "conv_dw7": lambda: M.Conv(16, 16, 7, 1, 3, 16, 1, False),
    "c2f_shortcut": lambda: M.C2f(32, 32, 2, True), "c2f_neck": lambda: M.C2f(48, 32, 1, False),
    "c2fcib_lk": lambda: M.C2fCIB(32, 32, 1, True, True), "c2fcib": lambda: M.C2fCIB(32, 32, 1, True, False),
    "scdown": lambda: M.SCDown(16, 32, 3, 2), "sppf": lambda: M.SPPF(32, 32, 5),
    "psa_1head": lambda: M.PSA(128, 128), "psa_2head": lambda: M.PSA(256, 256),
}
RATIO = 1.5


@pytest.mark.parametrize("name", sorted(AC_MODS))
def test_bf16_module_no_worse_than_reference_autocast(name):
    """distance to the fp32 golden: HIP bf16 mode <= 1.5 x the reference module under torch.autocast(bfloat16)"""
    g = load_golden(name)
    ac = load_golden("autocast_bf16")[name]
    y3d.set_compute_dtype(torch.bfloat16)
    mod = AC_MODS[name]()
    sd = {k[len("model.0."):]: v for k, v in g["state"].items()}
    mod.load_state_dict(sd, strict=False)
    for m in mod.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eps, m.momentum = 1e-3, 0.03
    mod = mod.to(DEV).train()
    x = g["x"].to(DEV).requires_grad_(True)
    y = mod(x)
    (y.float() * g["r"].to(DEV)).sum().backward()
    named = dict(mod.named_parameters())
    rows = [("y_train", _l2(y, g["y_train"]), _l2(ac["y_train"], g["y_train"])), ("dx", _l2(x.grad, g["dx"]), _l2(ac["dx"], g["dx"]))]
    # parameter gradients: one aggregate distance over the fixture's gradients (single tensors of 16-24 values are too noisy)
    num_h = num_r = den = 0.0
    for k, gv in g["grads"].items():
        ah = named[k[len("model.0."):]].grad.detach().float().cpu()
        ar = ac[f"grads/{k}"]
        num_h += float((ah - gv).pow(2).sum())
        num_r += float((ar - gv).pow(2).sum())
        den += float(gv.pow(2).sum())
    rows.append(("param grads", (num_h / den) ** 0.5, (num_r / den) ** 0.5))
    report = ", ".join(f"{what}: hip {h:.2e} / ref-autocast {r:.2e}" for what, h, r in rows)
    print(f"[bf16 yardstick] {name}: {report}")
    for what, h, r in rows:
        assert h <= RATIO * r, f"{name} {what}: HIP bf16 is {h:.3e} from the fp32 golden, the reference's autocast run {r:.3e} ({report})"


def test_bf16_e2e_tiny3d_no_worse_than_reference_autocast():
    """the tiny end-to-end 3D model: head maps (smooth part) and loss items against the fp32 golden, HIP bf16 vs reference autocast"""
    from test_hip_modules import TINY, _tiny_cfg
    from oracle import restate as RS
    g = load_golden("e2e_tiny3d_s")
    ac = load_golden("autocast_bf16")["e2e_tiny3d_s"]
    cfg = _tiny_cfg("yolov10s_3D.yaml", **TINY, kernel_size_1=3, kernel_size_2=3, num_scales=3)
    spec = RS.build_spec(cfg)
    with torch.no_grad():
        ref = RS.forward(spec, {k: v.clone() for k, v in g["state"].items()}, g["img"], True)  # fp32 head maps (the oracle is pinned to the golden)
    y3d.set_compute_dtype(torch.bfloat16)
    model = y3d.YOLOv10_3DDetectionModel(cfg)
    model.load(g["state"])
    model = model.to(DEV).train()
    batch = {k: v.to(DEV) for k, v in g["batch"].items()}
    batch["img"] = g["img"].to(DEV)
    with torch.no_grad():
        out = model.predict(batch["img"])
    model.load(g["state"])
    num_h = num_r = den = 0.0
    for key, acs in (("one2many", "o2m"), ("one2one", "o2o")):
        for j, (a, b) in enumerate(zip(out[key], ref[key])):
            r = ac[f"{acs}/{j}"]
            num_h += float((a.float().cpu() - b).pow(2).sum())
            num_r += float((r - b).pow(2).sum())
            den += float(b.pow(2).sum())
    dh, dr = (num_h / den) ** 0.5, (num_r / den) ** 0.5
    loss, items = model(batch)
    loss.backward()
    ih = float((items.float().cpu() - g["items"]).abs().max() / g["items"].abs().max())
    ir = float((ac["items"] - g["items"]).abs().max() / g["items"].abs().max())
    print(f"[bf16 yardstick] e2e_tiny3d_s: head maps hip {dh:.2e} / ref-autocast {dr:.2e}; loss items (max-norm) hip {ih:.2e} / ref-autocast {ir:.2e}")
    assert dh <= RATIO * dr, f"head maps: HIP bf16 {dh:.3e} vs reference autocast {dr:.3e}"
    # the assigner is discrete: a flipped positive moves an item by O(10 %) in the reference's own 16-bit run too (it is 17 % off here)
    # (round 2 allowed max(1.5 x reference, 0.3); the floor is dropped where 1.5 x the reference's own distance is the tighter bound)
    assert ih <= RATIO * ir, f"loss items: HIP bf16 {ih:.3e} vs reference autocast {ir:.3e}"


# ---------------------------------------------------------------------------------------------------------
# stacked head weights follow torch-side writers
# ---------------------------------------------------------------------------------------------------------
def _tiny_model(seed):
    from test_hip_modules import TINY, _tiny_cfg
    cfg = _tiny_cfg("yolov10s_3D.yaml", **TINY, kernel_size_1=3, kernel_size_2=3, num_scales=3)
    torch.manual_seed(seed)
    m = y3d.YOLOv10_3DDetectionModel(cfg)
    g = torch.Generator().manual_seed(seed)
    for b in m.modules():
        if isinstance(b, torch.nn.BatchNorm2d):
            with torch.no_grad():
                b.weight.copy_(1 + 0.2 * (torch.rand(b.weight.shape, generator=g) - 0.5))
                b.bias.copy_(0.2 * (torch.rand(b.bias.shape, generator=g) - 0.5))
                b.running_mean.copy_(0.1 * (torch.rand(b.bias.shape, generator=g) - 0.5))
                b.running_var.copy_(1 + 0.5 * torch.rand(b.bias.shape, generator=g))
    return m


def _tiny_batch(S=64, B=2):
    from bench import synth_batch
    return synth_batch(B, S, S, 3, DEV)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_stacked_head_sees_load_state_dict(dtype):
    """forward, load_state_dict(other weights), forward == a fresh model with those weights — training and eval mode.  The stacked
    head convs (ops.StackedConvs) cache packed weights; `load_state_dict` copies into the per-branch Parameters, whose version
    counters the flat views do not share."""
    y3d.set_compute_dtype(dtype)
    batch = _tiny_batch()
    sB = {k: v.clone() for k, v in _tiny_model(1).state_dict().items()}
    sC = {k: v.clone() for k, v in _tiny_model(2).state_dict().items()}
    mA = _tiny_model(0).to(DEV).train()
    lossA, _ = mA(batch)
    lossA.backward()  # forward and backward packs registered
    mA.zero_grad(set_to_none=True)
    mA.load_state_dict(sB)
    _, itemsA = mA(batch)
    mB = _tiny_model(1).to(DEV).train()
    _, itemsB = mB(batch)
    assert torch.equal(itemsA, itemsB), f"stale packed weights after load_state_dict (train): {itemsA} vs {itemsB}"
    # eval: the sparse head's stacked regression convs cache packed + folded weights
    img = torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(4)).to(DEV)
    mA.eval()
    with torch.no_grad():
        mA(img)
        mA.load_state_dict(sC)
        yA = mA(img)["one2one"][0]
        mC = _tiny_model(2).to(DEV).eval()
        yC = mC(img)["one2one"][0]
    assert torch.equal(yA, yC), "stale packed weights after load_state_dict (eval)"


def test_stacked_head_under_torch_optim_matches_fused_sgd():
    """3 training steps with torch.optim.SGD (the reference trainer's optimizer) == 3 steps with the fused HIP optimizer: the
    stacked head must repack after every torch-side update"""
    from yolov10_3d_amd.optim import FusedSGD
    y3d.set_compute_dtype(torch.float32)
    batch = _tiny_batch()
    ma, mb = _tiny_model(0).to(DEV).train(), _tiny_model(0).to(DEV).train()
    ma.model[-1].restack()
    mb.model[-1].restack()
    kw = dict(lr=0.01, momentum=0.9, nesterov=True, weight_decay=5e-4)
    oa = torch.optim.SGD(ma.parameters(), **kw)
    ob = FusedSGD(list(mb.parameters()), **kw)
    for step in range(3):
        la, ia = ma(batch)
        lb, ib = mb(batch)
        # stale packed weights would move the items by O(1); the two optimizers' different rounding alone, amplified by the discrete
        # assignment over three steps, has been seen at 3e-4 relative
        assert torch.allclose(ia, ib, rtol=1e-3, atol=1e-5), f"step {step}: loss items diverged: {ia} vs {ib}"
        la.backward()
        lb.backward()
        oa.step()
        ob.step(max_norm=None)
        oa.zero_grad(set_to_none=True)
        ob.zero_grad(set_to_none=True)
    for (k, p), q in zip(ma.named_parameters(), mb.parameters()):
        # unclipped steps on a loss of O(1000): per-tensor max-norm (a stale stacked weight shows as O(1) relative differences)
        e = float((p.detach() - q.detach()).abs().max() / p.detach().abs().max().clamp(min=1e-3))
        assert e <= 1e-3, f"{k} differs after 3 steps: {e:.3e} relative"


def test_assignment_does_not_depend_on_padded_row_bound():
    """y3d_tal3d_assign / y3d_loss3d with the device-side row bound from y3d_pad_targets (rows >= n_used skipped) and without it
    (all 64 capacity rows walked): identical integer outputs, target scores and loss items"""
    from types import SimpleNamespace
    from bench import synth_batch
    from yolov10_3d_amd import loss as PL
    y3d.set_compute_dtype(torch.float32)
    torch.manual_seed(7)
    B, nc = 4, 3
    shapes, strides = [(40, 40), (20, 20), (10, 10)], [8.0, 16.0, 32.0]
    batch = synth_batch(B, 320, 320, seed=13, device=DEV)
    maps = []
    for (h, w) in shapes:
        t = torch.randn(B, 38, h, w, device=DEV)
        t[:, 0:3] -= 2.0
        t[:, 5:7] = 2 + 4 * torch.rand(B, 2, h, w, device=DEV)
        t[:, 36] = 10 + 30 * torch.rand(B, h, w, device=DEV)
        maps.append(ops._dense_any(t, torch.float32))
    head = SimpleNamespace(stride=torch.tensor(strides), nc=nc, no=38)
    model = SimpleNamespace(model=[head], args=SimpleNamespace(**y3d.tasks.DEFAULT_HYP))
    for topk in (10, 1):
        crit = PL.DDDetectionLoss(model, tal_topk=topk)
        gt, n_used = crit.targets(batch, B, 40, 40, torch.device(DEV))
        assert gt.shape == (B, 64, 17) and 1 <= int(n_used) <= 8
        outs = []
        for nu in (n_used, None):
            _, items = crit(maps, batch, targets=(gt, nu))
            outs.append((items.clone(), *[t.clone() for t in crit.last_assignment]))
        for a, b in zip(outs[0], outs[1]):
            assert torch.equal(a, b)
        assert int(outs[0][1].sum()) > 0


# ---------------------------------------------------------------------------------------------------------
# f2 / f3: the callers either side of the hot path, against the reference's own outputs
# ---------------------------------------------------------------------------------------------------------
def test_kitti_image_aug_matches_reference_samples():
    """mirror / mixup blend / affine crop / 255 of KITTIDataset.__getitem__ (Pillow arithmetic) on the device: every pixel of the
    reference's own samples, as the float tensor AND as uint8 NHWC straight into the stem"""
    import numpy as np
    import os
    from conftest import GOLDEN
    from yolov10_3d_amd import kitti
    z = np.load(os.path.join(GOLDEN, "kitti_aug.npz"))
    src = [torch.from_numpy(a).to(DEV) for a in z["src"]]
    W, H = [int(v) for v in z["resolution"]]
    n = int(z["n"])
    idx = [[int(v) for v in z[f"s{i}/index"]] for i in range(n)]
    imgs = [src[i0] for i0, _, _, _ in idx]
    partners = [src[i1] if mixed else None for _, i1, _, mixed in idx]
    flips = [bool(f) for _, _, f, _ in idx]
    tinv = [z[f"s{i}/trans_inv"] for i in range(n)]
    out = kitti.augment_images(imgs, partners, flips, tinv, (W, H), "float")
    ref8 = torch.from_numpy(np.stack([z[f"s{i}/img8"] for i in range(n)]))
    assert torch.equal(out.cpu(), ref8.float() / 255.0)  # the reference's tensor, bit for bit
    out8 = kitti.augment_images(imgs, partners, flips, tinv, (W, H), "uint8")
    assert torch.equal(out8.cpu(), ref8.permute(0, 2, 3, 1))
    # no-mixup batch takes the null-partner path
    keep = [i for i in range(n) if not idx[i][3]]
    o2 = kitti.augment_images([imgs[i] for i in keep], [None] * len(keep), [flips[i] for i in keep], [tinv[i] for i in keep], (W, H), "uint8")
    assert torch.equal(o2.cpu(), ref8.permute(0, 2, 3, 1)[keep])
    # the uint8 form feeds the stem: same activations as the float tensor
    y3d.set_compute_dtype(torch.float32)
    torch.manual_seed(0)
    stem = M.Conv(3, 16, 3, 2).to(DEV).train()
    a = stem(out)
    b = stem(out8.permute(0, 3, 1, 2))
    assert torch.equal(a, b)


def test_kde_depth_fusion_matches_reference():
    """the validator's one-to-many depth fusion (scikit-learn KernelDensity per detection on the host in the reference) as one launch"""
    import numpy as np
    import os
    from conftest import GOLDEN
    from yolov10_3d_amd import kitti
    g = np.load(os.path.join(GOLDEN, "kde_fusion.npz"))
    O, Mm, F = (torch.from_numpy(g[k]) for k in ("predsO", "predsM", "fused"))
    out = kitti.aggregate_o2m_preds(O.to(DEV), Mm.to(DEV)).cpu()
    fused = F[..., -4] != O[..., -4]
    assert int(fused.sum()) >= 50
    other = torch.ones(37, dtype=torch.bool)
    other[-4] = False
    assert torch.equal(out[..., other], F[..., other])           # everything but the depth is passed through
    assert torch.equal(out[..., -4][~fused], F[..., -4][~fused])  # no votes: depth unchanged
    # the fused depth is one of 500 float32 proposals: the same proposal as the reference (libm exp / log may differ in the last
    # bit from numpy's, which can move a tied arg-max by one proposal in rare cases)
    same = out[..., -4] == F[..., -4]
    assert float(same.float().mean()) >= 0.98, f"{int((~same).sum())} depths differ"
    rng = (O[..., -4].max() - O[..., -4].min()).item()
    assert float((out[..., -4] - F[..., -4]).abs().max()) <= rng / 100


# ---------------------------------------------------------------------------------------------------------
# fp8 weights (BASELINE configs[4])
# ---------------------------------------------------------------------------------------------------------
def test_fp8w_quantizer_bit_exact_vs_torch_float8():
    """codes / per-channel power-of-two scales / effective weights of csrc/fp8w.hip against torch.float8_e4m3fn (round to nearest even)"""
    from oracle import restate as RS
    L = y3d.lib()
    torch.manual_seed(3)
    ws = [torch.randn(24, 16, 3, 3) * 0.1, torch.randn(7, 40, 1, 1) * 3.0, torch.randn(130, 9) * 1e-4, torch.zeros(4, 8, 1, 1)]
    ws[0][3] = 0                      # an all-zero row
    ws[1][2, :5, 0, 0] = torch.tensor([448.0, -448.0, 449.0, 1e-9, -0.0])
    ws[2][5] *= 1e6                   # rows of very different magnitude
    dev = [w.to(DEV).contiguous() for w in ws]
    outs = [(torch.empty_like(w), torch.empty(w.shape, dtype=torch.uint8, device=DEV), torch.empty(w.shape[0], device=DEV)) for w in dev]
    desc, rb, r = [], [], 0
    for w, (e, c, s) in zip(dev, outs):
        desc += [w.data_ptr(), e.data_ptr(), c.data_ptr(), s.data_ptr(), w.shape[0], w[0].numel()]
        rb.append(r)
        r += w.shape[0]
    # (the tables must outlive the launch: a temporary freed right after .data_ptr() hands its block to the next allocation)
    t_desc, t_rb = torch.tensor(desc, dtype=torch.int64, device=DEV), torch.tensor(rb, dtype=torch.int32, device=DEV)
    L.mt_fp8w_quantize(t_desc.data_ptr(), t_rb.data_ptr(), len(ws), r, ops.stream())
    for w, (e, c, s) in zip(ws, outs):
        c_ref, s_ref, e_ref = RS.fp8w_quantize(w)
        assert torch.equal(s.cpu(), s_ref)
        cc = c.cpu()
        same = cc == c_ref
        # +0 / -0 codes of values that round to zero may differ in the sign bit only
        assert bool((same | (((cc | c_ref) & 0x7f) == 0)).all()), f"{int((~same).sum())} codes differ"
        assert torch.equal(e.cpu(), e_ref)
        assert torch.equal(e.cpu().bfloat16().float(), e.cpu()), "effective weights must be exact in bf16"
        back = torch.empty_like(e)
        L.fp8w_dequantize(c.data_ptr(), s.data_ptr(), back.data_ptr(), w.shape[0], w[0].numel(), ops.stream())
        assert torch.equal(back, e)


def test_fp8w_model_matches_oracle_on_quantized_weights():
    """the tiny 3D model with `set_weight_quant("fp8")` (exact-fp32 compute) == the CPU oracle run on the fp8-valued weights: loss
    items and gradients within the 1e-3 of the fp32 parity mode (the gradient of a quantised weight goes to its fp32 master)"""
    from test_hip_modules import TINY, _tiny_cfg, check, grad_floor
    from oracle import restate as RS
    g = load_golden("e2e_tiny3d_s")
    cfg = _tiny_cfg("yolov10s_3D.yaml", **TINY, kernel_size_1=3, kernel_size_2=3, num_scales=3)
    spec = RS.build_spec(cfg)
    st = RS.fp8w_state(spec, {k: v.clone() for k, v in g["state"].items()})
    nq = sum(1 for k in st if not torch.equal(st[k], g["state"][k]))
    assert nq >= 60, f"only {nq} weights were quantised"
    for v in st.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    for k in list(st):
        if "running" in k or "num_batches" in k:
            st[k] = st[k].detach()
    batch = {k: v for k, v in g["batch"].items()}
    batch["img"] = g["img"]
    preds = RS.forward(spec, st, g["img"], True)
    loss_o, items_o, _ = RS.loss3d(preds, batch, RS.model_strides(spec), 3)
    loss_o.backward()
    y3d.set_compute_dtype(torch.float32)
    y3d.set_weight_quant("fp8")
    try:
        model = y3d.YOLOv10_3DDetectionModel(cfg)
        model.load(g["state"])
        model = model.to(DEV).train()
        dbatch = {k: v.to(DEV) for k, v in batch.items()}
        loss, items = model(dbatch)
        loss.backward()
        check(items, items_o.detach(), 1e-3, "loss items (fp8 weights)")
        # not the unquantised result: the mode is on
        assert float((items.cpu() - g["items"]).abs().max()) > 1e-3 * float(g["items"].abs().max())
        named = dict(model.named_parameters())
        grads = {k: st[k].grad for k in g["grads"] if st[k].grad is not None}
        gf = grad_floor(grads)
        for k, gv in grads.items():
            check(named[k].grad, gv, 5e-3, f"grad {k}", gf)
        # torch-side weight update is seen (shadow + packs follow the master's version), 1-byte state dict round trip
        with torch.no_grad():
            for p in model.parameters():
                p.mul_(1.25)
        loss2, items2 = model(dbatch)
        sd8 = y3d.tasks.fp8_state_dict(model)
        assert len(sd8) >= 60 and all(c.dtype == torch.uint8 for c, _ in sd8.values())
        y3d.set_weight_quant(None)
        m3 = y3d.tasks.load_fp8_state_dict(y3d.YOLOv10_3DDetectionModel(cfg).to(DEV), sd8)
        n3 = dict(m3.named_parameters())
        for k, (c, s_) in sd8.items():
            ref = RS.fp8w_quantize(named[k].detach().cpu())[2]
            assert torch.equal(n3[k].detach().cpu(), ref), k
        assert torch.isfinite(items2).all() and not torch.allclose(items2, items)
    finally:
        y3d.set_weight_quant(None)


def test_fp8w_bf16_conv_products_are_exact():
    """bf16 kernels on fp8-valued weights: with integer activations every product and partial sum is exact, so the bf16 conv on
    w_eff equals the fp32 host conv on w_eff bit for bit - the 'fp8 weight x bf16 activation' MFMA the format stands for"""
    from oracle import restate as RS
    gen = torch.Generator().manual_seed(9)
    B, Cin, Cout, H, W = 4, 128, 256, 16, 16
    x = _sparse_int((B, Cin, H, W), gen, 0.25)
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) * 0.05
    _, scale, w_eff = RS.fp8w_quantize(w)
    y3d.set_compute_dtype(torch.bfloat16)
    y, _, _, _ = _conv_abi(x, w_eff, 1, _sparse_int((B, Cout, H, W), gen))
    y_ref = F.conv2d(x, w_eff, padding=1)
    # the fp32 sums of <= 1152 products with 4 significant bits each at one exponent per row are exact; the bf16 store rounds once
    assert torch.equal(y, y_ref.bfloat16().float())


@pytest.mark.parametrize("mid,hw,B", [(128, (16, 16), 2), (128, (10, 13), 3), (64, (8, 20), 2)])
def test_proj_bn_mfma_backward_matches_materialised_path(mid, hw, B):
    """second head layer's BatchNorm backward recomputing dz = dout . W on the matrix cores (proj_bn_mfma.hip) against the path that
    materialises dz (projg_bwd_data + bn_act_bwd_reduce / _apply): same input / weight / BatchNorm gradients up to bf16 rounding of
    the dz tensor the old path stores; pixel counts that are not multiples of the 64-pixel step included"""
    from test_hip_modules import check
    y3d.set_compute_dtype(torch.bfloat16)
    chan = {k + "_c": mid for k in ("cls", "o2d", "s2d", "o3d", "s3d", "hd", "dep", "dep_un")}
    torch.manual_seed(11)
    head = M.v10Detect3d(3, (64,), False, chan, False, False, False, False, 1, False, False, 3, 3)
    head.stride = torch.tensor([8.0])
    head.bias_init()
    with torch.no_grad():
        for p_ in head.o2m_heads.parameters():
            p_.add_(0.02 * torch.randn_like(p_))
    head = head.to(DEV).train()
    x = torch.randn(B, 64, *hw, device=DEV)
    r = [torch.randn(B, 38, *hw, device=DEV) for _ in range(2)]
    res = {}
    for flag in (True, False):
        ops.PROJ_BN_MFMA = flag
        try:
            head.zero_grad(set_to_none=True)
            xi = x.clone().requires_grad_(True)
            out = head([xi])
            assert "_y3d_maps" in out
            ((out["one2one"][0].float() * r[0]).sum() + (out["one2many"][0].float() * r[1]).sum()).backward()
            res[flag] = (xi.grad.float().clone(), {k: v.grad.float().clone() for k, v in head.named_parameters() if v.grad is not None},
                         out["one2one"][0].float().clone(), out["one2many"][0].float().clone())
        finally:
            ops.PROJ_BN_MFMA = True
    # forward projections on the matrix cores (bf16 weights) vs the VALU form (fp32 weights): bf16 rounding of 24 x 128 weights
    check(res[True][2], res[False][2], 2e-2, "one2one map")
    check(res[True][3], res[False][3], 2e-2, "one2many map")
    check(res[True][0], res[False][0], 2e-2, "dx")
    assert set(res[True][1]) == set(res[False][1])
    big = max(float(v.abs().max()) for v in res[False][1].values())
    for k, v in res[False][1].items():
        check(res[True][1][k], v, 3e-2, f"grad {k}", floor=1e-3 * big)


# ---- streaming 1x1 kernel (conv1x1_stream.hip) ---------------------------------------------------------------------------------
def _conv1x1_abi(x, w, dy, bias=None, affine=None, stream=True):
    """bf16 1x1 conv forward (+ BN partial sums, or bias, or folded affine + SiLU) and data gradient through the C ABI.
    x: (B, Ctot, H, W) NHWC device tensor whose channel slice [lo, lo + Cin) is the input (a channel-slice VIEW, as C2f feeds its cv2)"""
    L, st, dt = y3d.lib(), ops.stream(), BF16
    (xt, lo, Cin), Cout = x, w.shape[0]
    B, _, H, W = xt.shape
    bf = torch.bfloat16
    xin = xt[:, lo:lo + Cin]
    sb, sh, sw = ops.s3(xin)
    wd = w.to(DEV).contiguous()
    wp = torch.empty(Cout * Cin, dtype=bf, device=DEV)
    L.pack_weight_fwd(dt, wd.data_ptr(), wp.data_ptr(), Cout, Cin, Cin, 1, 1, st)
    y = ops.nhwc_empty(B, Cout, H, W, bf, DEV)
    old = L.set_stream1x1(1 if stream else 0)
    try:
        nblk = L.conv2d_stat_rows(dt, B, H, W, Cin, Cout, 1, 1, 1, 1, 0)  # depends on the kernel the entry point will pick
        part = torch.full((nblk, Cout, 2), float("nan"), dtype=torch.float32, device=DEV)
        if affine is not None:
            sc, sf = (t.to(DEV).contiguous() for t in affine)
            L.conv2d_fwd_affine(dt, xin.data_ptr(), sb, sh, sw, B, H, W, Cin, wp.data_ptr(), sc.data_ptr(), sf.data_ptr(), 1, y.data_ptr(), Cout, H, W,
                                Cout, 1, 1, 1, 1, 0, st)
        else:
            bd = bias.to(DEV).contiguous() if bias is not None else None
            L.conv2d_fwd(dt, xin.data_ptr(), sb, sh, sw, B, H, W, Cin, wp.data_ptr(), bd.data_ptr() if bd is not None else None, y.data_ptr(), Cout, H, W,
                         Cout, 1, 1, 1, 1, 0, None if bd is not None else part.data_ptr(), st)
        kp = L.conv_kpad(dt, Cout)
        wpd = torch.empty(Cin * kp, dtype=bf, device=DEV)
        L.pack_weight_dgrad(dt, wd.data_ptr(), wpd.data_ptr(), Cout, Cin, 1, 1, 1, st)
        dx = torch.zeros_like(xt)
        dxv = dx[:, lo:lo + Cin]
        dsb, dsh, dsw = ops.s3(dy)
        L.conv2d_bwd_data(dt, dy.data_ptr(), dsb, dsh, dsw, B, H, W, Cout, wpd.data_ptr(), dxv.data_ptr(), dxv.stride(3), H, W, Cin, 1, 1, 1, 1, 0, st)
        ns = L.conv2d_wgrad_plan(dt, B, H, W, Cin, Cout, 1, 1, 1, 1, 0)
        slab = torch.full((ns * Cout * Cin,), float("nan"), dtype=torch.float32, device=DEV)
        dW = torch.empty_like(wd)
        L.conv2d_bwd_weight(dt, xin.data_ptr(), sb, sh, sw, B, H, W, Cin, Cin, dy.data_ptr(), dsw, H, W, Cout, 1, 1, 1, 1, 0, slab.data_ptr(), ns,
                            dW.data_ptr(), 0, st)
        torch.cuda.synchronize()
    finally:
        L.set_stream1x1(old)
    _conv1x1_abi.dW = dW
    return y, part, dx


STREAM_SHAPES = [
    # B, H, W, Ctot, lo, Cin, Cout
    (2, 40, 40, 64, 0, 64, 64),        # one K chunk, BN = 64
    (3, 17, 23, 192, 64, 96, 128),     # ragged pixel count, channel-slice view, K = 96 (half-filled last chunk), BN = 128
    (2, 24, 24, 32, 0, 32, 32),        # K = 32: one MFMA step per stage
    (1, 33, 31, 384, 0, 384, 136),     # N = 136: two channel tiles, the second nearly empty
    (2, 20, 20, 256, 0, 256, 256),     # BN = 128 x 2 tiles (the weights of a 256-wide tile do not leave room for the ring)
    (5, 16, 16, 128, 0, 128, 256),     # BN = 256
    (2, 20, 20, 512, 0, 512, 520),     # weights too large to stay resident: their chunks travel in the ring (3 channel tiles of 256)
    (16, 40, 40, 256, 0, 256, 256),    # two resident tiles would stream the pixels twice: one streamed 256-channel tile
    (2, 24, 24, 48, 0, 48, 96),        # K = 48 (M-model widths): the second MFMA step of the only K chunk is half zeros
    (2, 20, 20, 336, 8, 144, 288),     # K = 144 = 2 chunks + 16, a slice view at channel offset 8, streamed weights (N > 256)
    (16, 64, 64, 256, 0, 256, 264),    # a large streamed-weight GEMM: 512 pixel tiles x two channel tiles
]


@pytest.mark.parametrize("shape", STREAM_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_stream1x1_matches_generic_kernel_and_exact_integer_conv(shape):
    """The streaming kernel accumulates in the generic kernel's K order: y and dx must be BIT-IDENTICAL to it, and on small-integer
    operands (fp32 accumulation exact, bf16 stores exact) equal to fp32 conv2d on the host; the BatchNorm partial sums are exact
    integers there too, whatever the grouping into rows."""
    B, H, W, Ctot, lo, Cin, Cout = shape
    g = torch.Generator().manual_seed(sum(shape))
    xf = _sparse_int((B, Ctot, H, W), g, 0.25)
    wf = _sparse_int((Cout, Cin, 1, 1), g, 0.25)
    dyf = _sparse_int((B, Cout, H, W), g, 0.25)
    xt = ops.nhwc_empty(B, Ctot, H, W, torch.bfloat16, DEV)
    xt.copy_(xf.to(DEV))
    dy = ops.nhwc_empty(B, Cout, H, W, torch.bfloat16, DEV)
    dy.copy_(dyf.to(DEV))
    L = y3d.lib()
    assert L.set_stream1x1(1) in (0, 1)
    y1, p1, dx1 = _conv1x1_abi((xt, lo, Cin), wf, dy, stream=True)
    dw1 = _conv1x1_abi.dW
    y0, p0, dx0 = _conv1x1_abi((xt, lo, Cin), wf, dy, stream=False)
    dw0 = _conv1x1_abi.dW
    assert torch.equal(y1, y0) and torch.equal(dx1, dx0)
    # weight gradient (wgrad1x1_stream.hip for Cin, Cout > 32): exact integers on these operands, whatever the split-K plan
    dwr = torch.einsum("bohw,bihw->oi", dyf, xf[:, lo:lo + Cin])[:, :, None, None]
    assert torch.equal(dw1.cpu(), dwr) and torch.equal(dw0.cpu(), dwr)
    yr = torch.nn.functional.conv2d(xf[:, lo:lo + Cin], wf)
    assert torch.equal(y1.float().cpu(), yr)
    dxr = torch.nn.functional.conv_transpose2d(dyf, wf)
    assert torch.equal(dx1[:, lo:lo + Cin].float().cpu(), dxr)
    assert float(dx1[:, :lo].abs().sum()) == 0 and float(dx1[:, lo + Cin:].abs().sum()) == 0  # the view's neighbours are untouched
    assert not torch.isnan(p1).any()
    ref = torch.stack([yr.sum((0, 2, 3)), (yr * yr).sum((0, 2, 3))], 1).double()
    assert torch.equal(p1.double().sum(0).cpu(), ref) and torch.equal(p0.double().sum(0).cpu(), ref)


@pytest.mark.parametrize("Cin,Cout", [(128, 72), (512, 512)], ids=["resident", "streamed_weights"])
def test_stream1x1_bias_and_affine_epilogues_match_generic_kernel(Cin, Cout):
    B, H, W = 2, 19, 21
    g = torch.Generator().manual_seed(5)
    xt = ops.nhwc_empty(B, Cin, H, W, torch.bfloat16, DEV)
    xt.copy_(torch.randn(B, Cin, H, W, generator=g).to(DEV))
    dy = ops.nhwc_empty(B, Cout, H, W, torch.bfloat16, DEV)
    dy.copy_(torch.randn(B, Cout, H, W, generator=g).to(DEV))
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    bias = torch.randn(Cout, generator=g)
    aff = (1 + 0.1 * torch.randn(Cout, generator=g), torch.randn(Cout, generator=g))
    for kw in ({"bias": bias}, {"affine": aff}):
        y1, _, dx1 = _conv1x1_abi((xt, 0, Cin), w, dy, stream=True, **kw)
        y0, _, dx0 = _conv1x1_abi((xt, 0, Cin), w, dy, stream=False, **kw)
        assert torch.equal(y1, y0) and torch.equal(dx1, dx0), list(kw)


def test_mt_pack_weights_matches_single_tensor_packing():
    """The per-step multi-tensor pack (LDS-staged transposes) against the per-conv packing kernels, both layouts, on the geometries
    the models use: 3x3 / 1x1 / 7x7, grouped, a padded stem (Cg_pad > Cg), a K that needs row padding, chunk boundaries inside rows."""
    L, st = y3d.lib(), ops.stream()
    g = torch.Generator().manual_seed(11)
    geos = [(128, 128, 1, 3), (2048, 128, 1, 3), (2048, 128, 16, 3), (64, 32, 1, 1), (256, 384, 1, 1), (36, 64, 1, 1), (32, 3, 1, 3), (96, 96, 1, 3),
            (64, 64, 1, 7), (24, 128, 1, 1)]
    for mode in (0, 1):
        ents, descs, ct, co = [], [], [], []
        CH = 16384
        for t, (Cout, Cin, G, k) in enumerate(geos):
            Cg, taps = Cin // G, k * k
            if mode == 1 and Cg % 8:
                continue
            w = torch.randn(Cout, Cg, k, k, generator=g).to(DEV)
            if mode == 0:
                Cgp = (Cg + 7) // 8 * 8
                kpad = taps * Cgp
                n, geo = Cout * kpad, (Cout, Cg, Cgp, taps, kpad)
                ref = torch.empty(n, dtype=torch.bfloat16, device=DEV)
                L.pack_weight_fwd(BF16, w.data_ptr(), ref.data_ptr(), Cout, Cg, Cgp, k, k, st)
            else:
                Cn = Cout // G
                kpad = L.conv_kpad(BF16, taps * Cn)
                n, geo = G * Cg * kpad, (G, Cn, Cg, taps, kpad)
                ref = torch.empty(n, dtype=torch.bfloat16, device=DEV)
                L.pack_weight_dgrad(BF16, w.data_ptr(), ref.data_ptr(), Cout, Cg, G, k, k, st)
            dst = torch.full((n,), float("nan"), dtype=torch.bfloat16, device=DEV)
            ti = len(ents)
            ents.append((w, ref, dst))
            descs += [w.data_ptr(), dst.data_ptr(), *geo, mode]
            for j in range((n + CH - 1) // CH):
                ct.append(ti)
                co.append(j)
        t_desc = torch.tensor(descs, dtype=torch.int64, device=DEV)
        t_ct = torch.tensor(ct, dtype=torch.int32, device=DEV)
        t_co = torch.tensor(co, dtype=torch.int32, device=DEV)
        L.mt_pack_weights(BF16, t_desc.data_ptr(), t_ct.data_ptr(), t_co.data_ptr(), len(ct), CH, st)
        torch.cuda.synchronize()
        for i, (w, ref, dst) in enumerate(ents):
            assert torch.equal(dst.view(torch.int16), ref.view(torch.int16)), (mode, i, tuple(w.shape))


@pytest.mark.parametrize("kind", ["c2f", "c2f_shortcut", "sppf", "psa", "c2fcib", "c2fcib_lk"])
def test_concat_placement_is_bit_identical_to_copying(kind):
    """C2f / SPPF producers write straight into their slice of the concat buffer (ops.place); with the switch off every input is copied.
    Same kernels on the same values either way: outputs and gradients must not differ in a single bit.  PSA concatenates a slice of its
    cv1 output with a new tensor: that slice must NOT be taken for an in-place input (its neighbour is still needed by the backward)."""
    from importlib import import_module
    M = import_module("yolov10-3d_amd.modules")
    torch.manual_seed(3)
    mod = {"c2f": lambda: M.C2f(64, 128, 2), "c2f_shortcut": lambda: M.C2f(64, 64, 1, True), "sppf": lambda: M.SPPF(128, 128),
           "psa": lambda: M.PSA(128, 128), "c2fcib": lambda: M.C2fCIB(64, 64, 2, True, False),
           "c2fcib_lk": lambda: M.C2fCIB(64, 128, 1, True, True)}[kind]().to(DEV).train()
    x0 = torch.randn(2, mod.cv1.conv.in_channels, 20, 24, device=DEV)
    res = []
    old = ops.PLACEMENT
    try:
        for flag in (True, False):
            ops.PLACEMENT = flag
            ops.reset_placement()
            for p in mod.parameters():
                p.grad = None
            x = x0.clone().requires_grad_(True)
            y = mod(x)
            (y.float() * torch.linspace(-1, 1, y.numel(), device=DEV).view(y.shape)).sum().backward()
            res.append([y.detach().float().clone(), x.grad.clone()] + [p.grad.clone() for p in mod.parameters()])
    finally:
        ops.PLACEMENT = old
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("name,S", [("yolov10s_3D.yaml", 256), ("yolov10n.yaml", 96)])
def test_cross_layer_concat_placement_is_bit_identical_and_saves_the_copies(name, S):
    """tasks._predict_once: from the second forward of an input geometry on, the rows that produce a Concat's skip member (layers 4, 6,
    10, 13 of the v10 tables: C2f / PSA) write it straight into the concat buffer through their closing 1x1 conv (ops.place_final /
    final_place).  Same kernels on the same values: loss items, every gradient and the eval output must not differ in a single bit
    from the run that copies; and the copies must actually be gone (one y3d_copy2d per skip member less)."""
    import bench
    is3d = "3D" in name
    torch.manual_seed(5)
    model = (y3d.YOLOv10_3DDetectionModel if is3d else y3d.YOLOv10DetectionModel)(name).to(DEV).train()
    batch = bench.synth_batch(2, S, S, 3, DEV, nc=model.yaml["nc"])
    state = {k: v.clone() for k, v in model.state_dict().items()}
    L = ops.lib()
    orig = L.copy2d
    counts, res = {}, {}
    old = ops.PLACEMENT
    try:
        for flag in (True, False):
            ops.PLACEMENT = flag
            model.__dict__.pop("_cat_shapes", None)
            n = []
            L.copy2d = lambda *a: (n.append(a[5] * a[6]), orig(*a))[1]
            outs = []
            for it in range(2):  # the first pass records the concat layouts, the second places
                model.load_state_dict(state)
                model.train()
                for p_ in model.parameters():
                    p_.grad = None
                n.clear()
                loss, items = model(batch)
                loss.backward()
                fwd_copies = len(n)
                outs.append([items.clone()] + [p_.grad.clone() for p_ in model.parameters() if p_.grad is not None])
            model.eval()
            with torch.no_grad():
                ye = model(batch["img"])["one2one"][0].clone()
            res[flag], counts[flag] = outs[1] + [ye], fwd_copies
            for a, b in zip(outs[0], outs[1]):
                assert torch.equal(a, b), "first (recording) and second (placing) pass differ"
    finally:
        L.copy2d = orig
        ops.PLACEMENT = old
        model.__dict__.pop("_cat_shapes", None)
    assert len(res[True]) == len(res[False])
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)
    assert counts[True] <= counts[False] - 4, f"copies per training forward: {counts[True]} with placement, {counts[False]} without"


S2_FWD_SHAPES = [
    # B, H, W, Cin, Cout   (input size; 3x3 stride 2 pad 1)
    (2, 64, 64, 32, 64),      # the 320 -> 160 layer of S in small: whole tiles
    (3, 34, 38, 32, 64),      # ragged output tiles (17 x 19)
    (2, 17, 21, 16, 32),      # odd input: the last output row / column reads one row / column of padding (N widths)
    (1, 40, 24, 24, 64),      # input channels not a multiple of 32: zero-filled chunks of the 64-byte LDS row
    (2, 33, 47, 8, 24),       # one chunk of input channels, output channels not a multiple of 16
    (4, 96, 160, 32, 48),     # more tiles than workgroups per XCD slot; three output-channel tiles
]


@pytest.mark.parametrize("shape", S2_FWD_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_conv3x3s2_forward_narrow_kernel_exact(shape):
    """csrc/conv3x3_small.hip, ST = 2 (parity-plane halo): the stride-2 forward of the first stages through y3d_conv2d_fwd - outputs AND
    BatchNorm partial sums exact on small-integer operands against fp32 conv2d - and through y3d_conv2d_fwd_affine (folded BatchNorm +
    SiLU epilogue) against the same conv followed by the affine map, within one bf16 rounding"""
    L, st, dt = y3d.lib(), ops.stream(), BF16
    B, H, W, Cin, Cout = shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    g = torch.Generator().manual_seed(sum(shape))
    x = _sparse_int((B, Cin, H, W), g, 0.25)
    w = _sparse_int((Cout, Cin, 3, 3), g, 0.25)
    bf = torch.bfloat16
    xin = ops.nhwc_empty(B, Cin, H, W, bf, DEV)
    xin.copy_(x.to(DEV))
    wd = w.to(DEV).contiguous()
    wp = torch.empty(Cout * 9 * Cin, dtype=bf, device=DEV)
    L.pack_weight_fwd(dt, wd.data_ptr(), wp.data_ptr(), Cout, Cin, Cin, 3, 3, st)
    sb, sh, sw = ops.s3(xin)
    nblk = L.conv2d_stat_rows(dt, B, H, W, Cin, Cout, 1, 3, 3, 2, 1)
    part = torch.full((nblk, Cout, 2), float("nan"), dtype=torch.float32, device=DEV)
    y = ops.nhwc_empty(B, Cout, Ho, Wo, bf, DEV)
    y.fill_(7.0)
    L.conv2d_fwd(dt, xin.data_ptr(), sb, sh, sw, B, H, W, Cin, wp.data_ptr(), None, y.data_ptr(), Cout, Ho, Wo, Cout, 1, 3, 3, 2, 1, part.data_ptr(), st)
    ref = F.conv2d(x, w, stride=2, padding=1)
    assert float(ref.abs().max()) <= 256
    assert torch.equal(y.float().cpu(), ref), f"forward: {int((y.float().cpu() != ref).sum())} of {ref.numel()} outputs differ"
    s_ref = torch.stack((ref.double().sum((0, 2, 3)), (ref.double() ** 2).sum((0, 2, 3))), 1)
    assert torch.equal(part.double().sum(0).cpu(), s_ref), "BatchNorm partial sums differ"
    sc = (0.5 + torch.rand(Cout, generator=g)).to(DEV)
    sf = (torch.rand(Cout, generator=g) - 0.5).to(DEV)
    z = ops.nhwc_empty(B, Cout, Ho, Wo, bf, DEV)
    L.conv2d_fwd_affine(dt, xin.data_ptr(), sb, sh, sw, B, H, W, Cin, wp.data_ptr(), sc.data_ptr(), sf.data_ptr(), 1, z.data_ptr(), Cout, Ho, Wo, Cout, 1, 3, 3, 2, 1, st)
    zr = F.silu(ref * sc.cpu().view(1, -1, 1, 1) + sf.cpu().view(1, -1, 1, 1))
    err = (z.float().cpu() - zr).abs().max() / zr.abs().max()
    assert float(err) < 6e-3, f"affine epilogue: {float(err):.2e}"


S2_SHAPES = [
    # B, H, W, Cin, Cout
    (2, 64, 64, 32, 64),      # the 320 -> 160 layer's geometry in small
    (3, 34, 38, 64, 128),     # two K slabs, ragged tiles, two input-channel blocks
    (2, 17, 21, 128, 128),    # odd image: the last dx row / column has no odd-parity taps
    (1, 40, 24, 24, 64),      # input channels not a multiple of 32
]


@pytest.mark.parametrize("shape", S2_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_conv3x3s2_dgrad_resident_tile_matches_generic_and_exact(shape):
    """conv3x3s2_dgrad.hip (all four parity classes from one dy tile) against the generic parity-class kernel (bit-identical: same
    accumulation order) and against fp32 conv_transpose2d on small-integer operands (exact)."""
    L, st, dt = y3d.lib(), ops.stream(), BF16
    B, H, W, Cin, Cout = shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    g = torch.Generator().manual_seed(sum(shape))
    wf = _sparse_int((Cout, Cin, 3, 3), g, 0.25)
    dyf = _sparse_int((B, Cout, Ho, Wo), g, 0.25)
    dy = ops.nhwc_empty(B, Cout, Ho, Wo, torch.bfloat16, DEV)
    dy.copy_(dyf.to(DEV))
    wd = wf.to(DEV).contiguous()
    kp = L.conv_kpad(dt, 9 * Cout)
    wpd = torch.empty(Cin * kp, dtype=torch.bfloat16, device=DEV)
    L.pack_weight_dgrad(dt, wd.data_ptr(), wpd.data_ptr(), Cout, Cin, 1, 3, 3, st)
    dsb, dsh, dsw = ops.s3(dy)
    outs = []
    for flag in (1, 0):
        old = L.set_stream1x1(flag)
        try:
            dx = ops.nhwc_empty(B, Cin, H, W, torch.bfloat16, DEV)
            dx.fill_(float("nan"))
            L.conv2d_bwd_data(dt, dy.data_ptr(), dsb, dsh, dsw, B, Ho, Wo, Cout, wpd.data_ptr(), dx.data_ptr(), Cin, H, W, Cin, 1, 3, 3, 2, 1, st)
            torch.cuda.synchronize()
        finally:
            L.set_stream1x1(old)
        outs.append(dx.float().cpu())
    ref = torch.nn.functional.conv_transpose2d(dyf, wf, stride=2, padding=1, output_padding=(H - 1 - 2 * (Ho - 1), W - 1 - 2 * (Wo - 1)))
    assert ref.shape == outs[0].shape
    assert torch.equal(outs[0], outs[1])
    assert torch.equal(outs[0], ref)


def test_random_conv_geometries_new_kernels_vs_generic():
    """Seeded sweep over conv geometries (1x1 / 3x3, strides 1 / 2, channel counts around the kernels' tile widths, ragged maps, batch 1..5):
    the round-2 kernels (streaming 1x1, narrow 3x3, stride-2 data gradient, their weight gradients) against the generic implicit GEMM
    on small-integer operands, where every path must give the SAME exact integers for y, dx and dW."""
    import os
    import random
    rnd = random.Random(int(os.environ.get("Y3D_SWEEP_SEED", "7")))
    count = int(os.environ.get("Y3D_SWEEP_COUNT", "100"))
    L = y3d.lib()
    y3d.set_compute_dtype(torch.bfloat16)
    chans = [8, 16, 24, 32, 40, 48, 64, 72, 96, 128, 136, 160, 192, 256, 264, 320, 384, 512]
    done = 0
    while done < count:
        k = rnd.choice([1, 1, 3, 3])
        s = 1 if k == 1 else rnd.choice([1, 1, 2])
        cin, cout = rnd.choice(chans), rnd.choice(chans)
        if k == 3 and s == 1 and (cin > 64 or cout > 64):
            continue  # the wide 3x3 kernels have their own tests above
        B, H, W = rnd.randint(1, 5), rnd.randint(6, 41), rnd.randint(8, 45)
        if cin * cout * k * k * H * W * B > 3e9:
            continue
        done += 1
        p = k // 2
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        g = torch.Generator().manual_seed(done)
        xf, wf, dyf = _sparse_int((B, cin, H, W), g, 0.25), _sparse_int((cout, cin, k, k), g, 0.25), _sparse_int((B, cout, Ho, Wo), g, 0.25)
        ref_y = F.conv2d(xf, wf, stride=s, padding=p)
        if float(ref_y.abs().max()) > 256:
            continue
        outs = []
        for flag in (1, 0):
            old = L.set_stream1x1(flag)
            try:
                st, dt, bf = ops.stream(), BF16, torch.bfloat16
                xin = ops.nhwc_empty(B, cin, H, W, bf, DEV)
                xin.copy_(xf.to(DEV))
                dyd = ops.nhwc_empty(B, cout, Ho, Wo, bf, DEV)
                dyd.copy_(dyf.to(DEV))
                wd = wf.to(DEV).contiguous()
                sb, sh, sw = ops.s3(xin)
                wp = torch.empty(cout * k * k * cin, dtype=bf, device=DEV)
                L.pack_weight_fwd(dt, wd.data_ptr(), wp.data_ptr(), cout, cin, cin, k, k, st)
                nblk = L.conv2d_stat_rows(dt, B, H, W, cin, cout, 1, k, k, s, p)
                part = torch.full((nblk, cout, 2), float("nan"), dtype=torch.float32, device=DEV)
                y = ops.nhwc_empty(B, cout, Ho, Wo, bf, DEV)
                L.conv2d_fwd(dt, xin.data_ptr(), sb, sh, sw, B, H, W, cin, wp.data_ptr(), None, y.data_ptr(), cout, Ho, Wo, cout, 1, k, k, s, p, part.data_ptr(), st)
                kp = L.conv_kpad(dt, k * k * cout)
                wpd = torch.empty(cin * kp, dtype=bf, device=DEV)
                L.pack_weight_dgrad(dt, wd.data_ptr(), wpd.data_ptr(), cout, cin, 1, k, k, st)
                dx = ops.nhwc_empty(B, cin, H, W, bf, DEV)
                dsb, dsh, dsw = ops.s3(dyd)
                L.conv2d_bwd_data(dt, dyd.data_ptr(), dsb, dsh, dsw, B, Ho, Wo, cout, wpd.data_ptr(), dx.data_ptr(), cin, H, W, cin, 1, k, k, s, p, st)
                ns = L.conv2d_wgrad_plan(dt, B, H, W, cin, cout, 1, k, k, s, p)
                slab = torch.full((ns * cout * k * k * cin,), float("nan"), dtype=torch.float32, device=DEV)
                dW = torch.empty_like(wd)
                L.conv2d_bwd_weight(dt, xin.data_ptr(), sb, sh, sw, B, H, W, cin, cin, dyd.data_ptr(), cout, Ho, Wo, cout, 1, k, k, s, p, slab.data_ptr(), ns,
                                    dW.data_ptr(), 0, st)
                torch.cuda.synchronize()
                outs.append((y.float().cpu(), part.double().sum(0).cpu(), dx.float().cpu(), dW.cpu()))
            finally:
                L.set_stream1x1(old)
        tag = f"k{k} s{s} {cin}->{cout} B{B} {H}x{W}"
        ref_dx = torch.nn.grad.conv2d_input(xf.shape, wf, dyf, stride=s, padding=p)
        ref_dw = torch.nn.grad.conv2d_weight(xf, wf.shape, dyf, stride=s, padding=p)
        ref_st = torch.stack((ref_y.double().sum((0, 2, 3)), (ref_y.double() ** 2).sum((0, 2, 3))), 1)
        for (yy, stt, dxx, dww), which in zip(outs, ("new kernels", "generic")):
            assert torch.equal(yy, ref_y), f"{tag}: y ({which})"
            assert torch.equal(stt, ref_st), f"{tag}: BatchNorm partial sums ({which})"
            assert torch.equal(dxx, ref_dx), f"{tag}: dx ({which})"
            assert torch.equal(dww, ref_dw), f"{tag}: dW ({which})"


# ---------------------------------------------------------------------------------------------------------
# round 3: every BASELINE configuration produces a bench line; skipped steps; target overflow
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("argv,what", [
    (["--model", "yolov10l.yaml", "--imgsz", "640", "--batch", "2"], "configs[3] model (L, 2D head)"),
    (["--model", "yolov10x_3D.yaml", "--weights", "fp8", "--batch", "2"], "configs[4] model (X + 3D head, fp8 weights)"),
    (["--model", "yolov10m_3D.yaml", "--batch", "2"], "configs[2] model (M + 3D head: non-uniform branches, 1x1 second layer)"),
], ids=["l2d", "x3d_fp8", "m3d"])
def test_bench_main_prints_a_line_for_every_baseline_model(argv, what, capsys):
    """bench.py crashed on the 2D yamls after warm-up (its roofline key assumed the stacked 3D head); no test ran
    bench.main() past argument parsing.  One step + one warm-up of each non-default BASELINE model through bench.main itself."""
    import json
    import bench
    try:
        bench.main(argv + ["--steps", "1", "--warmup", "1", "--infer-steps", "1", "--no-cpu-baseline"])
    finally:
        y3d.set_weight_quant(None)
        y3d.set_compute_dtype(torch.bfloat16)
    line = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["metric"] == "train_images_per_sec" and out["value"] > 0 and out["infer_images_per_sec"] > 0, what
    assert all(v == v and abs(v) < 1e6 for v in out["loss_items"]), out["loss_items"]
    roof = out["roofline"]
    assert roof is not None and roof["bound"] == "mfma" and 0 < roof["frac"] < 1 and roof["launches_timed"] >= 1, roof
    assert out["steps_skipped_nonfinite"] == 0
    # both training legs ran (launch by launch, and replayed from one hipGraph); the headline is the faster one and says which
    assert out["train_images_per_sec_eager"] > 0 and out["train_images_per_sec_graph"] > 0, "the hipGraph leg did not run"
    assert out["train_mode"] in ("eager", "hipgraph")
    assert out["infer_images_per_sec_val_batch"] > 0 and out["infer_val_batch"] == 2 * out["config"]["global_batch"]
    assert out["value"] == max(out["train_images_per_sec_eager"], out["train_images_per_sec_graph"])
    assert abs(out["ms_per_step"] - 1e3 * out["config"]["global_batch"] / out["value"]) < 1e-2 * out["ms_per_step"]


def test_l2d_1280_hires_step_eval_and_postprocess():
    """BASELINE configs[3] at its own size: YOLOv10-L (2D head), 1280 x 1280, B = 1, bf16 - one training step (finite items and
    gradients), then eval: 33 600 anchors, and `v10postprocess` (HIP) of the one-to-one output equal to the oracle's postprocess of
    the SAME maps (scores to fp32 rounding, labels and boxes at the selected anchors exactly)"""
    from bench import synth_batch
    from oracle import restate as RS
    from yolov10_3d_amd.loss import v10postprocess
    y3d.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(0)
    model = y3d.YOLOv10DetectionModel("yolov10l.yaml").to(DEV).train()
    nc = model.yaml["nc"]
    batch = synth_batch(1, 1280, 1280, 11, DEV, nc=nc)
    for _ in range(2):  # the second pass runs on moved running statistics
        loss, items = model(batch)
        model.zero_grad(set_to_none=True)
        loss.backward()
    assert torch.isfinite(items).all(), items
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    model.eval()
    with torch.no_grad():
        y = model(batch["img"])["one2one"][0]
    assert y.shape == (1, 4 + nc, 160 * 160 + 80 * 80 + 40 * 40) and y.shape[2] == 33600
    preds = y.permute(0, 2, 1)
    bx, sc, lab = v10postprocess(preds, 300, nc)
    pc = preds.float().cpu()
    bx_o, sc_o, lab_o = RS.postprocess2d(pc, 300, nc)
    # the selected SCORES are determined (exactly tied scores - frequent among bf16 logits at random init - may come from different
    # anchors / classes: lowest index first here, library order in torch.topk) ...
    assert torch.equal(sc.cpu(), sc_o), f"max |score difference| {float((sc.cpu() - sc_o).abs().max()):.3e}"
    # ... and every detection must be a true (anchor, class) of the map: its box is an anchor's box whose score of that class is the score
    bxc, scc, labc = bx.cpu()[0], sc.cpu()[0], lab.cpu()[0]
    for j in range(300):
        at = (pc[0, :, :4] == bxc[j]).all(-1).nonzero().flatten()
        assert at.numel() >= 1, f"detection {j}: its box is no anchor's box"
        assert bool((pc[0, at, 4 + int(labc[j])] == scc[j]).any()), f"detection {j}: score / label do not belong to its box"
    untied = (sc_o[0, 1:] != sc_o[0, :-1]) & torch.cat(((sc_o[0, 2:] != sc_o[0, 1:-1]), torch.tensor([True])))
    if bool(untied.any()):  # where the oracle's score has no equal neighbour the pick itself is determined
        k = untied.nonzero().flatten() + 1
        assert torch.equal(labc[k], lab_o[0, k].long()) and torch.equal(bxc[k], bx_o[0, k])


def test_nonfinite_gradient_norm_skips_the_step_on_the_device():
    """f1: GradScaler.step's rule (engine/trainer.py:567-572) without a host synchronisation - a NaN / inf gradient leaves the
    parameters, the momentum / Adam state, AdamW's bias-correction step count and (guarded) the EMA untouched; the next finite step
    is applied as if the skipped one had not happened"""
    from yolov10_3d_amd.optim import FusedAdamW, FusedSGD, ModelEMA
    for cls in (FusedSGD, FusedAdamW):
        torch.manual_seed(4)
        net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8), torch.nn.Conv2d(8, 4, 1)).to(DEV)
        twin = copy.deepcopy(net)
        opts = [cls([p for p in m.parameters()], lr=0.05) for m in (net, twin)]
        ema = ModelEMA(net)
        grads = [[torch.randn_like(p) for p in net.parameters()] for _ in range(3)]

        def run(m, o, gs, poison=None):
            for p, g in zip(m.parameters(), gs):
                p.grad = g.clone()
            if poison is not None:
                list(m.parameters())[1].grad[0] = poison
            o.step(max_norm=10.0)
            o.zero_grad()

        run(net, opts[0], grads[0])
        run(twin, opts[1], grads[0])
        ema.update(net, guard=opts[0].last_norm)
        e0 = [v.clone() for v in ema.ema.state_dict().values()]
        for poison in (float("nan"), float("inf")):
            before = [p.detach().clone() for p in net.parameters()]
            state = opts[0]._state["flat"].clone()
            run(net, opts[0], grads[1], poison)   # skipped
            ema.update(net, guard=opts[0].last_norm)
            assert float(opts[0].last_norm[2]) == 0.0
            assert all(torch.equal(a, b.detach()) for a, b in zip(before, net.parameters())), "a skipped step moved the parameters"
            assert torch.equal(state, opts[0]._state["flat"]), "a skipped step moved the optimizer state"
            assert all(torch.equal(a, b) for a, b in zip(e0, ema.ema.state_dict().values())), "a guarded EMA update ran after a skipped step"
        run(net, opts[0], grads[2])
        run(twin, opts[1], grads[2])
        assert [float(v) for v in opts[0].last_norm[2:5]] == [1.0, 2.0, 2.0] and [float(v) for v in opts[1].last_norm[2:5]] == [1.0, 2.0, 0.0]
        for a, b in zip(net.parameters(), twin.parameters()):
            assert torch.equal(a.detach(), b.detach()), f"{cls.__name__}: the step after a skipped one differs from the never-poisoned twin"


def test_pad_targets_overflow_is_reported():
    """more boxes in one image than the assigner kernels' 64 rows must not be dropped silently"""
    from yolov10_3d_amd import loss as PL
    PL.check_target_overflow(wait=True)
    rows = torch.zeros(70 + 3, 6, device=DEV)
    rows[:70, 0] = 1  # image 1 has 70 boxes, image 0 has 3
    rows[:, 2:] = 0.5
    gt, n_used = PL.pad_targets(rows, 2, 5, (64.0, 64.0))
    assert int(n_used) == PL.TARGET_CAP and gt.shape == (2, 64, 5)
    with pytest.raises(y3d.Y3DError, match="70 ground-truth boxes"):
        PL.check_target_overflow(wait=True)
    PL.check_target_overflow(wait=True)  # reported once
    gt, n_used = PL.pad_targets(rows[60:], 2, 5, (64.0, 64.0))  # 10 + 3 boxes: fine
    PL.check_target_overflow(wait=True)
    assert int(n_used) == 10


@pytest.mark.parametrize("G,Cg,Cn,hw,B", [(14, 64, 64, (40, 40), 3), (2, 96, 40, (17, 23), 2), (16, 64, 24, (24, 24), 2)], ids=["m_head_14x64", "2x96to40", "16x64to24"])
def test_grouped_1x1_streaming_kernel_exact(G, Cg, Cn, hw, B):
    """grouped 1x1 convs (the second head layer of the M widths: 14 groups of 64 -> 64, k = 1) on the streaming kernel's group
    dimension (round 3): y, the BatchNorm partial sums and dx against fp32 conv2d / conv_transpose on small-integer operands (exact),
    and bit-identical to the generic kernel; the weight gradient (generic path) alongside"""
    H, W = hw
    gen = torch.Generator().manual_seed(G * 1000 + Cg + Cn)
    xf, wf, dyf = _sparse_int((B, G * Cg, H, W), gen, 0.25), _sparse_int((G * Cn, Cg, 1, 1), gen, 0.25), _sparse_int((B, G * Cn, H, W), gen, 0.25)
    L, st, dt, bf = y3d.lib(), ops.stream(), BF16, torch.bfloat16
    y3d.set_compute_dtype(bf)
    outs = []
    for flag in (1, 0):
        old = L.set_stream1x1(flag)
        try:
            xin = ops.nhwc_empty(B, G * Cg, H, W, bf, DEV)
            xin.copy_(xf.to(DEV))
            dyd = ops.nhwc_empty(B, G * Cn, H, W, bf, DEV)
            dyd.copy_(dyf.to(DEV))
            wd = wf.to(DEV).contiguous()
            sb, sh, sw = ops.s3(xin)
            wp = torch.empty(G * Cn * Cg, dtype=bf, device=DEV)
            L.pack_weight_fwd(dt, wd.data_ptr(), wp.data_ptr(), G * Cn, Cg, Cg, 1, 1, st)
            nblk = L.conv2d_stat_rows(dt, B, H, W, G * Cg, G * Cn, G, 1, 1, 1, 0)
            part = torch.full((nblk, G * Cn, 2), float("nan"), dtype=torch.float32, device=DEV)
            y = ops.nhwc_empty(B, G * Cn, H, W, bf, DEV)
            L.conv2d_fwd(dt, xin.data_ptr(), sb, sh, sw, B, H, W, G * Cg, wp.data_ptr(), None, y.data_ptr(), G * Cn, H, W, G * Cn, G, 1, 1, 1, 0, part.data_ptr(), st)
            kp = L.conv_kpad(dt, Cn)
            wpd = torch.empty(G * Cg * kp, dtype=bf, device=DEV)
            L.pack_weight_dgrad(dt, wd.data_ptr(), wpd.data_ptr(), G * Cn, Cg, G, 1, 1, st)
            dx = ops.nhwc_empty(B, G * Cg, H, W, bf, DEV)
            dsb, dsh, dsw = ops.s3(dyd)
            L.conv2d_bwd_data(dt, dyd.data_ptr(), dsb, dsh, dsw, B, H, W, G * Cn, wpd.data_ptr(), dx.data_ptr(), G * Cg, H, W, G * Cg, G, 1, 1, 1, 0, st)
            torch.cuda.synchronize()
            outs.append((y.float().cpu(), part.double().sum(0).cpu(), dx.float().cpu(), nblk))
        finally:
            L.set_stream1x1(old)
    ref_y = F.conv2d(xf, wf, groups=G)
    ref_dx = torch.nn.grad.conv2d_input(xf.shape, wf, dyf, groups=G)
    ref_st = torch.stack((ref_y.double().sum((0, 2, 3)), (ref_y.double() ** 2).sum((0, 2, 3))), 1)
    if Cg >= 32 and Cn % 4 == 0 and B * H * W >= 128:
        assert outs[0][3] != outs[1][3] or G == 1, "the streaming kernel was expected to take this shape (its partial rows are its workers)"
    for (yy, stt, dxx, _), which in zip(outs, ("streaming", "generic")):
        assert torch.equal(yy, ref_y), f"y ({which})"
        assert torch.equal(stt, ref_st), f"BatchNorm partial sums ({which})"
        assert torch.equal(dxx, ref_dx), f"dx ({which})"


def test_bench_configuration_bf16_step_tracks_the_exact_fp32_step():
    """BASELINE configs[1] exactly as bench.py runs it - YOLOv10-S + 3D head, 640x640, B = 32, bf16, synth_batch seed 1 - against the SAME
    step in the exact-fp32 mode of the same kernels on the GPU (the mode the oracle tests hold to the reference within 1e-3): loss items,
    the assigner's foreground masks, every parameter's gradient norm.  Bounds are ~1.5-2x what was measured on an MI355X (printed).  At
    random init the alignment metrics are nearly tied, so bf16 rounding re-assigns 20 % of the one-to-many and 35 % of the one-to-one
    positives; the loss items (averages over all positives) still agree within 1.6 % / 8 %."""
    import bench
    torch.manual_seed(0)
    model = y3d.YOLOv10_3DDetectionModel("yolov10s_3D.yaml").to(DEV).train()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    batch = bench.synth_batch(32, 640, 640, 1, DEV)
    res = {}
    try:
        for dt in (torch.float32, torch.bfloat16):
            y3d.set_compute_dtype(dt)
            model.load_state_dict(state)
            model.zero_grad(set_to_none=True)
            loss, items = model(batch)
            loss.backward()
            crit = model.criterion
            fg = [crit.one2many.last_assignment[0].clone(), crit.one2one.last_assignment[0].clone()]
            gi = [crit.one2many.last_assignment[1].clone(), crit.one2one.last_assignment[1].clone()]
            norms = {k: float(p.grad.float().norm()) for k, p in model.named_parameters() if p.grad is not None}
            res[dt] = (items.float().cpu(), fg, gi, norms)
            assert torch.isfinite(items).all()
    finally:
        y3d.set_compute_dtype(torch.bfloat16)
    i32, fg32, gi32, n32 = res[torch.float32]
    i16, fg16, gi16, n16 = res[torch.bfloat16]
    rel = ((i16 - i32).abs() / i32.abs().clamp(min=1e-6)).tolist()
    agree = []
    for a, b, ga, gb in zip(fg32, fg16, gi32, gi16):
        both = (a & b)
        agree.append((float(both.sum()) / float(a.sum()), float((both & (ga == gb)).sum()) / float(a.sum()), int(a.sum()), int(b.sum())))
    # gradients that are zero in exact arithmetic (a BatchNorm bias in front of another BatchNorm) are pure rounding noise in both modes:
    # norms are compared on the scale of the largest one
    floor = 1e-3 * max(n32.values())
    ratios = sorted((n16[k] / n32[k], k) for k in n32 if n32[k] > floor)
    dev = sorted((abs(n16[k] - n32[k]) / (n32[k] + floor), k) for k in n32)
    print(f"bf16 vs exact fp32 at the bench configuration: loss items rel. diff {[round(v, 4) for v in rel]}")
    print(f"  foreground agreement (fraction of fp32 positives kept, kept with the same box, #fp32, #bf16): one-to-many {agree[0]}, one-to-one {agree[1]}")
    print(f"  gradient-norm ratio bf16 / fp32 over {len(ratios)} parameters: median {ratios[len(ratios) // 2][0]:.4f}, worst {dev[-1]}, 99th percentile {dev[int(0.99 * len(dev))]}")
    assert max(rel[:6]) < 0.035, f"one-to-many loss items: {rel[:6]}"
    assert max(rel[6:]) < 0.16, f"one-to-one loss items: {rel[6:]}"
    assert agree[0][1] > 0.65 and agree[1][1] > 0.5, agree  # random init: near-tied metrics, the discrete choice is fragile (measured 0.80 / 0.65)
    # measured: median 1.5 %, 99th percentile 34 %, worst 46 % - the tail is the P5-level branches of the one-to-one set, whose gradient is the
    # sum over a few dozen positives of which bf16 rounding re-assigns a third at random init (agreement above); the body follows the median
    p50, p90 = dev[len(dev) // 2][0], dev[int(0.9 * len(dev))][0]
    print(f"  |norm16 - norm32| / norm32: median {p50:.4f}, 90th percentile {p90:.4f}")
    assert p50 < 0.04 and p90 < 0.25 and dev[int(0.99 * len(dev))][0] < 0.5 and dev[-1][0] < 0.7, (p50, p90, dev[-3:])


def test_captured_data_parallel_step_matches_the_eager_reducer_step():
    """the N > 1 step (forward, loss, backward with the reducer's hook-driven bucket gathers + RCCL all-reduces,
    clip, SGD) recorded into ONE hipGraph - rehearsed on one GPU as a process group of size 1 over RCCL (Y3D_FORCE_DDP=1: the collectives
    are issued and captured) - leaves the model where the eager reducer step leaves it, bit for bit (tools/probe/ddp_graph_probe.py, in
    a child process: it owns a process group and a segfault of the runtime must not take the suite down)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, Y3D_FORCE_DDP="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "probe", "ddp_graph_probe.py")], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "captured reducer step == eager reducer step: True" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
