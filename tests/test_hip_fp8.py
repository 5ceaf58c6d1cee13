"""GPU: the fp8 MFMA convolution family (csrc/conv3x3_fp8.hip, BASELINE configs[4]) through the C ABI against the oracle's restatement of its
operand formats (oracle/restate.py: fp8w_quantize, mx_quantize_act, conv3x3_fp8).  The reference has no 8-bit path: the stated bounds are
    * quantisers: bit-exact codes and scale bytes against torch.float8_e4m3fn arithmetic;
    * convolution on operands whose quantisation is exact (small integers): bit-exact against an fp32 convolution;
    * convolution on real operands: fp8 x fp8 products are exact and only the fp32 accumulation order differs from the oracle's float64
      sum of the SAME quantised operands: |y - y_oracle| <= 2 * K * 2^-24 * sum|x||w| + one bf16 rounding of the result."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import yolov10_3d_amd as y3d  # noqa: E402
from yolov10_3d_amd import ops  # noqa: E402
from oracle import restate as RS  # noqa: E402  (the checker)

DEV = "cuda"


def quantize_act(x_nchw_bf16):
    """bf16 (B, C, H, W) -> (codes (B, H, W, C) uint8, scales (B, H, W, C / 32) uint8) through y3d_fp8_quantize_act"""
    xin = ops.to_nhwc(x_nchw_bf16, torch.bfloat16, dense=True)
    B, C, H, W = xin.shape
    q = torch.empty(B, H, W, C, dtype=torch.uint8, device=xin.device)
    s = torch.empty(B, H, W, C // 32, dtype=torch.uint8, device=xin.device)
    ops.lib().fp8_quantize_act(xin.data_ptr(), xin.stride(3), B * H * W, C, q.data_ptr(), s.data_ptr(), ops.stream())
    return q, s


def pack_weight(w):
    """fp32 OIHW master -> (wq (rows, 9, Cg) uint8, ws (rows,) uint8, w_eff) through the fp8w quantiser + y3d_fp8_pack_weight_fwd"""
    w = w.float().contiguous()
    rows, Cg = w.shape[0], w.shape[1]
    K = Cg * 9
    codes = torch.empty(rows, K, dtype=torch.uint8, device=w.device)
    weff = torch.empty_like(w)
    scale = torch.empty(rows, dtype=torch.float32, device=w.device)
    desc = torch.tensor([w.data_ptr(), weff.data_ptr(), codes.data_ptr(), scale.data_ptr(), rows, K], dtype=torch.int64, device=w.device)
    rb = torch.zeros(1, dtype=torch.int32, device=w.device)
    L = ops.lib()
    L.mt_fp8w_quantize(desc.data_ptr(), rb.data_ptr(), 1, rows, ops.stream())
    wq = torch.empty(rows, 9, Cg, dtype=torch.uint8, device=w.device)
    ws = torch.empty(rows, dtype=torch.uint8, device=w.device)
    L.fp8_pack_weight_fwd(codes.data_ptr(), scale.data_ptr(), rows, Cg, wq.data_ptr(), ws.data_ptr(), ops.stream())
    return wq, ws, weff


def conv_fp8(x, w, groups, stats=True, affine=None, act=0):
    q, s = quantize_act(x)
    wq, ws, _ = pack_weight(w)
    B, C, H, W = x.shape
    Cout = w.shape[0]
    y = ops.nhwc_empty(B, Cout, H, W, torch.bfloat16, x.device)
    L = ops.lib()
    part = None
    if stats and affine is None:
        rows = L.conv3x3_fp8_stat_rows(B, H, W)
        part = torch.full((rows, Cout, 2), float("nan"), dtype=torch.float32, device=x.device)
    sc, sh = (affine if affine is not None else (None, None))
    L.conv3x3_fp8_fwd(q.data_ptr(), s.data_ptr(), B, H, W, C, wq.data_ptr(), ws.data_ptr(), y.data_ptr(), y.stride(3), Cout, groups,
                      part.data_ptr() if part is not None else None, sc.data_ptr() if sc is not None else None, sh.data_ptr() if sh is not None else None, act,
                      ops.stream())
    torch.cuda.synchronize()
    return y, part


@pytest.mark.parametrize("shape", [(2, 64, 5, 7), (1, 128, 16, 16), (3, 96, 3, 9)])
def test_fp8_act_quantizer_bit_exact_vs_oracle(shape):
    torch.manual_seed(sum(shape))
    B, C, H, W = shape
    x = torch.randn(B, C, H, W) * torch.exp2(torch.randint(-12, 8, (B, C // 32, 1, H, W)).float()).repeat_interleave(32, 1).reshape(B, C, H, W)
    x[0, :32, 0, 0] = 0.0                       # an all-zero block: scale byte 127
    x[0, 32:64, 0, 0] = 448.0                   # exactly the largest code at scale 1
    x[0, 32, 0, 1] = 449.0                      # just above: the next power of two
    x[0, :32, 1, 1] = 2.0 ** -130               # below the scale range: exponent clamps at -127
    x = x.to(torch.bfloat16)
    q, s = quantize_act(x.to(DEV))
    cq, cs, _ = RS.mx_quantize_act(x.float())
    assert torch.equal(s.cpu().permute(0, 3, 1, 2), cs), "E8M0 scale bytes differ"
    assert torch.equal(q.cpu().permute(0, 3, 1, 2), cq), "e4m3 codes differ"


CONV_CASES = [  # B, Cin, Cout, groups, H, W
    (4, 128, 128, 1, 8, 16), (5, 128, 256, 1, 16, 16), (2, 256, 128, 2, 20, 20), (3, 1024, 1024, 8, 12, 40), (1, 384, 96, 2, 9, 23),
    (9, 2048, 2048, 16, 8, 16), (2, 128, 2048, 1, 24, 24),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_fp8_conv_exact_on_integer_operands(case):
    """operands whose quantisation is exact - activations in {0, +-1, +-2, +-3} times a per-block power of two, weights in {0, +-1, +-2} times
    a per-row power of two - make every product and every fp32 partial sum exact: the kernel must equal an fp32 host convolution bit for
    bit, BatchNorm partial sums included (any tile order)"""
    B, Cin, Cout, g, H, W = case
    torch.manual_seed(Cin + Cout + H)
    x = torch.randint(-3, 4, (B, Cin, H, W)).float() * (torch.rand(B, Cin, H, W) < 0.3)
    x = x * torch.exp2(torch.randint(-3, 4, (B, Cin // 32, 1, H, W)).float()).repeat_interleave(32, 1).reshape(B, Cin, H, W)
    w = torch.randint(-2, 3, (Cout, Cin // g, 3, 3)).float() * (torch.rand(Cout, Cin // g, 3, 3) < 0.25)
    w = w * torch.exp2(torch.randint(-4, 3, (Cout, 1, 1, 1)).float())
    ref = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1, 1, g)
    assert float(ref.abs().max()) < 2 ** 15
    y, part = conv_fp8(x.to(torch.bfloat16).to(DEV), w.to(DEV), g)
    refb = ref.float().to(torch.bfloat16).float()  # the stored value: one bf16 rounding of an exact fp32 sum
    assert torch.equal(y.float().cpu(), refb), f"max |diff| {float((y.float().cpu() - refb).abs().max())}"
    ps = part.double().sum(0).cpu()
    # sums of the stored values: exact (multiples of 2^-7 below 2^15 add exactly in fp32 over a tile); their squares need up to 44 bits
    assert torch.equal(ps[:, 0], refb.double().sum((0, 2, 3)))
    assert torch.allclose(ps[:, 1], (refb.double() ** 2).sum((0, 2, 3)), rtol=1e-5, atol=0)


@pytest.mark.parametrize("case", [(4, 128, 256, 1, 16, 16), (2, 2048, 2048, 16, 20, 20)], ids=["128to256", "head_layer2_p5"])
def test_fp8_conv_real_operands_within_accumulation_bound(case):
    B, Cin, Cout, g, H, W = case
    torch.manual_seed(7)
    x = torch.nn.functional.silu(torch.randn(B, Cin, H, W) * 1.5).to(torch.bfloat16)
    w = torch.randn(Cout, Cin // g, 3, 3) * 0.05
    y, _ = conv_fp8(x.to(DEV), w.to(DEV), g)
    _, _, w_eff = RS.fp8w_quantize(w)
    _, _, x_eff = RS.mx_quantize_act(x.float())
    ref = torch.nn.functional.conv2d(x_eff.double(), w_eff.double(), None, 1, 1, 1, g)
    mag = torch.nn.functional.conv2d(x_eff.double().abs(), w_eff.double().abs(), None, 1, 1, 1, g)
    K = 9 * Cin // g
    bound = 2 * K * 2.0 ** -24 * mag + 2.0 ** -8 * ref.abs() + 1e-30
    err = (y.double().cpu() - ref).abs()
    assert bool((err <= bound).all()), f"worst error / bound {float((err / bound).max()):.3f}"
    # and the format itself: how far the fp8 result is from the unquantised convolution (reported, loosely bounded)
    full = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1, 1, g)
    rel = float((ref - full).norm() / full.norm())
    print(f"fp8 (MX activations x fp8w weights) vs unquantised conv: relative L2 {rel:.3e}")
    assert rel < 6e-2


def test_fp8_conv_affine_epilogue_matches_training_form():
    """eval form: folded BatchNorm + SiLU in the epilogue = the training form's raw output put through the same affine map"""
    torch.manual_seed(11)
    B, Cin, Cout, g, H, W = 3, 256, 256, 2, 12, 20
    x = torch.nn.functional.silu(torch.randn(B, Cin, H, W)).to(torch.bfloat16).to(DEV)
    w = (torch.randn(Cout, Cin // g, 3, 3) * 0.05).to(DEV)
    sc = (torch.rand(Cout, device=DEV) + 0.5).contiguous()
    sh = torch.randn(Cout, device=DEV).contiguous()
    raw, _ = conv_fp8(x, w, g, stats=False)
    out, _ = conv_fp8(x, w, g, affine=(sc, sh), act=1)
    # the raw form stores bf16(acc), the affine form applies the map to the fp32 accumulator: equal within one bf16 rounding of acc
    want = torch.nn.functional.silu(raw.float() * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    tol = 2.0 ** -7 * (raw.float().abs() * sc.view(1, -1, 1, 1) + 1.0)
    assert bool(((out.float() - want).abs() <= tol).all())
