"""fp8 MFMA conv forward (csrc/conv3x3_fp8.hip) next to the bf16 persistent kernel on the head shapes, through the C ABI (HIP events,
back-to-back launches on random operands).  python tools/fp8_conv_bench.py [B]"""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import yolov10_3d_amd as y3d
from yolov10_3d_amd import ops
from test_hip_fp8 import quantize_act, pack_weight
DEV = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L = ops.lib()

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

print(f"{'shape':44s} {'bf16 ms':>8s} {'TF/s':>6s} {'fp8 ms':>8s} {'TF/s':>6s} {'quant ms':>9s} speed-up (conv only / with quantiser)")
for (Cin, Cout, g, H, W) in [(2048, 2048, 16, 80, 80), (128, 2048, 1, 80, 80), (2048, 2048, 16, 40, 40), (256, 2048, 1, 40, 40), (2048, 2048, 16, 20, 20), (512, 2048, 1, 20, 20),
                             (640, 2048, 1, 40, 40), (640, 2048, 1, 20, 20)]:
    torch.manual_seed(0)
    x = torch.nn.functional.silu(torch.randn(B, Cin, H, W, device=DEV)).to(torch.bfloat16)
    xin = ops.to_nhwc(x, torch.bfloat16, dense=True)
    w = (torch.randn(Cout, Cin // g, 3, 3, device=DEV) * 0.05)
    flops = 2.0 * B * H * W * Cout * (Cin // g) * 9
    # bf16
    wp = torch.empty(Cout * 9 * (Cin // g), dtype=torch.bfloat16, device=DEV)
    L.pack_weight_fwd(ops.code(torch.bfloat16), w.data_ptr(), wp.data_ptr(), Cout, Cin // g, Cin // g, 3, 3, ops.stream())
    y = ops.nhwc_empty(B, Cout, H, W, torch.bfloat16, DEV)
    rows = L.conv2d_stat_rows(ops.code(torch.bfloat16), B, H, W, Cin, Cout, g, 3, 3, 1, 1)
    part = torch.empty(rows * Cout * 2, dtype=torch.float32, device=DEV)
    sb, sh, sw = ops.s3(xin)
    t16 = timeit(lambda: L.conv2d_fwd(ops.code(torch.bfloat16), xin.data_ptr(), sb, sh, sw, B, H, W, Cin, wp.data_ptr(), None, y.data_ptr(), Cout, H, W, Cout, g, 3, 3, 1, 1,
                                      part.data_ptr(), ops.stream()))
    # fp8
    q, s = quantize_act(x)
    wq, ws, _ = pack_weight(w)
    rows8 = L.conv3x3_fp8_stat_rows(B, H, W)
    part8 = torch.empty(rows8 * Cout * 2, dtype=torch.float32, device=DEV)
    t8 = timeit(lambda: L.conv3x3_fp8_fwd(q.data_ptr(), s.data_ptr(), B, H, W, Cin, wq.data_ptr(), ws.data_ptr(), y.data_ptr(), y.stride(3), Cout, g, part8.data_ptr(), None, None, 0,
                                          ops.stream()))
    tq = timeit(lambda: L.fp8_quantize_act(xin.data_ptr(), xin.stride(3), B * H * W, Cin, q.data_ptr(), s.data_ptr(), ops.stream()))
    print(f"B={B} {Cin:4d}->{Cout} g={g:2d} @{H}x{W}".ljust(44) + f" {t16:8.3f} {flops / t16 / 1e9:6.0f} {t8:8.3f} {flops / t8 / 1e9:6.0f} {tq:9.3f}  x{t16 / t8:.2f} / x{t16 / (t8 + tq):.2f}")
