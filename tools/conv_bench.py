"""Micro-benchmark of y3d_conv2d_fwd / bwd_data / bwd_weight through the C ABI (no model): TFLOP/s per shape.
    python tools/conv_bench.py [--dtype bf16|f32] [--iters 20]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolov10_3d_amd as y3d
from yolov10_3d_amd import ops

SHAPES = [  # B, H, W, Cin, Cout, k, s, g
    (32, 80, 80, 2048, 2048, 3, 1, 16),   # fused head layer 2 @P3 (the headline shape x16)
    (32, 80, 80, 128, 2048, 3, 1, 1),     # fused head layer 1 @P3
    (32, 40, 40, 256, 2048, 3, 1, 1),     # layer 1 @P4
    (32, 40, 40, 2048, 2048, 3, 1, 16),   # layer 2 @P4
    (32, 20, 20, 512, 2048, 3, 1, 1),     # layer 1 @P5
    (32, 40, 40, 128, 128, 3, 1, 1),      # body bottleneck
    (32, 80, 80, 64, 64, 3, 1, 1),
    (32, 40, 40, 384, 256, 1, 1, 1),      # 1x1
    (32, 80, 80, 64, 128, 3, 2, 1),       # stride 2
    (32, 80, 80, 96, 96, 3, 1, 1),        # M-3D body (channel counts that are multiples of 32, not 64)
    (32, 40, 40, 192, 192, 3, 1, 1),
    (32, 160, 160, 48, 48, 3, 1, 1),
    (32, 80, 80, 2048, 384, 1, 1, 16),    # the 16 head projections of a level as a grouped 1x1 conv padded to 24 outputs per branch
    (32, 40, 40, 2048, 384, 1, 1, 16),
    (32, 320, 320, 32, 64, 3, 2, 1),      # first backbone downsample: narrow stride-2 (the data gradient is latency-bound)
    (32, 160, 160, 64, 128, 3, 2, 1),
    (16, 160, 160, 80, 80, 3, 1, 1),      # X widths: 2.5 K slabs on the persistent kernel (round 3)
    (16, 80, 80, 160, 160, 3, 1, 1),
]

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", type=int, default=-1)
    ap.add_argument("--bwd", action="store_true")
    ap.add_argument("--no-tile", action="store_true", help="route everything through the generic implicit-GEMM kernels")
    a = ap.parse_args()
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    y3d.set_compute_dtype(dtype)
    L = y3d.lib()
    if a.no_tile:
        L.set_tile_kernels(0)
    dt = ops.code(dtype)
    dev = "cuda"
    for i, (B, H, W, Cin, Cout, k, s, g) in enumerate(SHAPES):
        if a.only >= 0 and i != a.only:
            continue
        p = k // 2
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        x = ops.nhwc_empty(B, Cin, H, W, dtype, dev); x.normal_()
        w = torch.randn(Cout, Cin // g, k, k, device=dev) * 0.05
        wp = torch.empty(Cout * k * k * (Cin // g), dtype=dtype, device=dev)
        st = ops.stream()
        L.pack_weight_fwd(dt, w.data_ptr(), wp.data_ptr(), Cout, Cin // g, Cin // g, k, k, st)
        y = ops.nhwc_empty(B, Cout, Ho, Wo, dtype, dev)
        rows = L.conv2d_stat_rows(dt, B, H, W, Cin, Cout, g, k, k, s, p)
        part = torch.empty(rows * Cout * 2, device=dev)
        sb, sh, sw = ops.s3(x)
        flops = 2.0 * B * Ho * Wo * Cout * (Cin // g) * k * k
        def fwd():
            L.conv2d_fwd(dt, x.data_ptr(), sb, sh, sw, B, H, W, Cin, wp.data_ptr(), None, y.data_ptr(), Cout, Ho, Wo, Cout, g, k, k, s, p, part.data_ptr(), st)
        runs = [("fwd", fwd)]
        if a.bwd:
            dy = ops.nhwc_empty(B, Cout, Ho, Wo, dtype, dev); dy.normal_()
            kp = L.conv_kpad(dt, k * k * (Cout // g))
            wpd = torch.empty(Cin * kp, dtype=dtype, device=dev)
            L.pack_weight_dgrad(dt, w.data_ptr(), wpd.data_ptr(), Cout, Cin // g, g, k, k, st)
            dx = ops.nhwc_empty(B, Cin, H, W, dtype, dev)
            dsb, dsh, dsw = ops.s3(dy)
            ns = L.conv2d_wgrad_plan(dt, B, H, W, Cin, Cout, g, k, k, s, p)
            slab = torch.empty(ns * Cout * k * k * (Cin // g), device=dev)
            dW = torch.empty_like(w)
            runs.append(("dgrad", lambda: L.conv2d_bwd_data(dt, dy.data_ptr(), dsb, dsh, dsw, B, Ho, Wo, Cout, wpd.data_ptr(), dx.data_ptr(), Cin, H, W, Cin, g, k, k, s, p, st)))
            runs.append(("wgrad", lambda: L.conv2d_bwd_weight(dt, x.data_ptr(), sb, sh, sw, B, H, W, Cin, Cin, dy.data_ptr(), Cout, Ho, Wo, Cout, g, k, k, s, p, slab.data_ptr(), ns, dW.data_ptr(), 0, st)))
        for name, fn in runs:
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            print(f"{name:6s} B{B} {H}x{W} {Cin}->{Cout} k{k} s{s} g{g}: {ms:8.3f} ms  {flops / ms / 1e9:8.1f} TFLOP/s", flush=True)

if __name__ == "__main__":
    main()
