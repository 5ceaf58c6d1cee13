"""1x1 convolution forward / data gradient through the C ABI: streaming kernel (conv1x1_stream.hip) against the generic implicit-GEMM
kernel (conv_gemm.hip) on the body's shapes - equality of the outputs, time per launch, effective bandwidth.

    python tools/conv1x1_bench.py [reps]

Bytes = pixel operand + output (the weights are L2-resident)."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
y3d = importlib.import_module("yolov10-3d_amd")
ops = importlib.import_module("yolov10-3d_amd.ops")

DEV = torch.device("cuda:0")
BF16 = 1

# (H, Cin, Cout) at B = 32, from tools/layer_report.py on yolov10s_3D.yaml 640x640
SHAPES = [(320, 32, 32), (160, 96, 64), (160, 64, 64), (80, 384, 128), (80, 128, 256), (80, 256, 128), (80, 192, 128), (80, 128, 128),
          (40, 384, 256), (40, 256, 256), (40, 768, 256), (40, 256, 512), (40, 512, 256),
          (20, 768, 512), (20, 512, 512), (20, 512, 256), (20, 256, 512), (20, 1024, 512), (20, 256, 256)]


def run(L, st, dt, B, H, Cin, Cout, reps, stream_on):
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(H * 1000 + Cin + Cout)
    x = ops.nhwc_empty(B, Cin, H, H, bf, DEV)
    x.copy_(torch.randn(B, Cin, H, H, generator=g).to(DEV))
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5).to(DEV)
    sb, sh, sw = ops.s3(x)
    wp = torch.empty(Cout * Cin, dtype=bf, device=DEV)
    L.pack_weight_fwd(dt, w.data_ptr(), wp.data_ptr(), Cout, Cin, Cin, 1, 1, st)
    old0 = L.set_stream1x1(1 if stream_on else 0)
    nblk = L.conv2d_stat_rows(dt, B, H, H, Cin, Cout, 1, 1, 1, 1, 0)  # depends on the kernel the entry point will pick
    L.set_stream1x1(old0)
    part = torch.zeros((nblk, Cout, 2), dtype=torch.float32, device=DEV)
    y = ops.nhwc_empty(B, Cout, H, H, bf, DEV)
    kp = L.conv_kpad(dt, Cout)
    wpd = torch.empty(Cin * kp, dtype=bf, device=DEV)
    L.pack_weight_dgrad(dt, w.data_ptr(), wpd.data_ptr(), Cout, Cin, 1, 1, 1, st)
    dx = ops.nhwc_empty(B, Cin, H, H, bf, DEV)
    old = L.set_stream1x1(1 if stream_on else 0)

    def fwd():
        L.conv2d_fwd(dt, x.data_ptr(), sb, sh, sw, B, H, H, Cin, wp.data_ptr(), None, y.data_ptr(), Cout, H, H, Cout, 1, 1, 1, 1, 0, part.data_ptr(), st)

    def dgrad():
        ysb, ysh, ysw = ops.s3(y)
        L.conv2d_bwd_data(dt, y.data_ptr(), ysb, ysh, ysw, B, H, H, Cout, wpd.data_ptr(), dx.data_ptr(), Cin, H, H, Cin, 1, 1, 1, 1, 0, st)

    ns = L.conv2d_wgrad_plan(dt, B, H, H, Cin, Cout, 1, 1, 1, 1, 0)
    slab = torch.empty(ns * Cout * Cin, dtype=torch.float32, device=DEV)
    dW = torch.empty_like(w)

    def wgrad():
        ysb, ysh, ysw = ops.s3(y)
        L.conv2d_bwd_weight(dt, x.data_ptr(), sb, sh, sw, B, H, H, Cin, Cin, y.data_ptr(), ysw, H, H, Cout, 1, 1, 1, 1, 0, slab.data_ptr(), ns,
                            dW.data_ptr(), 0, st)

    out = []
    try:
        for fn in (fwd, dgrad, wgrad):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / reps * 1e3)
    finally:
        L.set_stream1x1(old)
    return out, y.clone(), part.double().sum(0), dx.clone(), dW.clone()


def main():
    global SHAPES
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    if len(sys.argv) > 2:  # "H,Cin,Cout;H,Cin,Cout;..."
        SHAPES = [tuple(int(v) for v in t.split(",")) for t in sys.argv[2].split(";")]
    L, st, dt = y3d.lib(), ops.stream(), BF16
    B = 32
    print(f"{'shape':>22} {'fwd old':>9} {'fwd new':>9} {'TB/s':>6} {'dgrad old':>10} {'dgrad new':>10} {'TB/s':>6} {'wgrad old':>10} {'wgrad new':>10} {'TF/s':>6}  equal")
    tot = [0.0] * 6
    for H, Cin, Cout in SHAPES:
        (fo, do, wo), y0, p0, dx0, dw0 = run(L, st, dt, B, H, Cin, Cout, reps, False)
        (fn, dn, wn), y1, p1, dx1, dw1 = run(L, st, dt, B, H, Cin, Cout, reps, True)
        M = B * H * H
        byt = M * (Cin + Cout) * 2
        eq = bool(torch.equal(y0, y1)) and bool(torch.equal(dx0, dx1))  # dW: the two paths use different split-K plans
        dwerr = float((dw0 - dw1).abs().max() / dw0.abs().max())
        perr = float(((p0 - p1).abs() / (p0.abs() + 1e-3)).max())
        print(f"{H:4d}x{H:<4d} {Cin:4d}->{Cout:<4d} {fo:9.1f} {fn:9.1f} {byt / fn / 1e6:6.2f} {do:10.1f} {dn:10.1f} {byt / dn / 1e6:6.2f} {wo:10.1f} {wn:10.1f} {2.0 * M * Cin * Cout / wn / 1e6:6.0f}  {eq} stats rel {perr:.1e} dW rel {dwerr:.1e}")
        for i, v in enumerate((fo, fn, do, dn, wo, wn)):
            tot[i] += v
    print("sum us: fwd old %.0f new %.0f, dgrad old %.0f new %.0f, wgrad old %.0f new %.0f" % tuple(tot))


if __name__ == "__main__":
    main()
