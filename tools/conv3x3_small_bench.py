"""Narrow 3x3 convs through the C ABI: resident-weights kernel (conv3x3_small.hip) against the kernels it replaces.
    python tools/conv3x3_small_bench.py"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
y3d = importlib.import_module("yolov10-3d_amd"); ops = importlib.import_module("yolov10-3d_amd.ops")
DEV = torch.device("cuda:0"); L, st, dt = y3d.lib(), ops.stream(), 1
for (B, H, Cin, Cout) in [(32, 160, 32, 32), (32, 80, 64, 64), (32, 40, 64, 64)]:
    bf = torch.bfloat16
    x = ops.nhwc_empty(B, Cin, H, H, bf, DEV); x.copy_(torch.randn(B, Cin, H, H, device=DEV))
    w = torch.randn(Cout, Cin, 3, 3, device=DEV) * 0.05
    wp = torch.empty(Cout * 9 * Cin, dtype=bf, device=DEV)
    L.pack_weight_fwd(dt, w.data_ptr(), wp.data_ptr(), Cout, Cin, Cin, 3, 3, st)
    kp = L.conv_kpad(dt, 9 * Cout); wpd = torch.empty(Cin * kp, dtype=bf, device=DEV)
    L.pack_weight_dgrad(dt, w.data_ptr(), wpd.data_ptr(), Cout, Cin, 1, 3, 3, st)
    y = ops.nhwc_empty(B, Cout, H, H, bf, DEV); dx = ops.nhwc_empty(B, Cin, H, H, bf, DEV)
    sb, sh, sw = ops.s3(x); ysb, ysh, ysw = ops.s3(y)
    res = {}
    for flag in (0, 1):
        old = L.set_stream1x1(flag)
        nblk = L.conv2d_stat_rows(dt, B, H, H, Cin, Cout, 1, 3, 3, 1, 1)
        part = torch.zeros(nblk * Cout * 2, device=DEV)
        fns = {"fwd": lambda: L.conv2d_fwd(dt, x.data_ptr(), sb, sh, sw, B, H, H, Cin, wp.data_ptr(), None, y.data_ptr(), Cout, H, H, Cout, 1, 3, 3, 1, 1, part.data_ptr(), st),
               "dgrad": lambda: L.conv2d_bwd_data(dt, y.data_ptr(), ysb, ysh, ysw, B, H, H, Cout, wpd.data_ptr(), dx.data_ptr(), Cin, H, H, Cin, 1, 3, 3, 1, 1, st)}
        for k, f in fns.items():
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            res[(k, flag)] = e0.elapsed_time(e1) / 20 * 1e3
        L.set_stream1x1(old)
    fl = 2.0 * B * H * H * Cin * Cout * 9
    print(f"{H}x{H} {Cin}->{Cout}: fwd {res[('fwd', 0)]:.1f} -> {res[('fwd', 1)]:.1f} us ({fl / res[('fwd', 1)] / 1e6:.0f} TFLOP/s), "
          f"dgrad {res[('dgrad', 0)]:.1f} -> {res[('dgrad', 1)]:.1f} us ({fl / res[('dgrad', 1)] / 1e6:.0f} TFLOP/s)")
