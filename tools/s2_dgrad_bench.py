"""3x3 stride-2 data gradient through the C ABI: resident-tile kernel (conv3x3s2_dgrad.hip) against the generic parity-class kernel.
    python tools/s2_dgrad_bench.py"""
import importlib, sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
y3d = importlib.import_module("yolov10-3d_amd"); ops = importlib.import_module("yolov10-3d_amd.ops")
DEV = torch.device("cuda:0"); L, st, dt = y3d.lib(), ops.stream(), 1
for (B, H, Cin, Cout) in [(32, 320, 32, 64), (32, 160, 64, 128), (32, 80, 128, 128)]:
    Ho = H // 2
    dy = ops.nhwc_empty(B, Cout, Ho, Ho, torch.bfloat16, DEV); dy.copy_(torch.randn(B, Cout, Ho, Ho, device=DEV))
    w = torch.randn(Cout, Cin, 3, 3, device=DEV) * 0.05
    kp = L.conv_kpad(dt, 9 * Cout); wpd = torch.empty(Cin * kp, dtype=torch.bfloat16, device=DEV)
    L.pack_weight_dgrad(dt, w.data_ptr(), wpd.data_ptr(), Cout, Cin, 1, 3, 3, st)
    dx = ops.nhwc_empty(B, Cin, H, H, torch.bfloat16, DEV)
    sb, sh, sw = ops.s3(dy)
    res = []
    for flag in (0, 1):
        old = L.set_stream1x1(flag)
        f = lambda: L.conv2d_bwd_data(dt, dy.data_ptr(), sb, sh, sw, B, Ho, Ho, Cout, wpd.data_ptr(), dx.data_ptr(), Cin, H, H, Cin, 1, 3, 3, 2, 1, st)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
        L.set_stream1x1(old)
    mb = (B * Ho * Ho * Cout + B * H * H * Cin) * 2 / 1e6
    print(f"{H}x{H} {Cout}->{Cin}: generic {res[0]:.1f} us, resident tile {res[1]:.1f} us ({mb:.0f} MB of tensors: {mb / res[1]:.2f} TB/s)")
