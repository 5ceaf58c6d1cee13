"""Depth-wise convs of S-3D through the C ABI: time per launch and effective bandwidth (tensor bytes in + out).
    python tools/dw_bench.py"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
y3d = importlib.import_module("yolov10-3d_amd"); ops = importlib.import_module("yolov10-3d_amd.ops")
DEV = torch.device("cuda:0"); L, st, dt = y3d.lib(), ops.stream(), 1


def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (B, H, C, k, s) in [(32, 80, 256, 3, 2), (32, 40, 512, 3, 2), (32, 40, 256, 3, 2), (32, 20, 256, 3, 1), (32, 20, 512, 3, 1), (32, 20, 512, 7, 1)]:
    p = k // 2
    Ho = (H + 2 * p - k) // s + 1
    bf = torch.bfloat16
    x = ops.nhwc_empty(B, C, H, H, bf, DEV); x.copy_(torch.randn(B, C, H, H, device=DEV))
    y = ops.nhwc_empty(B, C, Ho, Ho, bf, DEV); dx = ops.nhwc_empty(B, C, H, H, bf, DEV)
    w = torch.randn(C, 1, k, k, device=DEV)
    wp = torch.empty(k * k * C, device=DEV); L.dw_pack_weight(w.data_ptr(), wp.data_ptr(), C, k, k, st)
    M = B * Ho * Ho
    part = torch.zeros(L.dw_blocks(M) * C * 2, device=DEV)
    slab = torch.zeros(L.dw_wgrad_blocks(M) * k * k * C, device=DEV); gw = torch.empty_like(w)
    sb, sh, sw = ops.s3(x); ysb, ysh, ysw = ops.s3(y)
    tf = timeit(lambda: L.dwconv2d_fwd(dt, x.data_ptr(), sb, sh, sw, B, H, H, C, wp.data_ptr(), y.data_ptr(), C, Ho, Ho, k, k, s, p, part.data_ptr(), st))
    td = timeit(lambda: L.dwconv2d_bwd_data(dt, y.data_ptr(), ysb, ysh, ysw, B, Ho, Ho, C, wp.data_ptr(), dx.data_ptr(), C, H, H, k, k, s, p, st))
    tw = timeit(lambda: L.dwconv2d_bwd_weight(dt, x.data_ptr(), sb, sh, sw, B, H, H, C, y.data_ptr(), C, Ho, Ho, k, k, s, p, slab.data_ptr(), gw.data_ptr(), 0, st))
    mb = (B * H * H * C + M * C) * 2 / 1e6
    print(f"{H}x{H} C={C} k{k} s{s}: {mb:6.1f} MB  fwd {tf:6.1f} us ({mb / tf:.2f} TB/s)  dgrad {td:6.1f} us ({mb / td:.2f})  wgrad {tw:6.1f} us ({mb / tw:.2f})")
