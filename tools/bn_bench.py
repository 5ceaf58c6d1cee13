"""Bandwidth of the BatchNorm/SiLU elementwise kernels through the C ABI on the shapes that carry the S-3D step.
usage: python tools/bn_bench.py   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import yolov10_3d_amd as y3d
from yolov10_3d_amd import ops

L = y3d.lib()
dev = torch.device("cuda", 0)
st = ops.stream()


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (P, C) in [(32 * 80 * 80, 2048), (32 * 40 * 40, 2048), (32 * 160 * 160, 64), (32 * 80 * 80, 128), (32 * 40 * 40, 256), (32 * 20 * 20, 512)]:
    y = torch.randn(P, C, device=dev).bfloat16()
    dz = torch.randn(P, C, device=dev).bfloat16()
    z = torch.empty_like(y)
    dy = torch.empty_like(y)
    f = [torch.rand(C, device=dev) + 0.5 for _ in range(6)]
    nb = L.bn_bwd_blocks(P, C)
    part = torch.empty(nb, C, 2, device=dev)
    eb = P * C * 2 / 1e9
    t = timeit(lambda: L.bn_act_fwd(1, y.data_ptr(), C, f[0].data_ptr(), f[1].data_ptr(), 1, 0, None, 0, z.data_ptr(), C, P, C, st))
    t2 = timeit(lambda: L.bn_act_bwd_reduce(1, y.data_ptr(), C, dz.data_ptr(), C, None, 0, f[0].data_ptr(), f[1].data_ptr(), f[2].data_ptr(), f[3].data_ptr(), 1, 0,
                                            part.data_ptr(), P, C, st))
    t3 = timeit(lambda: L.bn_act_bwd_apply(1, y.data_ptr(), C, dz.data_ptr(), C, None, 0, f[0].data_ptr(), f[1].data_ptr(), f[2].data_ptr(), f[3].data_ptr(),
                                           f[4].data_ptr(), f[5].data_ptr(), 1, 0, 1, dy.data_ptr(), C, None, 0, P, C, st))
    # the same kernels without the activation (act = 0): how much of the time is SiLU arithmetic rather than HBM traffic
    u = timeit(lambda: L.bn_act_fwd(1, y.data_ptr(), C, f[0].data_ptr(), f[1].data_ptr(), 0, 0, None, 0, z.data_ptr(), C, P, C, st))
    u2 = timeit(lambda: L.bn_act_bwd_reduce(1, y.data_ptr(), C, dz.data_ptr(), C, None, 0, f[0].data_ptr(), f[1].data_ptr(), f[2].data_ptr(), f[3].data_ptr(), 0, 0,
                                            part.data_ptr(), P, C, st))
    u3 = timeit(lambda: L.bn_act_bwd_apply(1, y.data_ptr(), C, dz.data_ptr(), C, None, 0, f[0].data_ptr(), f[1].data_ptr(), f[2].data_ptr(), f[3].data_ptr(),
                                           f[4].data_ptr(), f[5].data_ptr(), 0, 0, 1, dy.data_ptr(), C, None, 0, P, C, st))
    print(f"   no-act: fwd {u * 1e3:7.1f} us | bwd_reduce {u2 * 1e3:7.1f} us | bwd_apply {u3 * 1e3:7.1f} us")
    print(f"P={P:7d} C={C:5d} ({eb * 1e3:7.1f} MB/tensor): fwd {t * 1e3:7.1f} us {2 * eb / t * 1e3:7.0f} GB/s | bwd_reduce {t2 * 1e3:7.1f} us {2 * eb / t2 * 1e3:7.0f} GB/s | "
          f"bwd_apply {t3 * 1e3:7.1f} us {3 * eb / t3 * 1e3:7.0f} GB/s", flush=True)
