"""BatchNorm + SiLU passes through the C ABI on the body's tensor shapes: time per launch and effective HBM bandwidth
(bytes = tensors read + written).   python tools/bn_bench.py [reps]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
y3d = importlib.import_module("yolov10-3d_amd")
ops = importlib.import_module("yolov10-3d_amd.ops")
DEV = torch.device("cuda:0")
BF16 = 1
SHAPES = [(320, 32), (160, 64), (160, 32), (80, 128), (80, 64), (80, 256), (40, 256), (40, 128), (40, 512), (20, 512), (20, 256), (80, 2048), (40, 2048), (20, 2048), (40, 1152), (80, 896)]


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    L, st, dt, B = y3d.lib(), ops.stream(), BF16, 32
    print(f"{'P x C':>18} {'MB':>7} {'fwd us':>8} {'TB/s':>6} {'reduce':>8} {'TB/s':>6} {'apply':>8} {'TB/s':>6} {'fin':>6} {'bfin':>6} blocks")
    tot = [0.0] * 5
    for H, C in SHAPES:
        P = B * H * H
        bf = torch.bfloat16
        y = torch.randn(P, C, device=DEV).to(bf)
        dz = torch.randn(P, C, device=DEV).to(bf)
        z = torch.empty_like(y)
        dy = torch.empty_like(y)
        f = [torch.rand(C, device=DEV) + 0.5 for _ in range(8)]
        nb = L.bn_bwd_blocks(P, C)
        part = torch.zeros(nb * C * 2, device=DEV)
        nbf = (P + 127) // 128
        partf = torch.rand(nbf * C * 2, device=DEV)
        mb = P * C * 2 / 1e6
        t_f = timeit(lambda: L.bn_act_fwd(dt, y.data_ptr(), C, f[0].data_ptr(), f[1].data_ptr(), 1, 0, None, 0, z.data_ptr(), C, P, C, st), reps)
        t_r = timeit(lambda: L.bn_act_bwd_reduce(dt, y.data_ptr(), C, dz.data_ptr(), C, None, 0, f[0].data_ptr(), f[1].data_ptr(), f[2].data_ptr(),
                                                 f[3].data_ptr(), 1, 0, part.data_ptr(), P, C, st), reps)
        t_a = timeit(lambda: L.bn_act_bwd_apply(dt, y.data_ptr(), C, dz.data_ptr(), C, None, 0, f[0].data_ptr(), f[1].data_ptr(), f[2].data_ptr(),
                                                f[3].data_ptr(), f[4].data_ptr(), f[5].data_ptr(), 1, 0, 1, dy.data_ptr(), C, None, 0, P, C, st), reps)
        t_n = timeit(lambda: L.bn_finalize(partf.data_ptr(), nbf, C, P, f[0].data_ptr(), f[1].data_ptr(), 1e-3, 0.03, f[6].data_ptr(), f[7].data_ptr(),
                                           f[2].data_ptr(), f[3].data_ptr(), f[4].data_ptr(), f[5].data_ptr(), st), reps)
        t_b = timeit(lambda: L.bn_bwd_finalize(part.data_ptr(), nb, C, P, f[6].data_ptr(), f[7].data_ptr(), 0, f[4].data_ptr(), f[5].data_ptr(), st), reps)
        print(f"{P:9d} x {C:<6d} {mb:7.1f} {t_f:8.1f} {2 * mb / t_f:6.2f} {t_r:8.1f} {2 * mb / t_r:6.2f} {t_a:8.1f} {3 * mb / t_a:6.2f} {t_n:6.1f} {t_b:6.1f} {nb}")
        for i, v in enumerate((t_f, t_r, t_a, t_n, t_b)):
            tot[i] += v
    print("sum us: fwd %.0f reduce %.0f apply %.0f finalize %.0f bwd_finalize %.0f" % tuple(tot))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "ab":  # row-wide slabs off / on (bn_act.hip: slab_width)
        y3d.lib().set_bn_wide_slabs(0)
        print("---- 64-channel slabs everywhere")
        main()
        y3d.lib().set_bn_wide_slabs(1)
        print("---- row-wide slabs for bf16, C >= 512")
    main()
