"""Precision study for the Winograd route north_star names: F(2x2, 3x3) with bf16 MFMA operands against the
direct bf16 convolution, on the head's 128 -> 128 3x3 shape, CPU arithmetic emulating the kernels' roundings exactly:

  direct   : x, w rounded to bf16, products and sums in fp32 (the MFMA), output rounded to bf16         (what conv3x3_wide computes)
  winograd : V = B^T d B from the bf16 input (exact in fp32) ROUNDED TO bf16, U = G g G^T from the fp32 master weights ROUNDED TO bf16
             (both are MFMA operands), M = sum_c U.V in fp32, Y = A^T M A in fp32, output rounded to bf16
  reference: fp64 convolution of the UNROUNDED x, w

Reports relative L2 errors and their ratio.  The integration bar (tests/test_hip_bench_path.py RATIO = 1.5 against the reference's own
autocast run, where the direct kernels sit at ~0.85) leaves room for a factor ~1.75 over the direct form.
    python tools/probe/winograd_precision.py [seed]"""
import sys

import torch
import torch.nn.functional as F


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def winograd_f2x2_3x3(x, w, round_operands=True):
    """x: (B, C, H, W) fp32 (already bf16-valued), w: (K, C, 3, 3) fp32 master.  H, W even.  -> (B, K, H, W) fp32"""
    B, C, H, W = x.shape
    K = w.shape[0]
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
    Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
    At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
    U = torch.einsum("ij,kcjl,ml->kcim", G, w, G)                     # (K, C, 4, 4)
    xp = F.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                        # (B, C, H/2, W/2, 4, 4)
    V = torch.einsum("ij,bcyxjl,ml->bcyxim", Bt, tiles, Bt)           # exact in fp32: sums of four bf16 values
    if round_operands:
        U, V = bf(U), bf(V)
    M = torch.einsum("kcim,bcyxim->bkyxim", U.double(), V.double()).float()  # products exact, fp32-like accumulation (double here: upper bound on quality)
    Y = torch.einsum("ij,bkyxjl,ml->bkyxim", At, M, At)               # (B, K, H/2, W/2, 2, 2)
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B, K, H, W)


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    g = torch.Generator().manual_seed(seed)
    B, C, K, H, W = 2, 128, 128, 40, 40
    pre = torch.randn(B, C, H, W, generator=g)
    x = bf(pre * torch.sigmoid(pre))                                   # a post-SiLU activation, as the head's second layer sees it
    w = torch.randn(K, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
    ref = F.conv2d(x.double(), w.double(), padding=1)
    direct = bf(F.conv2d(x.double(), bf(w).double(), padding=1).float())
    wino = bf(winograd_f2x2_3x3(x, w))
    wino_exact = winograd_f2x2_3x3(x, w, round_operands=False)
    rel = lambda a: float((a.double() - ref).norm() / ref.norm())
    e_d, e_w, e_x = rel(direct), rel(wino), rel(wino_exact)
    print(f"seed {seed}: relative L2 error vs fp64 -- direct bf16 {e_d:.3e} | Winograd F(2x2,3x3) bf16 operands {e_w:.3e} | Winograd fp32 operands {e_x:.3e}")
    print(f"          Winograd / direct = {e_w / e_d:.2f}x   (bar: <= ~1.75x to stay inside 1.5x the reference's autocast distance)")
    # input gradient has the same structure (flipped taps); weight gradient would need F(3x3, 2x2): larger transforms, more loss


if __name__ == "__main__":
    main()
