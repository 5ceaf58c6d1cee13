"""Precision study for a true fp8 x fp8 MFMA forward: the head's 128 -> 128 3x3 convolution on a post-SiLU
input, CPU arithmetic emulating the roundings:

  bf16        : x, w in bf16, fp32 accumulation, bf16 output                                  (the shipped kernels)
  fp8w        : w as e4m3 codes with a power-of-two scale per output channel, x in bf16       (the shipped `--weights fp8` mode)
  fp8 x fp8   : x ALSO in e4m3 - (a) one scale per tensor, (b) one power-of-two scale per 32 consecutive input channels of a pixel
                (the block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 form, e8m0 scales)
against the fp64 convolution of the unrounded operands.  The integration bar of this repository is 1.5 x the distance of the
reference's own bf16 autocast run (tests/test_hip_bench_path.py), where the bf16 kernels sit at ~0.85.
    python tools/probe/fp8_precision.py [seed]"""
import sys

import torch
import torch.nn.functional as F

E4M3_MAX = 448.0


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def e4m3(x):
    return x.to(torch.float8_e4m3fn).to(torch.float32)


def pow2_scale(absmax):
    return torch.exp2(torch.ceil(torch.log2(absmax.clamp(min=1e-30) / E4M3_MAX)))


def q_rows(w):  # per output channel
    s = pow2_scale(w.abs().amax(dim=(1, 2, 3), keepdim=True))
    return e4m3(w / s) * s


def q_tensor(x):
    s = pow2_scale(x.abs().amax())
    return e4m3(x / s) * s


def q_block32(x):  # (B, C, H, W): blocks of 32 channels of one pixel share a scale
    B, C, H, W = x.shape
    xb = x.view(B, C // 32, 32, H, W)
    s = pow2_scale(xb.abs().amax(dim=2, keepdim=True))
    return (e4m3(xb / s) * s).view(B, C, H, W)


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    g = torch.Generator().manual_seed(seed)
    B, C, K, H, W = 2, 128, 128, 40, 40
    pre = torch.randn(B, C, H, W, generator=g)
    x = pre * torch.sigmoid(pre)
    w = torch.randn(K, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
    ref = F.conv2d(x.double(), w.double(), padding=1)
    rel = lambda a: float((a.double() - ref).norm() / ref.norm())
    conv = lambda a, b: bf(F.conv2d(a.double(), b.double(), padding=1).float())
    rows = [("bf16 x bf16 (shipped kernels)", rel(conv(bf(x), bf(w)))),
            ("fp8 weights x bf16 activations (shipped --weights fp8)", rel(conv(bf(x), q_rows(w)))),
            ("fp8 x fp8, one activation scale per tensor", rel(conv(q_tensor(x), q_rows(w)))),
            ("fp8 x fp8, block-scaled activations (32 channels)", rel(conv(q_block32(x), q_rows(w))))]
    base = rows[0][1]
    for name, e in rows:
        print(f"seed {seed}: {name:58s} relative L2 error {e:.3e}  ({e / base:5.1f} x bf16)")


if __name__ == "__main__":
    main()
