import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import yolov10_3d_amd as y3d
from test_hip_fp8 import conv_fp8
DEV = "cuda"
torch.manual_seed(0)
B, Cin, Cout, g, H, W = 4, 128, 128, 1, 8, 16
x = (torch.randint(-3, 4, (B, Cin, H, W)).float()).to(torch.bfloat16)
def run(w, label):
    ref = torch.nn.functional.conv2d(x.float().double(), w.double(), None, 1, 1, 1, g).float()
    y, part = conv_fp8(x.to(DEV), w.to(DEV), g)
    d = (y.float().cpu() - ref.to(torch.bfloat16).float()).abs()
    bad = d > 0
    print(f"{label}: wrong {int(bad.sum())}/{bad.numel()}  by image {bad.sum((1,2,3)).tolist()} by row {bad.sum((0,1,3)).tolist()} by col {bad.sum((0,1,2)).tolist()} by cout/16 {bad.reshape(B, Cout//16, 16, H, W).sum((0,2,3,4)).tolist()}")
    return y, ref
for tap in range(9):
    w = torch.zeros(Cout, Cin, 3, 3)
    w[:, :, tap // 3, tap % 3] = torch.randint(-2, 3, (Cout, Cin)).float()
    run(w, f"tap {tap} all channels")
for cb in range(4):
    w = torch.zeros(Cout, Cin, 3, 3)
    w[:, cb * 32:(cb + 1) * 32, 1, 1] = torch.randint(-2, 3, (Cout, 32)).float()
    run(w, f"centre tap, channels {cb*32}..{cb*32+31}")
w = torch.zeros(Cout, Cin, 3, 3); w[:, 0, 1, 1] = 1.0
y, ref = run(w, "centre tap, channel 0, weight 1")
print(y[0, 0].float().cpu()[:3, :8]); print(ref[0, 0][:3, :8])
