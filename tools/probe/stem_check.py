"""fused training stem against y3d_stem_im2col: column tensor and raw output, mismatch positions (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import yolov10_3d_amd as y3d
from yolov10_3d_amd import ops
B, H, W, C = 2, 258, 130, 16
if len(sys.argv) > 4:
    B, H, W, C = [int(v) for v in sys.argv[1:5]]
dev = "cuda"
L, st = ops.lib(), ops.stream()
torch.manual_seed(0)
x = torch.rand(B, 3, H, W, device=dev)
Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
xc0 = torch.zeros(B * Ho * Wo, 32, dtype=torch.bfloat16, device=dev)
L.stem_im2col(1, x.data_ptr(), xc0.data_ptr(), B, H, W, Ho, Wo, st)
wcol = torch.zeros(C, 32, device=dev)
wcol[:, :27] = torch.randn(C, 27, device=dev)
y = torch.zeros(B * Ho * Wo, C, dtype=torch.bfloat16, device=dev)
xc1 = torch.full((B * Ho * Wo, 32), 7.0, dtype=torch.bfloat16, device=dev)
rows = L.stem_conv_train_rows(B, H, W)
part = torch.zeros(rows, C, 2, device=dev)
L.stem_conv_train(x.data_ptr(), 0, wcol.data_ptr(), y.data_ptr(), C, xc1.data_ptr(), part.data_ptr(), B, H, W, C, st)
torch.cuda.synchronize()
bad = (xc0 != xc1).nonzero()
print("column tensor mismatches:", bad.shape[0], "of", xc0.numel())
for r, c in bad[:12].tolist():
    pix = r % (Ho * Wo)
    print(f"  pixel b={r // (Ho * Wo)} oy={pix // Wo} ox={pix % Wo} col={c}: im2col {float(xc0[r, c]):.4f} fused {float(xc1[r, c]):.4f}")
yr = (xc0.float() @ wcol.bfloat16().float().t())
print("y max err:", float((y.float() - yr).abs().max()), "of", float(yr.abs().max()))
s = part.double().sum(0)
print("sum err:", float((s[:, 0] - y.double().sum(0)).abs().max()), "sq err:", float((s[:, 1] - (y.double() ** 2).sum(0)).abs().max() / (y.double() ** 2).sum(0).max()))
