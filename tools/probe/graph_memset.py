"""does a hipMemsetAsync recorded into a hipGraph run again on every replay?  (y3d_pad_targets = memset(n_used) + a kernel that atomicMax'es)"""
import sys, torch
sys.path.insert(0, ".")
import yolov10_3d_amd as y3d
from yolov10_3d_amd import ops
L = ops.lib()
dev = "cuda"
rows = torch.zeros(32, 18, device=dev)
rows[:, 0] = -1
def fill(n):
    rows[:, 0] = -1
    rows[:n, 0] = 0
out = torch.empty(2, 64, 17, device=dev)
n_used = torch.full((2,), 99, dtype=torch.int32, device=dev)
fill(7)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    L.pad_targets(rows.data_ptr(), 32, 17, 2, 64, 1.0, 1.0, out.data_ptr(), n_used.data_ptr(), ops.stream())
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print("eager:", n_used.tolist())
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    L.pad_targets(rows.data_ptr(), 32, 17, 2, 64, 1.0, 1.0, out.data_ptr(), n_used.data_ptr(), ops.stream())
for n in (7, 3, 5, 1):
    fill(n)
    g.replay()
    torch.cuda.synchronize()
    print(f"replay with {n} boxes: n_used", n_used.tolist())
