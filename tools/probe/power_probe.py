"""Socket power and shader clock while the headline conv launch (16 groups of 128 -> 128 @80x80, B = 32, bf16) runs back to back on random
and on all-zero operands: is the 26 % clock difference of profiles/r03_pmc_mfma_busy.txt a POWER limit?   python tools/probe/power_probe.py   (GPU box)"""
import json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import yolov10_3d_amd as y3d
from yolov10_3d_amd import ops

L, st, dt, bf = y3d.lib(), ops.stream(), 1, torch.bfloat16
B, H, W, C, G = 32, 80, 80, 2048, 16
w = (torch.randn(C, C // G, 3, 3, device="cuda") * 0.05)
wp = torch.empty(C * 9 * (C // G), dtype=bf, device="cuda")
L.pack_weight_fwd(dt, w.data_ptr(), wp.data_ptr(), C, C // G, C // G, 3, 3, st)
y = ops.nhwc_empty(B, C, H, W, bf, "cuda")
rows = L.conv2d_stat_rows(dt, B, H, W, C, C, G, 3, 3, 1, 1)
part = torch.zeros(rows, C, 2, device="cuda")
samples = []
stop = False


def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
            d = json.loads(out)
            c = d.get("card0", next(iter(d.values())))
            samples.append({k: v for k, v in c.items() if "ower" in k or "sclk" in k.lower()})
        except Exception as e:  # noqa
            samples.append({"error": str(e)[:80]})
        time.sleep(0.25)


for name, x in (("random", torch.randn(B, C, H, W, device="cuda")), ("zeros", torch.zeros(B, C, H, W, device="cuda")), ("idle", None)):
    samples.clear()
    stop = False
    t = threading.Thread(target=poll)
    if x is not None:
        xin = ops.nhwc_empty(B, C, H, W, bf, "cuda")
        xin.copy_(x)
        if name == "zeros":
            wz = torch.zeros_like(wp)
        sb, sh, sw = ops.s3(xin)
    t.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 4.0:
        if x is None:
            time.sleep(0.1)
            continue
        for _ in range(50):
            L.conv2d_fwd(dt, xin.data_ptr(), sb, sh, sw, B, H, W, C, (wz if name == "zeros" else wp).data_ptr(), None, y.data_ptr(), C, H, W, C, G, 3, 3, 1, 1, part.data_ptr(), st)
        torch.cuda.synchronize()
        n += 50
    dt_s = time.perf_counter() - t0
    stop = True
    t.join()
    us = 1e6 * dt_s / max(n, 1)
    print(f"{name}: {n} launches, {us:.1f} us each, {2.0 * B * H * W * C * (C // G) * 9 / (us * 1e-6) / 1e12 if n else 0:.0f} TFLOP/s; rocm-smi samples (last 6):")
    for s in samples[-6:]:
        print("   ", s)
