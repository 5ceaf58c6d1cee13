"""Merge two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE — they do not fit one pass, MI355X_MICROARCH.md "PMC slots") into a
per-kernel table and the headline kernel's per-launch HBM traffic (profiles/<round>_pmc_headline.json, read by bench.py).

usage: python tools/pmc_summary.py <fetch_dir> <write_dir> <out_prefix> [headline-substring] [headline-grid]
Counter unit is KiB (1024 B: a store-only probe of the headline kernel reads 1.004x its tensor bytes with 1024, 0.98x with 1000).
gfx950 correction (guide §HBM): FETCH_SIZE reports half of the bytes of wide coalesced reads -> doubled.
"""
import csv, glob, json, os, sys
from collections import defaultdict


def load(d, name):
    acc = defaultdict(lambda: [0.0, 0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name:
                continue
            k = (r["Kernel_Name"], int(r["Grid_Size"]))
            a = acc[k]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
            a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return acc


KB = 1024.0
fd, wd, out = sys.argv[1:4]
hl = sys.argv[4] if len(sys.argv) > 4 else "conv3x3_wide3_kernel<16, 0>"
hg = int(sys.argv[5]) if len(sys.argv) > 5 else 0
F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
rows = []
for k in set(F) | set(W):
    f, w = F.get(k, [0, 1, 0]), W.get(k, [0, 1, 0])
    n = max(f[1], w[1])
    rows.append((f[2] + w[2], k, f[0] / max(f[1], 1) * KB / 1e6, w[0] / max(w[1], 1) * KB / 1e6, n, (f[2] / max(f[1], 1))))
rows.sort(reverse=True)
with open(out + "_summary.txt", "w") as o:
    o.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; per-dispatch averages in MB (1e6 B; counter unit KiB);\n"
            "# FETCHx2 = gfx950 correction for wide coalesced reads (MI355X_MICROARCH.md, HBM section)\n")
    for t, k, f, w, n, us in rows[:40]:
        o.write(f"{k[0][:110]:110s} grid={k[1]:9d} n={n:4d} FETCH={f:8.1f} FETCHx2={2 * f:8.1f} WRITE={w:8.1f} avg_us={us:8.1f}\n")
# headline: the kernel is persistent (one grid size for every shape), so the fused head layer-2 forward launches are told
# apart by their traffic: the dispatches within 10 % of the kernel's largest counter value
def top(d, name):
    vals = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and hl in r["Kernel_Name"]:
                vals.append((float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"]))
    if not vals:
        return None
    m = max(v[0] for v in vals)
    sel = [v for v in vals if v[0] >= 0.9 * m]
    return sum(v[0] for v in sel) / len(sel) * KB / 1e6, sum(v[1] for v in sel) / len(sel), len(sel), sel[0][2]


tf, tw = top(fd, "FETCH_SIZE"), top(wd, "WRITE_SIZE")
if tf and tw:
    json.dump({"commit": os.environ.get("Y3D_COMMIT", "unknown"), "kernel": tf[3], "launches": min(tf[2], tw[2]), "fetch_mb_raw": round(tf[0], 2), "fetch_mb_corrected": round(2 * tf[0], 2), "write_mb": round(tw[0], 2),
               "traffic_bytes_per_launch": int((2 * tf[0] + tw[0]) * 1e6), "avg_us_under_pmc": round((tf[1] + tw[1]) / 2, 1),
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; FETCH doubled per the gfx950 note; "
                         "dispatches within 10 % of the kernel's largest counter value (the fused head layer-2 forward launches)"},
              open(out + "_headline.json", "w"), indent=1)
    print(open(out + "_headline.json").read())
