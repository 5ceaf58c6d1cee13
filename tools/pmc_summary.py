"""Merge two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE — they do not fit one pass, MI355X_MICROARCH.md "PMC slots") into a
per-kernel table and the headline kernel's per-launch HBM traffic (profiles/<round>_pmc_headline.json, read by bench.py).

usage: python tools/pmc_summary.py <fetch_dir> <write_dir> <out_prefix> [headline-substring] [headline-grid]
Counter unit is KB.  gfx950 correction (guide §HBM): FETCH_SIZE reports half of the bytes of wide coalesced reads -> doubled.
"""
import csv, glob, json, sys
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


fd, wd, out = sys.argv[1:4]
hl = sys.argv[4] if len(sys.argv) > 4 else "conv3x3_tile_kernel"
hg = int(sys.argv[5]) if len(sys.argv) > 5 else 0
F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
rows = []
for k in set(F) | set(W):
    f, w = F.get(k, [0, 1, 0]), W.get(k, [0, 1, 0])
    n = max(f[1], w[1])
    rows.append((f[2] + w[2], k, f[0] / max(f[1], 1) / 1e3, w[0] / max(w[1], 1) / 1e3, n, (f[2] / max(f[1], 1))))
rows.sort(reverse=True)
with open(out + "_summary.txt", "w") as o:
    o.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; per-dispatch averages in MB (counter unit KB);\n"
            "# FETCHx2 = gfx950 correction for wide coalesced reads (MI355X_MICROARCH.md, HBM section)\n")
    for t, k, f, w, n, us in rows[:40]:
        o.write(f"{k[0][:110]:110s} grid={k[1]:9d} n={n:4d} FETCH={f:8.1f} FETCHx2={2 * f:8.1f} WRITE={w:8.1f} avg_us={us:8.1f}\n")
cand = [r for r in rows if hl in r[1][0] and (hg == 0 or r[1][1] == hg)]
if cand:
    t, k, f, w, n, us = max(cand, key=lambda r: r[1][1] if hg == 0 else r[0])
    json.dump({"kernel": k[0], "grid": k[1], "launches": n, "fetch_mb_raw": round(f, 2), "fetch_mb_corrected": round(2 * f, 2), "write_mb": round(w, 2),
               "traffic_bytes_per_launch": int((2 * f + w) * 1e6), "avg_us_under_pmc": round(us, 1),
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; FETCH doubled per the gfx950 note"},
              open(out + "_headline.json", "w"), indent=1)
    print(open(out + "_headline.json").read())
