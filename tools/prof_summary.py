"""Per-kernel AND per-launch-slot time of a profiled bench run.

    python tools/prof_summary.py <rocprofv3 output dir> <steps in the trace> [top N]

Reads rocprofv3's `*kernel_trace.csv` (one row per dispatch).  Two tables:
  1. by kernel name (what `--stats` prints), per training step;
  2. by (kernel name, grid size, launch slot): the k-th dispatch of a kernel inside a step always has the same shape (the step is a
     fixed launch sequence), so dispatch index modulo launches-per-step isolates ONE call site.  Persistent kernels use one grid size
     for every shape (conv3x3_wide: one workgroup per CU) — without the slot the 16 launches per step of `conv3x3_wide3_kernel<16, 0>`
     average into one meaningless row.  The slowest slot of that kernel is the roofline launch of bench.py
     (fused head layer 2 at P3, 966 GFLOP); its average here must agree with bench.py's live HIP-event figure.
Falls back to `*kernel_stats.csv` (table 1 only) when the trace is absent."""
import csv
import glob
import sys
from collections import defaultdict


def col(row, *names):
    for n in names:
        if n in row:
            return row[n]
    raise KeyError(names)


def main():
    d = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    traces = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if not traces:
        f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
        rows = list(csv.DictReader(open(f)))
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        print(f"total {tot / 1e6:.1f} ms  ({tot / 1e6 / steps:.2f} ms/step over {steps} steps)  [kernel_stats.csv only: no per-slot table]")
        for r in rows[:top]:
            print(f"{float(r['TotalDurationNs']) / 1e6 / steps:8.2f} ms/step {float(r['Percentage']):6.2f}% n={int(r['Calls']) / steps:7.1f} "
                  f"avg={float(r['AverageNs']) / 1e3:9.1f}us  {r['Name'][:100]}")
        return
    disp = []
    for f in traces:
        for r in csv.DictReader(open(f)):
            name = col(r, "Kernel_Name", "Name")
            t0, t1 = int(col(r, "Start_Timestamp")), int(col(r, "End_Timestamp"))
            grid = int(col(r, "Grid_Size_X", "Grid_Size")) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
            disp.append((t0, name, grid, t1 - t0))
    disp.sort()
    # steady-state window: a kernel that runs exactly once per step (the fused SGD update) delimits the steps; everything before its
    # first dispatch (model upload, EMA deep copy, first-step packing: thousands of one-time copies) is dropped, so are partial steps
    marker = [i for i, x in enumerate(disp) if "mt_sgd_kernel" in x[1] or "mt_adamw_kernel" in x[1]]
    if len(marker) >= 2:
        disp = disp[marker[0] + 1: marker[-1] + 1]
        steps = float(len(marker) - 1)
        print(f"# steady-state window: {len(marker) - 1} steps between optimizer updates (set-up and the first step are excluded)")
    tot = sum(x[3] for x in disp)
    print(f"# {len(disp)} dispatches, total {tot / 1e6:.1f} ms  ({tot / 1e6 / steps:.2f} ms/step over {steps:g} steps)")
    by_name = defaultdict(lambda: [0, 0])
    for _, name, grid, dt in disp:
        by_name[name][0] += dt
        by_name[name][1] += 1
    print("# ---- by kernel ----")
    for name, (t, n) in sorted(by_name.items(), key=lambda kv: -kv[1][0])[:top]:
        print(f"{t / 1e6 / steps:8.2f} ms/step {100 * t / tot:6.2f}% n={n / steps:7.1f} avg={t / n / 1e3:9.1f}us  {name[:100]}")
    # launch slots: only kernels whose dispatch count is a multiple of the step count form a fixed per-step sequence
    seq = defaultdict(list)
    for _, name, grid, dt in disp:
        seq[(name, grid)].append(dt)
    slots = []
    isteps = int(round(steps))
    for (name, grid), dts in seq.items():
        if isteps >= 1 and len(dts) % isteps == 0 and len(dts) >= isteps:
            per = len(dts) // isteps
            for s in range(per):
                v = dts[s::per]
                slots.append((sum(v) / len(v), name, grid, s, per, min(v), max(v), len(v)))
        else:
            slots.append((sum(dts) / len(dts), name, grid, -1, 0, min(dts), max(dts), len(dts)))
    slots.sort(reverse=True)
    print("# ---- by (kernel, grid, launch slot within the step): avg / min / max us over the steps ----")
    for avg, name, grid, s, per, lo, hi, n in slots[:top]:
        slot = f"slot {s:3d}/{per:<3d}" if s >= 0 else "unsynchronised"
        print(f"{avg / 1e3:9.1f}us  min {lo / 1e3:9.1f}  max {hi / 1e3:9.1f}  n={n:3d}  grid={grid:9d}  {slot}  {name[:90]}")
    wide = [x for x in slots if "conv3x3_wide3_kernel<16, 0>" in x[1] and x[3] >= 0]
    if wide:
        avg, name, grid, s, per, lo, hi, n = wide[0]
        print(f"# headline launch (slowest slot of conv3x3_wide3_kernel<16, 0>, the fused head layer-2 forward at P3): avg {avg / 1e3:.1f} us over {n} launches"
              f" = {966.3676416e9 / (avg * 1e-9) / 1e12:.1f} TFLOP/s at 966.37 GFLOP per launch (B=32, 640x640; profiled passes clock lower than the un-profiled bench)")


if __name__ == "__main__":
    main()
