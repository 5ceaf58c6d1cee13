# PMC counters of the kernels matching a regex in a 2-step bench run (run ON THE GPU BOX):  bash tools/pmc_kernel.sh <tag> <kernel regex> <counters...>
set -e
T=$1; RE=$2; shift 2
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --kernel-include-regex "$RE" --output-format csv -d $O/${T}_pmc -o run -- python3 $R/bench.py --steps 2 --warmup 1 --infer-steps 0 --no-cpu-baseline --no-train-graph $Y3D_BENCH_ARGS > $O/${T}_pmc.log 2>&1
python3 - "$O/${T}_pmc" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = (r["Kernel_Name"][:70], r["Grid_Size"])
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
for k in sorted(acc):
    print(k[0], "grid", k[1])
    for c, v in sorted(acc[k].items()):
        print(f"    {c:28s} {v / n[(k, c)]:16.1f} per dispatch ({n[(k, c)]} dispatches)")
PY
rm -rf $O/${T}_pmc
