# kernel stats of a 4-step bench run, filtered:  bash tools/prof_quick.sh <tag> <grep pattern>   (run ON THE GPU BOX)
set -e
T=${1:-q}
PAT=${2:-projg}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof -o run -- python3 $R/bench.py --steps 4 --warmup 1 --infer-steps 0 --no-cpu-baseline --no-train-graph > $O/${T}_under_rocprof.log 2>&1
python3 $R/tools/prof_summary.py $O/${T}_prof 5 70 > $O/${T}_kernel_stats_summary.txt
rm -rf $O/${T}_prof
head -3 $O/${T}_kernel_stats_summary.txt | tail -2
grep -E "$PAT" $O/${T}_kernel_stats_summary.txt | head -${3:-16} | cut -c1-150
