# kernel profile of the training step of one model, run ON THE GPU BOX:  bash tools/prof_model.sh TAG bench-args...
set -e
T=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof -o run -- python3 $R/bench.py --steps 4 --warmup 1 --infer-steps 0 --no-cpu-baseline --no-train-graph "$@" > $O/${T}_under_rocprof.log 2>&1
python3 $R/tools/prof_summary.py $O/${T}_prof 5 60 > $O/${T}_kernel_stats_summary.txt
rm -rf $O/${T}_prof
