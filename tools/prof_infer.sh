# eval-loop kernel profile, run ON THE GPU BOX:  bash tools/prof_infer.sh [tag]    (writes gpurun_out/<tag>_infer_kernel_stats_summary.txt)
set -e
T=${1:-r03b}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_infer_prof -o run -- python3 $R/tools/infer_loop.py 10 > $O/${T}_infer_under_rocprof.log 2>&1
python3 $R/tools/prof_summary.py $O/${T}_infer_prof 13 60 > $O/${T}_infer_kernel_stats_summary.txt
rm -rf $O/${T}_infer_prof
