# Round-N profile set of the default bench command, run ON THE GPU BOX:  bash tools/prof_all.sh r02 <commit>
# Writes under gpurun_out/ (scratch); the summaries to keep are copied into profiles/ afterwards (tracked).
set -e
TAG=${1:-r02}
export Y3D_COMMIT=${2:-unknown}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
python3 $R/bench.py > $O/${TAG}_bench_default.log 2>&1
echo bench done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o run -- python3 $R/bench.py --steps 4 --warmup 1 --infer-steps 0 --no-cpu-baseline > $O/${TAG}_bench_under_rocprof.log 2>&1
echo stats done
python3 $R/tools/prof_summary.py $O/${TAG}_prof 5 60 > $O/${TAG}_bench_kernel_stats_summary.txt
cp $(find $O/${TAG}_prof -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_f -o run -- python3 $R/bench.py --steps 2 --warmup 1 --infer-steps 0 --no-cpu-baseline > $O/${TAG}_pmc_f.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_w -o run -- python3 $R/bench.py --steps 2 --warmup 1 --infer-steps 0 --no-cpu-baseline > $O/${TAG}_pmc_w.log 2>&1
echo write done
cd $R
python3 tools/pmc_summary.py gpurun_out/${TAG}_pmc_f gpurun_out/${TAG}_pmc_w gpurun_out/${TAG}_pmc
python3 tools/layer_report.py yolov10s_3D.yaml 640 32 > gpurun_out/${TAG}_layer_report.txt 2>&1
tail -1 gpurun_out/${TAG}_bench_default.log | cut -c1-400
