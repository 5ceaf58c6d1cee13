# Round-4 measurement set, run ON THE GPU BOX in pieces (each a gpurun call):  bash tools/prof_r04.sh <part>
#   bench  : default bench + 300-step log;   stats : rocprofv3 kernel stats of the default command;   pmc : FETCH / WRITE / MFMA-busy passes
#   fp8    : X-3D fp8 line, its kernel stats and layer report;   reports : layer reports of the other configurations
set -e
PART=${1:-bench}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
T=r04
export Y3D_COMMIT=${2:-r04}
cd /tmp && export TMPDIR=/tmp
case $PART in
bench)
  python3 $R/bench.py > $O/${T}_bench_default.log 2>&1   # defaults: 100 timed steps, 10 warm-up, 40 eval batches
  python3 $R/bench.py --steps 300 --warmup 5 --no-cpu-baseline --infer-steps 2 > $O/${T}_bench_300steps.log 2>&1
  tail -1 $O/${T}_bench_default.log | cut -c1-300; tail -1 $O/${T}_bench_300steps.log | cut -c1-300 ;;
stats)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof -o run -- python3 $R/bench.py --steps 4 --warmup 1 --infer-steps 0 --no-cpu-baseline --no-train-graph > $O/${T}_bench_under_rocprof.log 2>&1
  python3 $R/tools/prof_summary.py $O/${T}_prof 5 70 > $O/${T}_bench_kernel_stats_summary.txt
  cp $(find $O/${T}_prof -name "*kernel_stats.csv" | head -1) $O/${T}_bench_kernel_stats.csv
  head -12 $O/${T}_bench_kernel_stats_summary.txt ;;
pmc)
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${T}_pmc_f -o run -- python3 $R/bench.py --steps 2 --warmup 1 --infer-steps 0 --no-cpu-baseline --no-train-graph > $O/${T}_pmc_f.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${T}_pmc_w -o run -- python3 $R/bench.py --steps 2 --warmup 1 --infer-steps 0 --no-cpu-baseline --no-train-graph > $O/${T}_pmc_w.log 2>&1
  cd $R && python3 tools/pmc_summary.py gpurun_out/${T}_pmc_f gpurun_out/${T}_pmc_w gpurun_out/${T}_pmc ;;
fp8)
  cd $R
  python3 bench.py --model yolov10x_3D.yaml --weights fp8 --batch 16 --no-cpu-baseline > $O/${T}_bench_x3d_fp8.log 2>&1
  python3 bench.py --weights fp8 --no-cpu-baseline > $O/${T}_bench_s3d_fp8.log 2>&1
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_fp8_prof -o run -- python3 $R/bench.py --weights fp8 --steps 4 --warmup 1 --infer-steps 0 --no-cpu-baseline --no-train-graph > $O/${T}_s3d_fp8_under_rocprof.log 2>&1
  python3 $R/tools/prof_summary.py $O/${T}_fp8_prof 5 40 > $O/${T}_s3d_fp8_kernel_stats_summary.txt
  cd $R
  python3 tools/layer_report.py yolov10s_3D.yaml 640 32 40 train fp8 > $O/${T}_layer_report_s3d_fp8.txt 2>&1
  python3 tools/layer_report.py yolov10x_3D.yaml 640 16 40 train fp8 > $O/${T}_layer_report_x3d_fp8.txt 2>&1
  tail -1 $O/${T}_bench_x3d_fp8.log | cut -c1-200; tail -1 $O/${T}_bench_s3d_fp8.log | cut -c1-200 ;;
reports)
  cd $R
  python3 bench.py --model yolov10m_3D.yaml --no-cpu-baseline > $O/${T}_bench_m3d.log 2>&1
  python3 bench.py --model yolov10l.yaml --imgsz 1280 --batch 8 --no-cpu-baseline > $O/${T}_bench_l2d_1280.log 2>&1
  python3 bench.py --model yolov10n_3D.yaml --no-cpu-baseline > $O/${T}_bench_n3d.log 2>&1
  python3 tools/layer_report.py yolov10s_3D.yaml 640 32 > $O/${T}_layer_report.txt 2>&1
  python3 tools/layer_report.py yolov10s_3D.yaml 640 32 400 eval > $O/${T}_layer_report_eval.txt 2>&1
  for f in m3d l2d_1280 n3d; do tail -1 $O/${T}_bench_$f.log | cut -c1-160; done ;;
esac
