set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
python3 $R/bench.py > $O/r01_bench_default.log 2>&1
echo bench done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_final -o run -- python3 $R/bench.py --steps 4 --warmup 1 --infer-steps 0 --no-cpu-baseline > $O/r01_bench_under_rocprof.log 2>&1
echo stats done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o run -- python3 $R/bench.py --steps 2 --warmup 1 --infer-steps 0 --no-cpu-baseline > $O/pmc_f.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o run -- python3 $R/bench.py --steps 2 --warmup 1 --infer-steps 0 --no-cpu-baseline > $O/pmc_w.log 2>&1
echo write done
cd $R
python3 tools/prof_summary.py gpurun_out/prof_final 5 45 > gpurun_out/r01_bench_kernel_stats_summary.txt
python3 tools/pmc_summary.py gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/r01_pmc
python3 tools/layer_report.py yolov10s_3D.yaml 640 32 > gpurun_out/r01_layer_report.txt 2>&1
tail -1 gpurun_out/r01_bench_default.log | cut -c1-250
