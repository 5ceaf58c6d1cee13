# Round-3 measurement set, run ON THE GPU BOX:  bash tools/prof_r03.sh     (writes under gpurun_out/; the summaries are copied into profiles/)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
T=r03
python3 $R/bench.py > $O/${T}_bench_default.log 2>&1
echo bench done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof -o run -- python3 $R/bench.py --steps 4 --warmup 1 --infer-steps 0 --no-cpu-baseline --no-train-graph > $O/${T}_bench_under_rocprof.log 2>&1
python3 $R/tools/prof_summary.py $O/${T}_prof 5 70 > $O/${T}_bench_kernel_stats_summary.txt
cp $(find $O/${T}_prof -name "*kernel_stats.csv" | head -1) $O/${T}_bench_kernel_stats.csv
echo stats done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_infer_prof -o run -- python3 $R/tools/infer_loop.py 10 > $O/${T}_infer_under_rocprof.log 2>&1
python3 $R/tools/prof_summary.py $O/${T}_infer_prof 13 45 > $O/${T}_infer_kernel_stats_summary.txt
echo infer stats done
cd $R
python3 bench.py --model yolov10m_3D.yaml --no-cpu-baseline > gpurun_out/${T}_bench_m3d.json 2> gpurun_out/${T}_bench_m3d.log
python3 bench.py --model yolov10x_3D.yaml --weights fp8 --batch 16 --no-cpu-baseline > gpurun_out/${T}_bench_x3d_fp8.json 2> gpurun_out/${T}_bench_x3d_fp8.log
python3 bench.py --model yolov10l.yaml --imgsz 1280 --batch 8 --no-cpu-baseline > gpurun_out/${T}_bench_l2d_1280.json 2> gpurun_out/${T}_bench_l2d_1280.log
python3 bench.py --model yolov10n_3D.yaml --no-cpu-baseline > gpurun_out/${T}_bench_n3d.json 2> gpurun_out/${T}_bench_n3d.log
echo configs done
python3 tools/layer_report.py yolov10s_3D.yaml 640 32 > gpurun_out/${T}_layer_report.txt 2>&1
python3 tools/layer_report.py yolov10m_3D.yaml 640 32 > gpurun_out/${T}_layer_report_m3d.txt 2>&1
python3 tools/layer_report.py yolov10x_3D.yaml 640 16 > gpurun_out/${T}_layer_report_x3d.txt 2>&1
python3 tools/layer_report.py yolov10l.yaml 1280 8 > gpurun_out/${T}_layer_report_l2d1280.txt 2>&1
python3 tools/layer_report.py yolov10s_3D.yaml 640 32 400 eval > gpurun_out/${T}_layer_report_eval.txt 2>&1
echo layer reports done
python3 -m pytest tests/test_hip_bench_path.py -m gpu -q -s -k autocast 2>&1 | grep "yardstick" > gpurun_out/${T}_bf16_yardstick.txt
tail -1 gpurun_out/${T}_bench_default.log | cut -c1-600
