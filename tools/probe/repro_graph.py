import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
import yolov10_3d_amd as y3d
import test_hip_bench_path as T
T.test_cross_layer_concat_placement_is_bit_identical_and_saves_the_copies("yolov10n.yaml", 96)
print("placement test done", flush=True)
import bench
bench.main(["--model", "yolov10l.yaml", "--imgsz", "640", "--batch", "2", "--steps", "1", "--warmup", "1", "--infer-steps", "1", "--no-cpu-baseline"])
