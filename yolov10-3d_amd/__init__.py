"""yolov10-3d_amd — MI355X (gfx950) native YOLOv10 / YOLOv10-3D hot path.

    import yolov10_3d_amd as y3d
    model = y3d.YOLOv10_3DDetectionModel("yolov10s_3D.yaml").cuda()
    loss, items = model(batch_dict)          # train:  reference nn/tasks.py:93-95 call convention
    preds = model.eval()(img)                # eval

HIP kernels: csrc/*.hip behind the C ABI of include/y3d.h (liby3d_hip.so, bound by _lib.py).
"""
from yolov10_3d_amd._lib import LIB_PATH, Y3DError, lib  # noqa: F401
from yolov10_3d_amd.ops import compute_dtype, fp8_conv, set_compute_dtype, set_fp8_conv, set_weight_quant, weight_quant  # noqa: F401
from yolov10_3d_amd import modules, tasks, loss, kitti  # noqa: F401
from yolov10_3d_amd.tasks import (DetectionModel, YOLOv10DetectionModel, YOLOv10_3DDetectionModel, parse_model,  # noqa: F401
                                  yaml_model_load)

__version__ = "0.1.0"
