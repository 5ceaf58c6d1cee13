"""TEST INFRASTRUCTURE — build-container only.  Never imported by the product path.

Imports the *reference* (baldhat/yolov10-3D, mounted read-only at /root/reference) as a
CPU oracle so that `oracle/make_golden.py` can mint golden vectors and so that
`tests/` can (optionally, when /root/reference exists) cross-check the CPU
restatement in `oracle/restate.py` against the real thing.

Nothing in here travels to the GPU box in a useful form: /root/reference does not
exist there and `available()` returns False.

What the shim does (all verified there):
  * registers inert stub modules for the 7 third-party packages the reference imports
    at module-import time but never reaches on the hot path
    (cv2, torchvision[.ops,.transforms,.transforms.functional], idlelib.pyparse,
     numba[.cuda], seaborn, thop, notion_client);
  * injects kernel_size_1/2 = 3 into 3D yamls that lack them (reference bug:
    nn/tasks.py:940-941 passes None, head.py:579 then raises TypeError);
  * pins distillation/fgdm/htl off (they need network / author-local files);
  * neutralises the hard-coded `.cuda()` in utils/loss.py:1132 for the duration of a
    3D-loss call.
"""
from __future__ import annotations

import contextlib
import importlib
import os
import sys
import types
from types import SimpleNamespace

REF_ROOT = os.environ.get("Y3D_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "ultralytics"))


class _Anything:
    """Inert attribute sink: any attribute / call / subscription yields another sink."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        # decorator use (numba.jit etc.): pass functions through unchanged
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return _Anything()

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Anything()

    def __getitem__(self, k):
        return _Anything()

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (object,)


def _stub(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []  # behave like a package so `import a.b` works once a.b is registered

    def _getattr(attr, _name=name):
        if attr.startswith("__") and attr.endswith("__"):
            raise AttributeError(attr)
        return _Anything()

    m.__getattr__ = _getattr  # PEP 562
    sys.modules[name] = m
    return m


_IMPORTED = None


def import_reference():
    """Return the imported `ultralytics` package of the reference (cached)."""
    global _IMPORTED
    if _IMPORTED is not None:
        return _IMPORTED
    if not available():
        raise RuntimeError(f"reference not present at {REF_ROOT}")
    for name in ("cv2", "seaborn", "thop", "notion_client"):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                _stub(name)
    if "cv2" in sys.modules and not hasattr(sys.modules["cv2"], "__version__"):
        sys.modules["cv2"].__version__ = "0.0-stub"
    try:
        importlib.import_module("torchvision")
    except Exception:
        tv = _stub("torchvision", __version__="0.15.2")
        tv.ops = _stub("torchvision.ops")
        tv.transforms = _stub("torchvision.transforms")
        tv.transforms.functional = _stub("torchvision.transforms.functional")
    try:
        importlib.import_module("numba")
    except Exception:
        nb = _stub("numba")
        nb.cuda = _stub("numba.cuda")
    try:
        importlib.import_module("idlelib.pyparse")
    except Exception:
        if "idlelib" not in sys.modules:
            _stub("idlelib")
        _stub("idlelib.pyparse", trans=None)
    os.environ.setdefault("YOLO_OFFLINE", "1")
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import ultralytics  # noqa: F401

    _IMPORTED = ultralytics
    return ultralytics


def model_args(**overrides) -> SimpleNamespace:
    """hyper-parameter namespace the reference losses read (cfg/default.yaml:102-139)."""
    import_reference()
    from ultralytics.utils import DEFAULT_CFG_DICT

    d = dict(DEFAULT_CFG_DICT)
    d.update(distillation=False, fgdm_loss=False, fgdm_supervision=False, htl=False)
    d.update(overrides)
    return SimpleNamespace(**d)


def load_yaml(rel: str) -> dict:
    """Load a model yaml of the reference (rel like 'v10-3D/yolov10s_3D.yaml'), with the
    kernel_size default fix."""
    import_reference()
    from ultralytics.nn.tasks import yaml_model_load

    d = yaml_model_load(os.path.join(REF_ROOT, "ultralytics", "cfg", "models", rel))
    if "3D" in rel:
        d.setdefault("kernel_size_1", 3)
        d.setdefault("kernel_size_2", 3)
        if d.get("kernel_size_1") is None:
            d["kernel_size_1"] = 3
        if d.get("kernel_size_2") is None:
            d["kernel_size_2"] = 3
    return d


def build_model(rel_or_dict, seed: int = 0, **arg_overrides):
    """Build a reference model with seeded random init (no network, no checkpoints)."""
    import torch

    import_reference()
    from ultralytics.nn.tasks import YOLOv10DetectionModel, YOLOv10_3DDetectionModel

    d = load_yaml(rel_or_dict) if isinstance(rel_or_dict, str) else rel_or_dict
    torch.manual_seed(seed)
    is3d = any(row[2] == "v10Detect3d" for row in d["head"])
    cls = YOLOv10_3DDetectionModel if is3d else YOLOv10DetectionModel
    m = cls(d, verbose=False)
    m.args = model_args(**arg_overrides)
    return m


@contextlib.contextmanager
def cpu_cuda_noop():
    """Make Tensor.cuda() a no-op (reference utils/loss.py:1132 hard-codes .cuda())."""
    import torch

    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        yield
    finally:
        torch.Tensor.cuda = orig
