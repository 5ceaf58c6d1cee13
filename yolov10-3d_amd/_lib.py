"""ctypes binding of liby3d_hip.so (the C ABI declared in include/y3d.h).

The prototypes are read from the header itself, so the binding can never drift from the ABI and
`tests/test_abi.py` can check that every declared symbol is exported.  There is NO fallback: if the
library is missing or a call fails, a Python exception is raised (the reference's error behaviour
is Python exceptions too)."""
from __future__ import annotations

import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liby3d_hip.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "y3d.h")

F32, BF16 = 0, 1

_CT = {"int": ctypes.c_int, "int64_t": ctypes.c_int64, "float": ctypes.c_float, "double": ctypes.c_double}


class Y3DError(RuntimeError):
    pass


def parse_header(path: str = HEADER):
    """-> {name: (restype, [argtypes])} for every function declared in y3d.h"""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|int)\s+(y3d_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    toks = a.split()
                    t = toks[1] if toks[0] == "const" else toks[0]
                    argtypes.append(_CT[t])
        protos[name] = (ctypes.c_char_p if "char" in ret else ctypes.c_int, argtypes)
    return protos


# functions whose int return value is a result, not a status
_PLAIN_INT = {"abi_version", "conv_kpad", "conv_stat_blocks", "conv2d_stat_rows", "conv2d_wgrad_splits", "conv2d_wgrad_plan", "set_tile_kernels", "set_stream1x1", "set_bn_wide_slabs", "get_stream1x1", "get_tile_kernels", "dw_blocks", "dw_wgrad_blocks", "bn_bwd_blocks", "proj_blocks", "proj_group_blocks", "proj_group_bn_bwd_blocks", "proj_group_bwd_weight_bn_mfma_blocks", "tal3d_scratch_floats", "v10_postprocess_scratch_floats", "conv2d_fwd_affine_res_ok", "stem_conv_train_rows", "conv3x3_fp8_ok", "conv3x3_fp8_stat_rows", "fp8_scale_pitch"}


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise Y3DError(
                f"{LIB_PATH} not found: build it with `python yolov10-3d_amd/csrc/build.py` "
                "(or __graft_entry__.build()). There is no CPU / PyTorch fallback for the HIP path.")
        self._dll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        for name, (ret, argtypes) in self.protos.items():
            fn = getattr(self._dll, name)  # AttributeError if the symbol is not exported
            fn.restype = ret
            fn.argtypes = argtypes

    def last_error(self) -> str:
        return (self._dll.y3d_last_error() or b"").decode()

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        fn = getattr(self._dll, "y3d_" + name)
        ret = self.protos["y3d_" + name][0]
        if ret is not ctypes.c_int or name in _PLAIN_INT:
            setattr(self, name, fn)
            return fn

        def call(*a):
            rc = fn(*a)
            if rc != 0:
                raise Y3DError(f"y3d_{name} failed ({rc}): {self.last_error()}")
            return rc

        setattr(self, name, call)
        return call


_lib = None


def lib() -> _Lib:
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib
