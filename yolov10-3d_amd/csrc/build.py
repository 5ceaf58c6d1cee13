"""Builds liby3d_hip.so (gfx950) in-tree:  python yolov10-3d_amd/csrc/build.py [--force]

One hipcc -c per source (in parallel, cached by mtime), then one link.  No torch involved: the
library is a plain C-ABI shared object (include/y3d.h)."""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "liby3d_hip.so")
OBJ = os.path.join(HERE, "_obj")
ARCH = "gfx950"
FLAGS = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function", "-ffp-contract=off"]


def sources():
    return sorted(f for f in os.listdir(HERE) if f.endswith((".hip", ".cpp")))


def needs(src, obj, deps):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in [src] + deps)


def compile_one(name, force):
    src = os.path.join(HERE, name)
    obj = os.path.join(OBJ, name + ".o")
    deps = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")] + [os.path.join(HERE, "..", "..", "include", "y3d.h")]
    if force or needs(src, obj, deps):
        cmd = ["hipcc", "-x", "hip", *FLAGS, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {name}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
        return name, True
    return name, False


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sources()
    with cf.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        res = list(ex.map(lambda s: compile_one(s, force), srcs))
    rebuilt = [n for n, r in res if r]
    objs = [os.path.join(OBJ, s + ".o") for s in srcs]
    if rebuilt or not os.path.exists(OUT):
        cmd = ["hipcc", "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", OUT]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"liby3d_hip.so: rebuilt {rebuilt or 'nothing'} -> {OUT}")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
