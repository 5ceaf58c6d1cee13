"""Builds liby3d_hip.so (gfx950) in-tree:  python yolov10-3d_amd/csrc/build.py [--force]

One hipcc -c per source (in parallel, cached by mtime), then one link.  No torch involved: the
library is a plain C-ABI shared object (include/y3d.h).  Every compile also records the compiler's
per-kernel resource report (`-Rpass-analysis=kernel-resource-usage`) next to the object:
`resource_usage()` returns it, and tests/test_abi.py fails the build when a conv / attention kernel
needs scratch (a spill inside those pipelines drains the LDS-DMA prefetch: conv3x3_wide3.hip header)."""
import concurrent.futures as cf
import os
import re
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
    if force or needs(src, obj, deps) or not os.path.exists(obj + ".usage.txt"):
        cmd = ["hipcc", "-x", "hip", *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {name}:\n{r.stdout}\n{r.stderr}")
        remarks = [ln for ln in r.stderr.splitlines() if "-Rpass-analysis=kernel-resource-usage" in ln]
        with open(obj + ".usage.txt", "w") as f:
            f.write("\n".join(remarks) + "\n")
        # warnings (with their source context) pass through; the remarks and THEIR context lines do not
        keep, on = [], False
        for ln in r.stderr.splitlines():
            if re.search(r": (warning|error|note):", ln):
                on = True
            elif "remark:" in ln:
                on = False
            if on:
                keep.append(ln)
        if keep:
            sys.stderr.write("\n".join(keep) + "\n")
        return name, True
    return name, False


def resource_usage():
    """-> {source file: {kernel (mangled name): {"vgprs", "sgprs", "scratch", "lds", "occupancy"}}} from the last compile of every source"""
    out = {}
    for name in sources():
        path = os.path.join(OBJ, name + ".o.usage.txt")
        if not os.path.exists(path):
            continue
        kernels, cur = {}, None
        for ln in open(path):
            m = re.search(r"Function Name: (\S+)", ln)
            if m:
                cur = kernels.setdefault(m.group(1), {})
                continue
            if cur is None:
                continue
            for key, pat in (("vgprs", r" VGPRs: (\d+)"), ("sgprs", r"TotalSGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                             ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)")):
                m = re.search(pat, ln)
                if m:
                    cur[key] = int(m.group(1))
        out[name] = kernels
    return out


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
