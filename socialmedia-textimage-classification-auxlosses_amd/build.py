"""Build libmmhip.so (hand-written HIP kernels + engine + C ABI) for gfx950 with hipcc, in-tree.

`python -m smtc_amd.build` or `build()`; hipcc cross-compiles without a GPU.  The library is plain C ABI
(include/mmhip.h): no torch headers, no pybind -- the Python side binds it with ctypes.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmmhip.so")
SOURCES = ["gemm.hip", "gemm8.hip", "x3.hip", "attention.hip", "rowops.hip", "heads.hip", "engine.hip", "early.hip", "capi_ops.hip", "image.hip"]
HEADERS = ["mmhip_common.h", "mmhip_kernels.h", os.path.join("..", "..", "include", "mmhip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-fgpu-rdc=0"]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (needed to build libmmhip.so for gfx950)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    cc = hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    flags = [f for f in FLAGS if f != "-fgpu-rdc=0"]

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        path = os.path.join(CSRC, src)
        if force or _stale(obj, [path] + hdrs):
            cmd = [cc] + flags + ["-c", path, "-o", obj]
            if verbose:
                print("[mmhip build]", " ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or _stale(LIB, objs):
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
        if verbose:
            print("[mmhip build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
