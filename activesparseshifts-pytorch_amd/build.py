#!/usr/bin/env python3
"""In-tree build of the MI355X shiftnd libraries (no setup.py, no JIT cache, no hipify).

  torchshifts/libshiftnd_hip.so   HIP kernels + C ABI (include/shiftnd_hip.h); hipcc, gfx950 only
  torchshifts/_C.so               torch dispatcher library (schemas, composite ops, autograd,
                                  CUDA/QuantizedCUDA adapters over the C ABI, CPU/QuantizedCPU host
                                  backend); g++ against the installed PyTorch-ROCm headers

Both are built next to the Python package so that `gpurun` ships them to the GPU box.
hipcc cross-compiles gfx950 without a GPU.  Usage:  python build.py [--force] [--jobs N]
"""
import argparse
import concurrent.futures as cf
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
PKG = os.path.join(HERE, "torchshifts")
OBJ = os.path.join(HERE, "build")
INC = os.path.join(ROOT, "include")

HIP_SOURCES = ["shiftnd_api.hip", "shiftnd_strided.hip", "shiftnd_plane.hip", "shiftnd_sweep.hip", "shiftnd_slide.hip", "shiftnd_step.hip", "shiftnd_step_fwd.hip", "shiftnd_walk3.hip", "shiftnd_walk.hip", "shiftnd_span.hip", "shiftnd_flat.hip", "shiftnd_bytes.hip", "shiftnd_small.hip", "shiftnd_rows.hip", "shiftnd_qpool.hip", "shiftnd_cl.hip", "shiftnd_cl_tiled.hip", "shiftnd_cl_tiled3.hip", "shiftnd_transpose.hip"]
CPP_SOURCES = ["torch_binding.cpp", "torch_cpu_backend.cpp"]
HEADERS = [os.path.join(CSRC, h) for h in ("shiftnd_common.hpp", "shiftnd_launch.hpp", "shiftnd_stage.hpp", "shiftnd_step.hpp")] + [
    os.path.join(INC, "shiftnd_hip.h")]

# -ffp-contract=off: interpolation must be mul+mul+add like the reference's x86-64 CPU build
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
               "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-I" + INC, "-I" + CSRC]


def newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def run(cmd):
    print("  " + " ".join(os.path.basename(c) if c.startswith("/") else c for c in cmd[:3]) + " ... " +
          os.path.basename(cmd[-1]), flush=True)
    subprocess.check_call(cmd)


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def build_hip(force, pool):
    lib = os.path.join(PKG, "libshiftnd_hip.so")
    objs, jobs = [], []
    for src in HIP_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src + ".o")
        objs.append(o)
        if force or newer(o, [s] + HEADERS):
            jobs.append(pool.submit(run, [hipcc()] + HIPCC_FLAGS + ["-c", s, "-o", o]))
    for j in jobs:
        j.result()
    if force or jobs or newer(lib, objs):
        check_no_scratch(objs)
        run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib])
    return lib


def check_no_scratch(objs):
    """Every kernel of this library is bandwidth-bound: one that needs a private segment (scratch) spills registers or indexes
    a local array dynamically inside its streaming loop.  tools/kernel_resources.py reads the AMDGPU metadata note of every
    code object (llvm-readelf --notes); the build fails when a kernel has private_segment_fixed_size > 0 or spilled VGPRs."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    bad = kr.check_no_scratch(objs)
    if bad:
        raise RuntimeError("kernels with scratch (bytes per lane): " + "; ".join("%s: %d" % b for b in bad[:20]))
    print("  no kernel of %d objects uses scratch" % len(objs), flush=True)


def build_torch(force, pool):
    import torch
    from torch.utils import cpp_extension as ce
    lib = os.path.join(PKG, "_C.so")
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    flags = ["-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "-fopenmp", "-DAT_PARALLEL_OPENMP=1",
             "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_C",
             "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
             "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas", "-Wno-sign-compare",
             "-I" + INC, "-I" + CSRC, "-I" + sysconfig.get_paths()["include"]]
    flags += ["-isystem" + p for p in ce.include_paths("cuda") if os.path.isdir(p)]
    objs, jobs = [], []
    for src in CPP_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src + ".o")
        objs.append(o)
        if force or newer(o, [s] + HEADERS):
            jobs.append(pool.submit(run, ["g++"] + flags + ["-c", s, "-o", o]))
    for j in jobs:
        j.result()
    if force or jobs or newer(lib, objs + [os.path.join(PKG, "libshiftnd_hip.so")]):
        run(["g++", "-shared", "-fopenmp"] + objs +
            ["-L" + PKG, "-lshiftnd_hip", "-L" + tlib, "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip",
             "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + tlib, "-o", lib])
    return lib


def build_all(force=False, jobs=None):
    os.makedirs(OBJ, exist_ok=True)
    with cf.ThreadPoolExecutor(max_workers=jobs or min(6, os.cpu_count() or 2)) as pool:
        hip = build_hip(force, pool)
        tor = build_torch(force, pool)
    return hip, tor


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    a = ap.parse_args()
    for p in build_all(a.force, a.jobs):
        print("built", p)
    sys.exit(0)
