#!/usr/bin/env python3
"""Per-kernel resources of the built gfx950 code objects: VGPRs, AGPRs, SGPRs, LDS, scratch, spills.

    python tools/kernel_resources.py [--filter walk_backward] [--scratch-only] [objects or libraries ...]

Reads the AMDGPU metadata note (llvm-readelf --notes) of every gfx950 code object bundled in the given .o / .so files
(default: activesparseshifts-pytorch_amd/build/*.hip.o).  `check_no_scratch()` is what build.py calls after compiling:
a kernel of this library that needs a private segment (scratch) spills in a bandwidth-bound loop -- the build fails.
Occupancy on gfx950: 512 VGPRs per SIMD lane, allocated in blocks of 8: waves per SIMD = floor(512 / ceil8(vgprs)), max 8.
"""
import argparse
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def _tool(name):
    p = os.path.join(LLVM, name)
    return p if os.path.exists(p) else name


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return [n.replace("shiftnd::(anonymous namespace)::", "").replace("void ", "") for n in out]


def code_objects(path, tmp):
    """gfx950 code objects inside an object file / shared library (its .hip_fatbin section holds one bundle per TU)"""
    fat = os.path.join(tmp, os.path.basename(path) + ".fatbin")
    subprocess.check_call([_tool("llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", path, fat])
    data = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), data)]
    outs = []
    for i, s in enumerate(starts):
        part = os.path.join(tmp, "%s.%d.bundle" % (os.path.basename(path), i))
        open(part, "wb").write(data[s:starts[i + 1] if i + 1 < len(starts) else len(data)])
        co = part + ".co"
        r = subprocess.run([_tool("clang-offload-bundler"), "--unbundle", "--type=o", "--targets=" + TARGET, "--input=" + part,
                            "--output=" + co], capture_output=True, text=True)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
            outs.append(co)
        elif TARGET.encode() in data[s:s + 4096]:   # the bundle holds a gfx950 entry and the bundler could not extract it
            raise RuntimeError("kernel_resources: clang-offload-bundler failed on %s: %s" % (os.path.basename(path), (r.stderr or "").strip()[:200]))
    return outs


FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size",
          "vgpr_spill_count", "sgpr_spill_count")


def kernels_of(co):
    txt = subprocess.run([_tool("llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    ks, cur = [], None
    for line in txt.splitlines():
        m = re.match(r"\s+-? *\.(\w+):\s+(\S+)\s*$", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2)
        if line.lstrip().startswith("- .agpr_count"):
            cur = {}
            ks.append(cur)
        if cur is None:
            continue
        if key == "name":
            cur["name"] = val
        elif key in FIELDS:
            cur[key] = int(val)
    return [k for k in ks if "name" in k]


def collect(paths):
    res = []
    with tempfile.TemporaryDirectory() as tmp:
        for p in paths:
            for co in code_objects(p, tmp):
                for k in kernels_of(co):
                    k["file"] = os.path.basename(p)
                    res.append(k)
    for k, d in zip(res, demangle([k["name"] for k in res])):
        k["demangled"] = d
    return res


def waves_per_simd(vgprs, agprs=0):
    tot = ((max(vgprs, 1) + 7) // 8) * 8 + ((agprs + 7) // 8) * 8
    return min(8, 512 // tot)


def default_objects():
    return sorted(glob.glob(os.path.join(ROOT, "activesparseshifts-pytorch_amd", "build", "*.hip.o")))


def check_no_scratch(paths=None):
    """-> list of (kernel, bytes per lane) that use scratch; build.py raises when it is not empty.
    Raises itself when the check could pass vacuously: an object that yields no kernel (the bundler failed, the note format
    changed) or a kernel record without the fields the check reads."""
    paths = list(paths or default_objects())
    ks = collect(paths)
    # (a host-only TU -- shiftnd_api.hip -- legitimately holds no kernel; a LIBRARY's worth of objects without kernels does not)
    if len(ks) < max(1, len(paths) // 2):
        raise RuntimeError("kernel_resources: %d kernels found in %d objects (clang-offload-bundler / llvm-readelf --notes output changed?)" % (len(ks), len(paths)))
    blind = [k["demangled"] for k in ks if "private_segment_fixed_size" not in k or "vgpr_count" not in k]
    if blind:
        raise RuntimeError("kernel_resources: %d kernel records without private_segment_fixed_size / vgpr_count (note format?): %s" % (len(blind), blind[0]))
    return [(k["demangled"], k["private_segment_fixed_size"]) for k in ks
            if k.get("private_segment_fixed_size", 0) > 0 or k.get("vgpr_spill_count", 0) > 0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("paths", nargs="*")
    ap.add_argument("--filter", default="")
    ap.add_argument("--scratch-only", action="store_true")
    ap.add_argument("--summary", action="store_true", help="counts per file only")
    a = ap.parse_args()
    ks = collect(a.paths or default_objects())
    if a.summary:
        per = {}
        for k in ks:
            per.setdefault(k["file"], []).append(k)
        for f, v in sorted(per.items()):
            print("%-28s %5d kernels, max vgpr %3d, with scratch %d" % (f, len(v), max(k["vgpr_count"] for k in v),
                                                                      sum(1 for k in v if k.get("private_segment_fixed_size", 0) > 0)))
        print("total %d kernels" % len(ks))
        return 0
    bad = 0
    for k in sorted(ks, key=lambda k: k["demangled"]):
        if a.filter and a.filter not in k["demangled"]:
            continue
        scr = k.get("private_segment_fixed_size", 0)
        if a.scratch_only and scr == 0:
            continue
        bad += scr > 0
        print("%-90s vgpr %3d agpr %3d sgpr %3d lds %6d scratch %4d  waves/SIMD %d" % (
            k["demangled"][:90], k["vgpr_count"], k.get("agpr_count", 0), k["sgpr_count"], k.get("group_segment_fixed_size", 0), scr,
            waves_per_simd(k["vgpr_count"], k.get("agpr_count", 0))))
    return 1 if (a.scratch_only and bad) else 0


if __name__ == "__main__":
    sys.exit(main())
