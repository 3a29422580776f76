#!/usr/bin/env python3
"""Kernels of the built objects that a traced run never launched.
  launched.txt: "<calls>\\t<demangled kernel name>" per line (tools/suite_kernels.sh writes it from rocprofv3's kernel stats)
usage: unlaunched_kernels.py gpurun_out/suite_kernels/launched.txt [--list]"""
import argparse, collections, os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources as kr


def norm(name):
    name = name.strip()
    if name.endswith(".kd"):
        name = name[:-3]
    name = re.sub(r"^void ", "", name)
    name = name.replace("shiftnd::(anonymous namespace)::", "").replace("shiftnd::", "")
    # template head only: arguments in parentheses differ between demanglers
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            name = name[:i]
            break
    return re.sub(r"\s+", "", name)


def library_kernels(paths):
    tmp = tempfile.mkdtemp()
    out = {}
    for p in paths:
        for co in kr.code_objects(p, tmp):
            txt = subprocess.run([kr._tool("llvm-readelf"), "-s", "--demangle", "-W", co], capture_output=True, text=True).stdout
            txt = txt[txt.find("'.symtab'"):]
            for l in txt.splitlines():
                f = l.split(None, 7)
                if len(f) >= 8 and f[3] == "FUNC" and f[4] in ("GLOBAL", "WEAK"):
                    out[norm(f[7])] = (os.path.basename(p), int(f[2]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("launched")
    ap.add_argument("--list", action="store_true")
    a = ap.parse_args()
    launched = {}
    for l in open(a.launched):
        c, n = l.rstrip("\n").split("\t", 1)
        launched[norm(n)] = int(c)
    lib = library_kernels(kr.default_objects())
    fam = collections.defaultdict(lambda: [0, 0, 0])
    missing = []
    for k, (obj, size) in lib.items():
        key = obj.replace("shiftnd_", "").replace(".hip.o", "") + ":" + re.split(r"<", k)[0]
        fam[key][0] += 1
        if k in launched:
            fam[key][1] += 1
        else:
            fam[key][2] += size
            missing.append((key, k))
    unknown = [k for k in launched if k not in lib]
    print("%d kernels in the library, %d launched, %d launched names not in the library" % (len(lib), sum(v[1] for v in fam.values()), len(unknown)))
    for key, (n, hit, size) in sorted(fam.items(), key=lambda kv: -(kv[1][0] - kv[1][1])):
        print("%-44s %4d kernels %4d launched %4d never  (%8d bytes never run)" % (key, n, hit, n - hit, size))
    if a.list:
        for key, k in sorted(missing):
            print("never:", k)
        for k in unknown[:40]:
            print("unknown:", k)


if __name__ == "__main__":
    main()
