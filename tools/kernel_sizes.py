#!/usr/bin/env python3
"""Code bytes per kernel family of the built objects (llvm-readelf symbol sizes of every gfx950 code object).
usage: kernel_sizes.py [objects...] [--top N]"""
import argparse, collections, os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources as kr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("paths", nargs="*")
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    paths = a.paths or kr.default_objects()
    fam = collections.defaultdict(lambda: [0, 0])
    tmp = tempfile.mkdtemp()
    for p in paths:
        for co in kr.code_objects(p, tmp):
            out = subprocess.run([kr._tool("llvm-readelf"), "-s", "--demangle", "-W", co], capture_output=True, text=True).stdout
            out = out[out.find("'.symtab'"):]   # the same symbols are listed under .dynsym too
            for l in out.splitlines():
                f = l.split(None, 7)
                if len(f) >= 8 and f[3] == "FUNC" and f[4] in ("GLOBAL", "WEAK"):
                    name = re.sub(r"^void ", "", f[7])
                    name = re.sub(r"shiftnd::\(anonymous namespace\)::|shiftnd::", "", name)
                    key = os.path.basename(p).replace("shiftnd_", "").replace(".hip.o", "") + ":" + re.split(r"[<(]", name)[0]
                    fam[key][0] += int(f[2]); fam[key][1] += 1
    rows = sorted(fam.items(), key=lambda kv: -kv[1][0])
    tot = sum(v[0] for _, v in rows); n = sum(v[1] for _, v in rows)
    for k, (b, c) in rows[:a.top]:
        print("%-48s %4d kernels %9d bytes  %7d / kernel" % (k, c, b, b // c))
    print("total %d kernels, %.2f MB of code" % (n, tot / 1e6))


if __name__ == "__main__":
    main()
