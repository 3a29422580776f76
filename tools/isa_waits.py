#!/usr/bin/env python3
"""Per kernel of a hipcc -S listing: memory instructions and `s_waitcnt vmcnt(0)` inside loops (a vmcnt(0) in a
loop that also issues loads usually means a load or store sits under a thread-dependent branch and the prefetch
is not in flight: see DESIGN section 9).   tools/isa_waits.py file.s [name filter]"""
import re
import subprocess
import sys

text = open(sys.argv[1]).read().split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
name, rows, cur = None, [], None
for ln in text:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        name = m.group(1)
        cur = {"name": name, "loop_vm0": 0, "vm0": 0, "loads": 0, "stores": 0, "loop_loads": 0, "lines": 0}
        rows.append(cur)
        continue
    if cur is None:
        continue
    if ln.startswith(".Lfunc_end"):
        cur = None
        continue
    cur["lines"] += 1
    inloop = "in Loop" in ln or "Inner Loop" in ln
    if ln.startswith(".LBB"):
        cur["inloop"] = "Loop" in ln
        continue
    il = cur.get("inloop", False)
    if re.search(r"\b(global|buffer|flat)_load", ln):
        cur["loads"] += 1
        cur["loop_loads"] += il
    if re.search(r"\b(global|buffer|flat)_store", ln):
        cur["stores"] += 1
    if "s_waitcnt" in ln and re.search(r"vmcnt\(0\)", ln):
        cur["vm0"] += 1
        cur["loop_vm0"] += il
names = [r["name"] for r in rows]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for r, d in zip(rows, dem):
    d = re.sub(r"shiftnd::\(anonymous namespace\)::", "", d)
    d = re.sub(r"\(.*", "", d)
    if flt in d:
        print("%-78s lines %6d loads %4d (loop %4d) stores %4d vmcnt(0) %3d (loop %3d)" % (d[:78], r["lines"], r["loads"], r["loop_loads"], r["stores"], r["vm0"], r["loop_vm0"]))
