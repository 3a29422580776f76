#!/usr/bin/env python3
"""Summarise a profiles/collect.sh output directory: per-kernel average duration (kernel trace) and
per-launch PMC counters (FETCH_SIZE doubled for gfx950 as MI355X_MICROARCH.md section HBM prescribes)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("shiftnd::(anonymous namespace)::", "").replace("void ", "")
    return name[:90]


def main(root):
    traffic = defaultdict(dict)  # kernel base name -> {"read_bytes", "written_bytes"} per launch (for bench.py)
    by_inst = defaultdict(dict)  # ... and per template instantiation (round-5 verdict: a base-name average can hide one bad instantiation)
    for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
        dur = defaultdict(list)
        res = {}
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            res[k] = "vgpr %s agpr %s sgpr %s lds %s wg %s grid %s" % (
                r.get("VGPR_Count", "?"), r.get("Accum_VGPR_Count", "?"), r.get("SGPR_Count", "?"),
                r.get("LDS_Block_Size", "?"), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")),
                r.get("Grid_Size_X", r.get("Grid_Size", "?")))
        print("== kernel trace:", os.path.relpath(f, root))
        print("%-92s %6s %12s %12s" % ("kernel", "calls", "avg_us", "min_us"))
        for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
            if any(t in k for t in ("sweep_", "plane_", "strided_", "reduce_weight", "cl_", "transpose", "slide_", "bytes_", "small_", "band_", "rows_", "step_", "walk_", "qpool_", "span_", "crop_", "row_", "flat_")):
                print("%-92s %6d %12.1f %12.1f   %s" % (k, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, res[k]))
    for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(list))
            for r in csv.DictReader(open(f)):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
            print("== pmc:", os.path.relpath(f, root))
            for k, ctrs in acc.items():
                if not any(t in k for t in ("sweep_", "plane_", "strided_", "cl_", "slide_", "bytes_", "small_", "band_", "rows_", "step_", "walk_", "qpool_", "span_", "crop_", "row_", "flat_")):
                    continue
                for c, v in ctrs.items():
                    avg = sum(v) / len(v)
                    note = ""
                    base = k.split("<")[0].split("(")[0].strip()
                    if base == "slide_kernel":  # one template for both directions: name them like shiftnd_last_kernel does
                        targs = k.split("<", 1)[1].split(",")
                        base = "slide_backward" if len(targs) > 3 and targs[3].strip() == "true" else "slide_forward"
                    if c == "FETCH_SIZE":
                        note = "  KB/launch; x2 (gfx950 correction) = %.3f GB read" % (avg * 2 * 1024 / 1e9)
                        traffic[base]["read_bytes"] = avg * 2 * 1024
                        by_inst[k]["read_bytes"] = avg * 2 * 1024
                    if c == "WRITE_SIZE":
                        note = "  KB/launch = %.3f GB written" % (avg * 1024 / 1e9)
                        traffic[base]["written_bytes"] = avg * 1024
                        by_inst[k]["written_bytes"] = avg * 1024
                    print("%-70s %-28s avg %16.1f over %d launches%s" % (k[:70], c, avg, len(v), note))


    if traffic:
        import json
        with open(os.path.join(root, "traffic.json"), "w") as fh:
            json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py (profiles/collect.sh); "
                                 "FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md",
                       "per_launch": traffic, "per_launch_by_instantiation": by_inst}, fh, indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
