#!/usr/bin/env python3
"""The store-data hazard of DESIGN section 9, checked on the BUILT code: a `buffer_store_dwordx3/x4` whose soffset is an
SGPR must not be followed, in the very next issue slot, by a VALU instruction that writes one of its data registers -- LLVM's hazard
recognizer inserts the wait state only when soffset is not a register.  shiftnd_common.hpp: buffer_store_b128_soffset puts `s_nop 1`
behind such stores; nothing but this check stops a future compiler (or a new kernel that calls the builtin directly) from bringing the
pair back.

    python3 tools/isa_store_hazard.py [objects or libraries ...]      (default: activesparseshifts-pytorch_amd/build/*.hip.o)
exit code 1 and one line per offending pair when any is found."""
import glob
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources as KR  # noqa: E402

STORE = re.compile(r"^buffer_store_dwordx[34]\s+v\[(\d+):(\d+)\],\s*(\S+),\s*s\[\d+:\d+\],\s*(\S+)")


def dest_vgprs(ins):
    """VGPRs a VALU instruction writes (first operand), as a set; empty for instructions with a scalar destination"""
    m = re.match(r"^v_\S+\s+(v\[(\d+):(\d+)\]|v(\d+))(?=[,\s]|$)", ins)
    if not m:
        return set()
    if m.group(4) is not None:
        return {int(m.group(4))}
    return set(range(int(m.group(2)), int(m.group(3)) + 1))


def scan(co):
    txt = subprocess.run([KR._tool("llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
    bad, stores, kernel, prev = [], 0, None, None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            kernel, prev = m.group(1), None
            continue
        ins = line.strip()
        if not ins or ins.startswith(("//", ";")):
            continue
        ins = ins.split("//")[0].strip()
        if prev is not None:
            lo, hi = prev
            if ins.startswith("v_") and dest_vgprs(ins) & set(range(lo, hi + 1)):
                bad.append((kernel, "v[%d:%d]" % (lo, hi), ins))
            prev = None
        m = STORE.match(ins)
        if m and re.match(r"^s\d+$", m.group(4)):   # soffset in an SGPR (not `0` / `off` / a literal)
            stores += 1
            prev = (int(m.group(1)), int(m.group(2)))
    return stores, bad


def check(paths):
    stores, bad = 0, []
    with tempfile.TemporaryDirectory() as tmp:
        for p in paths:
            for co in KR.code_objects(p, tmp):
                s, b = scan(co)
                stores += s
                bad += b
    return stores, bad


def main():
    paths = sys.argv[1:] or sorted(glob.glob(os.path.join(KR.ROOT, "activesparseshifts-pytorch_amd", "build", "*.hip.o")))
    stores, bad = check(paths)
    print("%d wide buffer stores with a register soffset in %d files; %d followed by a VALU write of their data" % (stores, len(paths), len(bad)))
    for k, regs, ins in bad:
        print("  %s: store of %s then `%s`" % (k, regs, ins))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
