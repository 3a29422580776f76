#!/usr/bin/env python3
"""For every s_barrier of a kernel in a `hipcc -S` listing: which LDS instructions (ds_*) were issued since the last
`s_waitcnt ... lgkmcnt(0)` -- i.e. may still be in flight when the wave arrives at the barrier -- and which memory instructions
follow before the next wait.  A barrier with LDS writes or reads outstanding does not order them against the other waves' accesses.
    python3 tools/isa_barriers.py listing.s <mangled-name-substring> [...]"""
import re
import sys


def kernels(path):
    name, body = None, []
    for line in open(path, errors="replace"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            if name:
                yield name, body
            name, body = m.group(1), []
        elif name is not None:
            if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
                yield name, body
                name, body = None, []
            else:
                body.append(line.rstrip())
    if name:
        yield name, body


def lgkm_zero(ins):
    m = re.search(r"lgkmcnt\((\d+)\)", ins)
    return ("s_waitcnt" in ins and m and int(m.group(1)) == 0) or re.match(r"\s*s_waitcnt\s+0x", ins or "") is not None and "lgkmcnt" not in ins and False


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    for name, body in kernels(path):
        if pats and not any(p in name for p in pats):
            continue
        ins = [l.strip() for l in body if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        bars = [i for i, l in enumerate(ins) if l.startswith("s_barrier")]
        print("== %s: %d instructions, %d s_barrier" % (name, len(ins), len(bars)))
        for b in bars:
            pending = []
            j = b - 1
            while j >= 0:
                l = ins[j]
                if l.startswith("s_waitcnt") and lgkm_zero(l):
                    break
                if l.startswith("s_barrier") or l.startswith("s_cbranch") or l.startswith("s_branch") or l.startswith("s_endpgm"):
                    pending.append("<" + l.split()[0] + ">")
                    break
                if l.startswith(("ds_", "flat_")):
                    pending.append(l.split()[0])
                j -= 1
            wait = ins[b - 1] if b else ""
            print("  barrier @%d  prev: %-40s outstanding LDS since lgkmcnt(0): %s" % (b, wait[:40], ", ".join(reversed(pending)) or "none"))


if __name__ == "__main__":
    main()
