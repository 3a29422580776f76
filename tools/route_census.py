#!/usr/bin/env python3
"""Which kernel serves which problem with every knob untouched: a seeded sample of geometries (1-3 dims, rows of 1..70000
elements, crops, contiguous and channels-last, four float dtypes and the quantized ones), forward and backward, keyed by
shiftnd_last_kernel().  Prints, per kernel name, the number of problems and a few example problems -- the evidence for
"this family has a default-routed shape" (tests/test_routing_gpu.py pins one example of each).
usage: route_census.py [--cases 4000] [--seed 0] [--out gpurun_out/route_census.txt]"""
import argparse, collections, os, sys
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "activesparseshifts-pytorch_amd"))
from torchshifts import abi  # noqa: E402

DEV = "cuda:0"
FLOATS = [torch.float32, torch.float64, torch.float16, torch.bfloat16]
QUANT = [torch.uint8, torch.int8, torch.int32]


def dims(rs, nd):
    """spatial sizes: a mix of tiny, typical feature-map and long-row shapes, <= 4M elements per (n, c) plane"""
    kind = rs.randint(0, 5)
    pick = lambda pool: int(pool[rs.randint(0, len(pool))])
    inner_pool = [1, 2, 3, 4, 5, 7, 8, 12, 14, 16, 24, 28, 31, 32, 48, 56, 62, 64, 96, 112, 128, 200, 222, 224, 225, 256, 384, 448, 512, 1000,
                  1024, 2048, 4096, 5000, 16384, 40000, 70000]
    outer_pool = [1, 2, 3, 4, 5, 7, 8, 14, 16, 28, 32, 56, 64, 100, 112, 224, 300, 512, 1024, 4096]
    while True:
        if kind == 0:
            s = [int(rs.randint(1, 20)) for _ in range(nd)]
        else:
            s = [pick(outer_pool) for _ in range(nd - 1)] + [pick(inner_pool)]
        if int(np.prod(s)) <= (1 << 22):
            return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=4000)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default="gpurun_out/route_census.txt")
    ap.add_argument("--via-op", action="store_true",
                    help="channels-last float problems through torch.ops.torchshifts._shift{N}d_forward/_backward (the op asks the "
                         "library whether a kernel reads the layout as it lies and changes the layout once otherwise) instead of "
                         "handing the strided tensors to the C ABI")
    ap.add_argument("--dump", default="", help="comma-separated kernel-name prefixes: write EVERY problem they served to <out>.dump")
    a = ap.parse_args()
    if a.via_op:
        import torchshifts  # noqa: F401  (registers the ops)
    rs = np.random.RandomState(a.seed)
    abi.set_path_policy(0)
    seen = collections.defaultdict(list)
    count = collections.Counter()

    dump_prefixes = [d for d in a.dump.split(",") if d]
    dumped = []

    def note(kind, desc):
        k = (kind, abi.last_kernel())
        count[k] += 1
        if len(seen[k]) < 6:
            seen[k].append(desc)
        if any(k[1].startswith(d) for d in dump_prefixes):
            dumped.append("%s %s %s" % (kind, k[1], desc))

    for it in range(a.cases):
        nd = int(rs.randint(1, 4))
        sp = dims(rs, nd)
        plane = int(np.prod(sp))
        budget = max(1, (1 << 23) // plane)
        N = int(min(budget, rs.choice([1, 2, 3, 8, 64])))
        C = int(min(max(1, budget // N), rs.choice([1, 2, 3, 16, 64, 256])))
        shape = [N, C] + sp
        crop = None
        if rs.rand() < 0.3 and min(sp) >= 5:
            crop = [[int(rs.randint(0, 3)), int(rs.randint(0, 3))] for _ in range(nd)]
        b, new = abi.check_borders(shape, crop, nd) if crop else (None, shape)
        pad, active = int(rs.randint(0, 5)), int(rs.randint(0, 2))
        cl = nd >= 2 and C > 1 and rs.rand() < 0.2
        quant = rs.rand() < 0.15
        desc = "%s crop=%s pad=%d active=%d%s" % (shape, crop, pad, active, " channels-last" if cl else "")
        if quant:
            tdt = QUANT[rs.randint(0, 3)]
            if tdt == torch.int32:
                xq = torch.randint(-1000, 1000, shape, dtype=tdt, device=DEV)
            else:
                info = torch.iinfo(tdt)
                xq = torch.randint(info.min, info.max + 1, shape, dtype=tdt, device=DEV)
            wq = torch.randint(118, 139, (C, nd), dtype=torch.uint8, device=DEV)
            outq = None
            if cl:
                xq = abi.to_channels_last(xq)
                outq = abi.to_channels_last(torch.empty(new, dtype=tdt, device=DEV))
            if cl and a.via_op:
                # the QuantizedCUDA op's rule (torch_binding.cpp: qshift_forward_hip): the LDS-tiled kernel when the library serves
                # the layout as it lies, otherwise one layout change, the contiguous kernel, one change back
                import ctypes
                pr = abi.problem(xq, pad, False, b)
                direct = abi.lib().shiftnd_forward_serves_channels_last(ctypes.byref(pr), xq.data_ptr(), abi.strides5(xq),
                                                                       outq.data_ptr(), abi.strides5(outq))
                if direct:
                    abi.forward_quantized(xq, wq, 128, 3, pad, b, out=outq)
                else:
                    abi.forward_quantized(xq.contiguous(), wq, 128, 3, pad, b)
            else:
                abi.forward_quantized(xq, wq, 128, 3, pad, b, out=outq)
            note("forward_quantized", "%s %s" % (str(tdt).replace("torch.", ""), desc))
            continue
        tdt = FLOATS[rs.randint(0, 4)]
        x = torch.rand(shape, device=DEV).to(tdt)
        go = torch.rand(new, device=DEV).to(tdt)
        w = ((torch.rand(C, nd, device=DEV) * 2 - 1) * float(rs.choice([1.5, 4.0, 40.0]))).to(tdt)
        out = gx = None
        if cl:   # input, output and both gradients channels-last (what a channels-last module hands over)
            x, go = abi.to_channels_last(x), abi.to_channels_last(go)
            out, gx = torch.empty_like(go), torch.empty_like(x)
        d = "%s %s" % (str(tdt).replace("torch.", ""), desc)
        if cl and a.via_op:
            bt = torch.tensor(b if b else abi.default_borders(x), dtype=torch.int32)
            getattr(torch.ops.torchshifts, "_shift%dd_forward" % nd)(x, w, bt, list(new), pad, bool(active))
            note("forward", d)
            getattr(torch.ops.torchshifts, "_shift%dd_backward" % nd)(go, w, x, bt, pad, bool(active))
            note("backward", d)
            continue
        abi.forward(x, w, pad, active, b, out=out)
        note("forward", d)
        abi.backward(go, w, x, pad, active, b, grad_x=gx)
        note("backward", d)
    torch.cuda.synchronize()
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        for (kind, k), n in sorted(count.items(), key=lambda kv: (kv[0][0], -kv[1])):
            f.write("%-18s %-34s %5d problems\n" % (kind, k, n))
            for dsc in seen[(kind, k)]:
                f.write("        %s\n" % dsc)
    if dump_prefixes:
        with open(a.out + ".dump", "w") as f:
            f.write("\n".join(dumped) + "\n")
    total = collections.Counter()
    for (kind, k), n in count.items():
        total[kind] += n
    tail = collections.Counter()
    for (kind, k), n in count.items():
        if k.startswith("strided_") or k.startswith("cl_backward") or k.startswith("cl_gather") or k.startswith("cl_active"):
            tail[kind] += n
    with open(a.out, "a") as f:
        for kind in sorted(total):
            f.write("fallback tail (strided_* / cl_backward / cl_gather_forward / cl_active_forward): %-18s %4d of %5d = %.2f %%\n"
                    % (kind, tail[kind], total[kind], 100.0 * tail[kind] / max(1, total[kind])))
    print(open(a.out).read())


if __name__ == "__main__":
    main()
