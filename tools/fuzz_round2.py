#!/usr/bin/env python3
"""Randomised parity run of the round-2 / round-3 / round-4 kernel families against the oracle (GPU box; a seeded, time-boxed slice of
it runs in `pytest -m gpu`: tests/test_fuzz_families_gpu.py):
    python3 tools/fuzz_round2.py [--seconds 120] [--seed 0]
channels-last tiled kernels (forward 1/2/4 bytes, active forward, backward), small-plane / row-band kernels, the byte
kernel with rounds.  Prints the number of cases per kernel and stops at the first mismatch."""
import argparse
import collections
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "activesparseshifts-pytorch_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from torchshifts import abi  # noqa: E402
from oracle import oracle as O  # noqa: E402

DEV = "cuda:0"
count = collections.Counter()


def rel_err(a, b):
    """max error relative to the largest weight gradient, or to 1 when all of them are small sums of large terms (a
    single channel whose terms cancel: the fp32 products then limit the relative error, not the kernel)"""
    d = np.abs(a.astype(np.float64) - b.astype(np.float64)).max()
    return d / max(np.abs(b).max(), 1.0)


def weights(rs, C, nd, sizes, span):
    w = rs.uniform(-span, span, size=(C, nd))
    k = rs.randint(0, 4)
    if k == 0:
        w[rs.randint(C)] = [s + 1.25 for s in sizes]
    if k == 1:
        w[rs.randint(C)] = [-(2 * s + 0.5) for s in sizes]
    if k == 2:
        w[rs.randint(C)] = [-3.0, -2.75, 3.0][:nd]
    return w


def case_cl(rs):
    C = int(rs.choice([4, 8, 12, 32, 36, 64, 100]))
    H = int(rs.choice([1, 5, 6, 9, 17, 33])); W = int(rs.randint(1, 41))
    N = int(rs.randint(1, 4))
    shape = (N, C, H, W)
    pad = int(rs.randint(0, 5)); active = int(rs.randint(0, 2))   # (periodic: the wrapped edges through the element pass)
    x = rs.uniform(-1, 1, size=shape).astype(np.float32); go = rs.uniform(-1, 1, size=shape).astype(np.float32)
    w = weights(rs, C, 2, shape[2:], 3.9).astype(np.float32)
    cl = torch.channels_last
    xd = torch.from_numpy(x).to(DEV).contiguous(memory_format=cl); gd = torch.from_numpy(go).to(DEV).contiguous(memory_format=cl)
    wd = torch.from_numpy(w).to(DEV)
    abi.set_tuning(21, int(rs.choice([0, 0, 3, 7])))
    ref = O.forward(x, w, pad, active)
    for out in (None, torch.empty(shape, device=DEV).contiguous(memory_format=cl)):
        o = abi.forward(xd, wd, pad, active, out=out)
        count[abi.last_kernel()] += 1
        assert np.array_equal(o.cpu().numpy(), ref), ("cl fwd", shape, pad, active)
    if rs.randint(2):   # the mixed form: NCHW incoming gradient, channels-last saved input and grad_x
        gd = torch.from_numpy(go).to(DEV)
    gx, gw = abi.backward(gd, wd, xd, pad, active, grad_x=torch.empty(shape, device=DEV).contiguous(memory_format=cl))
    count[abi.last_kernel()] += 1
    gx_o, _ = O.backward(go, w, x, pad, active)
    _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active)
    assert np.array_equal(gx.cpu().numpy(), gx_o), ("cl gx", shape, pad, active)
    if not rel_err(gw.cpu().numpy(), gw64) < 1e-5:   # post-mortem
        err = np.abs(gw.cpu().numpy() - gw64)
        print("CL GW MISMATCH", shape, pad, active, abi.last_kernel(), "weights", w.tolist(), "got", gw.cpu().numpy().tolist(), "want", gw64.tolist(), "err", err.tolist())
    assert rel_err(gw.cpu().numpy(), gw64) < 1e-5, ("cl gw", shape, pad, active)
    abi.set_tuning(21, 0)
    # quantized uint8, format kept
    if (C * 1) % 16 == 0:
        xq = rs.randint(0, 256, size=shape).astype(np.uint8); wq = rs.randint(122, 135, size=(C, 2)).astype(np.uint8)
        oq = torch.empty(shape, dtype=torch.uint8, device=DEV).contiguous(memory_format=cl)
        abi.forward_quantized(torch.from_numpy(xq).to(DEV).contiguous(memory_format=cl), torch.from_numpy(wq).to(DEV), 128, 3, pad, out=oq)
        count[abi.last_kernel() + "/u8"] += 1
        assert np.array_equal(oq.cpu().numpy(), O.forward_q(xq, wq, 128, 3, pad)), ("cl u8", shape, pad)


def case_ragged(rs):
    nd = int(rs.choice([1, 2, 2, 2, 3]))
    npdt = rs.choice([np.float32, np.float64])
    if nd == 1:
        sp = (int(rs.choice([3, 13, 301, 2301])),)
    elif nd == 2:
        sp = (int(rs.randint(1, 80)), int(rs.choice([1, 3, 5, 7, 13, 14, 27, 57, 113, 131])))
    else:
        sp = (int(rs.randint(1, 5)), int(rs.randint(1, 9)), int(rs.choice([3, 5, 7, 9])))
    N, C = int(rs.randint(1, 6)), int(rs.randint(1, 7))
    shape = (N, C) + sp
    if (sp[-1] * np.dtype(npdt).itemsize) % 16 == 0:
        return
    pad = int(rs.randint(0, 5)); active = int(rs.randint(0, 2))
    x = rs.uniform(-1, 1, size=shape).astype(npdt); go = rs.uniform(-1, 1, size=shape).astype(npdt)
    w = weights(rs, C, nd, sp, 4.5).astype(npdt)
    xd, gd, wd = torch.from_numpy(x).to(DEV), torch.from_numpy(go).to(DEV), torch.from_numpy(w).to(DEV)
    abi.set_tuning(25, int(rs.choice([0, 0, 1, 2, 5]))); abi.set_tuning(26, int(rs.choice([0, 0, 1, 3])))
    o = abi.forward(xd, wd, pad, active)
    count[abi.last_kernel()] += 1
    assert np.array_equal(o.cpu().numpy(), O.forward(x, w, pad, active)), ("ragged fwd", shape, pad, active, abi.last_kernel())
    gx, gw = abi.backward(gd, wd, xd, pad, active)
    count[abi.last_kernel()] += 1
    gx_o, _ = O.backward(go, w, x, pad, active)
    _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active)
    assert np.array_equal(gx.cpu().numpy(), gx_o), ("ragged gx", shape, pad, active, abi.last_kernel())
    e = rel_err(gw.cpu().numpy(), gw64)
    assert e < (1e-12 if npdt == np.float64 else 1e-5), ("ragged gw", shape, pad, active, abi.last_kernel(), str(npdt), e, w.tolist(), gw.cpu().numpy().tolist(), gw64.tolist())
    abi.set_tuning(25, 0); abi.set_tuning(26, 0)


def case_bytes(rs):
    H, W = [(8, 16), (28, 28), (56, 56), (4, 4), (112, 112), (12, 20), (150, 224), (530, 32), (75, 240)][rs.randint(9)]
    N, C = int(rs.randint(1, 41 if H * W <= 16384 else 5)), int(rs.randint(1, 5))
    shape = (N, C, H, W)
    pad = int(rs.randint(0, 5))
    xq = rs.randint(0, 256, size=shape).astype(np.uint8); wq = rs.randint(120, 137, size=(C, 2)).astype(np.uint8)
    abi.set_tuning(17, int(rs.choice([0, 0, 1, 3]))); abi.set_tuning(19, int(rs.choice([0, 0, 1, 2, 5]))); abi.set_tuning(29, int(rs.choice([0, 0, 5, 40])))
    o = abi.forward_quantized(torch.from_numpy(xq).to(DEV), torch.from_numpy(wq).to(DEV), 128, 11, pad)
    count[abi.last_kernel() + "/u8"] += 1
    assert np.array_equal(o.cpu().numpy(), O.forward_q(xq, wq, 128, 11, pad)), ("bytes", shape, pad)
    abi.set_tuning(17, 0); abi.set_tuning(19, 0); abi.set_tuning(29, 0)
    # the fused quantized shift + average pool, both of ATen's roundings, against a numpy restatement on the oracle's shift
    pool = (int(rs.randint(1, 4)), int(rs.randint(2, 4)))
    y = O.forward_q(xq, wq, 128, 11, pad).astype(np.int64) - 11
    psz = [-(-H // pool[0]), -(-W // pool[1])]
    for requant in (abi.REQUANT_ZP_INSIDE, abi.REQUANT_ZP_OUTSIDE):
        ref = np.zeros((N, C) + tuple(psz), dtype=np.uint8)
        for i in range(psz[0]):
            for j in range(psz[1]):
                win = y[:, :, i * pool[0]:min((i + 1) * pool[0], H), j * pool[1]:min((j + 1) * pool[1], W)]
                cnt = win.shape[2] * win.shape[3]
                acc = win.reshape(N, C, -1).sum(2).astype(np.float32)
                mult = np.float32(1.0 / cnt)
                q = (np.rint(np.float32(11) + acc * (np.float32(1) / (np.float32(1) / mult))) if requant == abi.REQUANT_ZP_INSIDE
                     else np.rint(acc * mult) + 11)
                ref[:, :, i, j] = np.clip(q, 0, 255).astype(np.uint8)
        op = abi.forward_quantized_pooled(torch.from_numpy(xq).to(DEV), torch.from_numpy(wq).to(DEV), 128, 11, pad, pool, requant=requant)
        count[abi.last_kernel() + "/u8"] += 1
        assert np.array_equal(op.cpu().numpy(), ref), ("qpool", shape, pool, pad, requant, abi.last_kernel())


def case_step(rs):
    """round 3: the one-step kernels (shiftnd_step.hip) -- 2-D problems whose rows are whole 16-byte pieces, every float
    dtype, forced through knobs 32-34; the forward also with crops"""
    tdt = [torch.float32, torch.float64, torch.float16, torch.bfloat16][rs.randint(4)]
    es = torch.empty(0, dtype=tdt).element_size()
    per16 = 16 // es
    H = int(rs.choice([1, 2, 5, 9, 18, 37, 64, 113]))
    W = per16 * int(rs.choice([1, 2, 3, 7, 14, 28, 56, 64, 100, 256][: (10 if es < 8 else 9)]))
    N, C = int(rs.randint(1, 4)), int(rs.randint(1, 6))
    shape = (N, C, H, W)
    pad = int(rs.randint(0, 5)); active = int(rs.randint(0, 2))
    x32 = rs.uniform(-1, 1, size=shape); g32 = rs.uniform(-1, 1, size=shape)
    xt, gt = torch.from_numpy(x32).to(tdt), torch.from_numpy(g32).to(tdt)
    wt = torch.from_numpy(weights(rs, C, 2, shape[2:], 4.5)).to(tdt)
    wide = np.float64 if tdt == torch.float64 else np.float32
    x, go, w = xt.to(torch.float64).numpy().astype(wide), gt.to(torch.float64).numpy().astype(wide), wt.to(torch.float64).numpy().astype(wide)
    xd, gd, wd = xt.to(DEV), gt.to(DEV), wt.to(DEV)
    exact = tdt in (torch.float32, torch.float64)
    from test_hip_parity import _ulp_close
    for k in (32, 33, 34):
        abi.set_tuning(k, 2)
    try:
        gx, gw = abi.backward(gd, wd, xd, pad, active)
        assert abi.last_kernel() == "step_backward", (shape, tdt, abi.last_kernel())
        count["step_backward"] += 1
        gx_o = torch.from_numpy(O.backward(go, w, x, pad, active)[0]).to(tdt)
        if exact or not active:
            assert torch.equal(gx.cpu(), gx_o), ("step gx", shape, tdt, pad, active)
        else:
            assert _ulp_close(gx.cpu(), gx_o, tdt, 32 * 2.0 ** -24), ("step gx", shape, tdt, pad, active)
        _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active)
        tol = {torch.float64: 1e-12, torch.float32: 1e-5}.get(tdt, 0.51 * float(torch.finfo(tdt).eps))   # (cases.gw16_tol: one rounding)
        assert rel_err(gw.to(torch.float64).cpu().numpy(), gw64) < tol, ("step gw", shape, tdt, pad, active)
        # forward, with a random crop that keeps the output rows whole 16-byte pieces
        crop = None
        if rs.randint(2) and H > 3 and W > 2 * per16:
            crop = [[int(rs.randint(0, 2)), int(rs.randint(0, 2))], [per16 * int(rs.randint(0, 2)), per16 * int(rs.randint(0, 2))]]
        b, new = abi.check_borders(list(shape), crop, 2)
        o = abi.forward(xd, wd, pad, active, b)
        count[abi.last_kernel()] += 1
        assert abi.last_kernel().startswith("step_"), (shape, tdt, crop, abi.last_kernel())
        ref = torch.from_numpy(O.forward(x, w, pad, active, b)).to(tdt)
        if exact or not active:
            assert torch.equal(o.cpu(), ref), ("step fwd", shape, tdt, pad, active, crop)
        else:
            assert _ulp_close(o.cpu(), ref, tdt, 32 * 2.0 ** -24), ("step fwd", shape, tdt, pad, active, crop)
    finally:
        for k in (32, 33, 34):
            abi.set_tuning(k, 0)


last = {}


def case_walk(rs):
    """round 3: the walk through the planes (walk_forward / walk_backward / walk_backward<POOL>) -- 3-D interpolating problems
    whose rows are whole 16-byte pieces, every float dtype (knob 35 bit 5), every padding; the pooled backward against the
    oracle's fused backward (fp32) or the two-step sequence on widened values (16-bit)"""
    tdt = [torch.float32, torch.float64, torch.float16, torch.bfloat16][rs.randint(4)]
    es = torch.empty(0, dtype=tdt).element_size()
    per16 = 16 // es
    D = int(rs.choice([2, 3, 5, 8, 17])); H = int(rs.choice([1, 2, 5, 9, 18, 37, 64]))
    W = per16 * int(rs.choice([1, 2, 3, 7, 14, 28, 64]))
    N, C = int(rs.randint(1, 3)), int(rs.randint(1, 5))
    shape = (N, C, D, H, W)
    pad = int(rs.randint(0, 5))
    xt = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt); gt = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt)
    wt = torch.from_numpy(weights(rs, C, 3, shape[2:], 4.5)).to(tdt)
    wide = np.float64 if tdt == torch.float64 else np.float32
    x, go, w = (t.to(torch.float64).numpy().astype(wide) for t in (xt, gt, wt))
    xd, gd, wd = xt.to(DEV), gt.to(DEV), wt.to(DEV)
    exact = tdt in (torch.float32, torch.float64)
    from test_hip_parity import _ulp_close
    last.update(shape=shape, tdt=tdt, pad=pad, xt=xt, gt=gt, wt=wt)   # (for a post-mortem: tools/_dbg.py)
    abi.set_tuning(35, 32)
    try:
        o = abi.forward(xd, wd, pad, 1)
        assert abi.last_kernel() == ("walk_forward16" if es == 2 else "walk_forward"), (shape, tdt, abi.last_kernel())
        count[abi.last_kernel()] += 1
        ref = torch.from_numpy(O.forward(x, w, pad, 1)).to(tdt)
        # 16-bit: 1 ulp of the type, plus 32 fp32 ulps of the (unit-scale) operands for results that cancel to almost nothing
        # (seven nested blends in 3-D, evaluated with mul + fma here and mul, mul, add in the oracle)
        floor16 = 32 * 2.0 ** -24
        assert torch.equal(o.cpu(), ref) if exact else _ulp_close(o.cpu(), ref, tdt, floor16), ("walk fwd", shape, tdt, pad)
        gx, gw = abi.backward(gd, wd, xd, pad, 1)
        assert abi.last_kernel() == ("walk_backward16" if es == 2 else "walk_backward"), (shape, tdt, abi.last_kernel())
        count[abi.last_kernel()] += 1
        gx_o = torch.from_numpy(O.backward(go, w, x, pad, 1)[0]).to(tdt)
        assert torch.equal(gx.cpu(), gx_o) if exact else _ulp_close(gx.cpu(), gx_o, tdt, floor16), ("walk gx", shape, tdt, pad)
        _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, 1)
        tol = {torch.float64: 1e-12, torch.float32: 1e-5}.get(tdt, 0.51 * float(torch.finfo(tdt).eps))   # (cases.gw16_tol: one rounding)
        assert rel_err(gw.to(torch.float64).cpu().numpy(), gw64) < tol, ("walk gw", shape, tdt, pad)
        gx, gw = abi.backward(gd, wd, xd, pad, 0)   # the sparse shift through the same walk
        assert abi.last_kernel() == ("walk_backward16_sparse" if es == 2 else "walk_backward_sparse"), (shape, tdt, abi.last_kernel())
        count[abi.last_kernel()] += 1
        assert torch.equal(gx.cpu(), torch.from_numpy(O.backward(go, w, x, pad, 0)[0]).to(tdt)), ("walk sparse gx", shape, tdt, pad)
        _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, 0)
        assert rel_err(gw.to(torch.float64).cpu().numpy(), gw64) < tol, ("walk sparse gw", shape, tdt, pad)
        if tdt != torch.float64:
            pool = (int(rs.randint(1, 4)), int(rs.randint(1, 4)), 2)
            y = O.forward(x, w, pad, 1)
            gpt = torch.from_numpy(rs.uniform(-1, 1, size=O.avg_pool(y, pool).shape)).to(tdt)
            outp = abi.forward_pooled(xd, wd, pad, 1, pool)
            fwalk = pool[1] == 1 or (pool[1] == 2 and H >= 2)
            assert abi.last_kernel() == ("walk_forward_pool" if fwalk else "plane_pool_forward"), (shape, tdt, pool, abi.last_kernel())
            count[abi.last_kernel()] += 1
            if tdt == torch.float32:
                assert np.array_equal(outp.cpu().numpy(), O.forward_pooled(x, w, pad, 1, pool)), ("walk pool fwd", shape, pool, pad)
            else:
                refp = O.avg_pool(torch.from_numpy(y).to(tdt).float().numpy(), pool)
                assert np.max(np.abs(outp.float().cpu().numpy() - refp)) <= float(torch.finfo(tdt).eps) * max(1.0, np.max(np.abs(refp))), ("walk pool fwd", shape, tdt, pool, pad)
            gx, gw = abi.backward_pooled(gpt.to(DEV), wd, xd, pad, 1, pool)
            assert abi.last_kernel() == "walk_backward_pool", (shape, tdt, pool, abi.last_kernel())
            count["walk_backward_pool"] += 1
            if tdt == torch.float32:
                gx_r, _ = O.backward_pooled(gpt.numpy(), w, x, pad, 1, pool)
                assert np.array_equal(gx.cpu().numpy(), gx_r), ("walk pool gx", shape, pool, pad)
                _, gw_r = O.backward_pooled(gpt.numpy().astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, 1, pool)
                assert rel_err(gw.cpu().numpy(), gw_r) < 1e-5, ("walk pool gw", shape, pool, pad)
            else:
                eps = float(torch.finfo(tdt).eps)
                g = torch.from_numpy(O.avg_pool_backward(gpt.float().numpy(), pool, y.shape[2:])).to(tdt).float().numpy()
                gx_r, gw_r = O.backward(g, w, x, pad, 1)
                assert np.max(np.abs(gx.float().cpu().numpy() - gx_r)) <= eps * max(1.0, np.max(np.abs(gx_r))), ("walk pool gx", shape, tdt, pool, pad)
                assert rel_err(gw.float().cpu().numpy(), gw_r) < 4 * eps, ("walk pool gw", shape, tdt, pool, pad)
            # ... and the sparse shift's pooled backward (one gradient tap: grad_x is a copy of the expanded gradient)
            gx, gw = abi.backward_pooled(gpt.to(DEV), wd, xd, pad, 0, pool)
            assert abi.last_kernel() == "walk_backward_pool", (shape, tdt, pool, abi.last_kernel())
            g = torch.from_numpy(O.avg_pool_backward(gpt.float().numpy(), pool, y.shape[2:])).to(tdt).to(torch.float64).numpy().astype(wide)
            gx_r, _ = O.backward(g, w, x, pad, 0)
            assert torch.equal(gx.cpu(), torch.from_numpy(gx_r).to(tdt)), ("walk sparse pool gx", shape, tdt, pool, pad)
            _, gw_r = O.backward(g.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, 0)
            assert rel_err(gw.to(torch.float64).cpu().numpy(), gw_r) < tol, ("walk sparse pool gw", shape, tdt, pool, pad)
    finally:
        abi.set_tuning(35, 0)


def _random_crop(rs, sizes):
    """borders argument: per dim [left cut, right cut], each side 0 .. a third of the dim, at least one element kept"""
    crop = []
    for s in sizes:
        l = int(rs.randint(0, max(1, s // 3) + 1)) if rs.randint(3) else 0
        r = int(rs.randint(0, max(1, s // 3) + 1)) if rs.randint(3) else 0
        if l + r >= s:
            l, r = 0, 0
        crop.append([l, r])
    return crop


def _check_float(tag, tdt, got, ref, exact, floor=0.0):
    from test_hip_parity import _ulp_close
    refd = torch.from_numpy(ref).to(tdt)
    if exact:
        assert torch.equal(got.cpu(), refd), tag
    else:
        assert _ulp_close(got.cpu(), refd, tdt, floor), tag


def case_crop(rs):
    """round 4: windows (crops) and Shift1d through the lean one-step kernels of shiftnd_span.hip -- 1-D / 2-D problems whose x rows
    are whole 16-byte pieces, every float dtype, every padding, both shifts, random asymmetric cuts; default routing (which
    kernel took a call is counted, the routing itself is pinned by tests/test_span_gpu.py and tests/test_routing_gpu.py)"""
    tdt = [torch.float32, torch.float64, torch.float16, torch.bfloat16][rs.randint(4)]
    es = torch.empty(0, dtype=tdt).element_size()
    per16 = 16 // es
    nd = 1 if rs.randint(3) == 0 else 2
    if nd == 1:
        sp = (per16 * int(rs.choice([1, 3, 16, 128, 130, 257, 300, 513, 1024])),)
    else:
        sp = (int(rs.choice([1, 2, 5, 9, 18, 37, 64])), per16 * int(rs.choice([1, 2, 3, 7, 14, 16, 28, 56, 100, 254])))
    N, C = int(rs.randint(1, 4)), int(rs.randint(1, 6))
    shape = (N, C) + sp
    crop = _random_crop(rs, sp) if (nd == 2 or rs.randint(2)) else None
    b, new = abi.check_borders(list(shape), crop, nd)
    total = int(np.prod(new))
    pad = int(rs.randint(0, 5)); active = int(rs.randint(0, 2))
    xt = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt); gt = torch.from_numpy(rs.uniform(-1, 1, size=new)).to(tdt)
    wt = torch.from_numpy(weights(rs, C, nd, sp, 4.5)).to(tdt)
    wide = np.float64 if tdt == torch.float64 else np.float32
    x, go, w = (t.to(torch.float64).numpy().astype(wide) for t in (xt, gt, wt))
    xd, gd, wd = xt.to(DEV), gt.to(DEV), wt.to(DEV)
    exact = tdt in (torch.float32, torch.float64) or not active
    cropped = list(new) != list(shape)
    o = abi.forward(xd, wd, pad, active, b)
    count[abi.last_kernel()] += 1
    _check_float(("crop fwd", shape, crop, tdt, pad, active, abi.last_kernel()), tdt, o, O.forward(x, w, pad, active, b), exact, 32 * 2.0 ** -24)
    gx, gw = abi.backward(gd, wd, xd, pad, active, b)
    count[abi.last_kernel()] += 1
    _check_float(("crop gx", shape, crop, tdt, pad, active, abi.last_kernel()), tdt, gx, O.backward(go, w, x, pad, active, b)[0], exact, 32 * 2.0 ** -24)
    _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active, b)
    tol = {torch.float64: 1e-12, torch.float32: 1e-5}.get(tdt, 0.51 * float(torch.finfo(tdt).eps))
    if tdt == torch.float32:   # (a sum that cancels: the fp32 evaluation's own error is the bar -- tests/test_span_gpu.py)
        tol = max(tol, 2 * rel_err(O.backward(go, w, x, pad, active, b)[1], gw64))
    e = rel_err(gw.to(torch.float64).cpu().numpy(), gw64)
    assert e < tol, ("crop gw", shape, crop, tdt, pad, active, abi.last_kernel(), e)


def case_cl_crop(rs):
    """round 4: the LDS-tiled channels-last kernels with a window -- forward (both output layouts), interpolating forward,
    backward with either gradient layout; fp32, and quint8 keeping the format"""
    C = int(rs.choice([4, 8, 12, 32, 36, 64, 100]))
    H = int(rs.choice([5, 6, 9, 17, 33])); W = int(rs.randint(2, 41))
    N = int(rs.randint(1, 4))
    shape = (N, C, H, W)
    crop = _random_crop(rs, (H, W))
    b, new = abi.check_borders(list(shape), crop, 2)
    pad = int(rs.randint(0, 5)); active = int(rs.randint(0, 2))
    x = rs.uniform(-1, 1, size=shape).astype(np.float32); go = rs.uniform(-1, 1, size=new).astype(np.float32)
    w = weights(rs, C, 2, shape[2:], 3.9).astype(np.float32)
    cl = torch.channels_last
    xd = torch.from_numpy(x).to(DEV).contiguous(memory_format=cl)
    wd = torch.from_numpy(w).to(DEV)
    abi.set_tuning(21, int(rs.choice([0, 0, 3, 7])))
    try:
        ref = O.forward(x, w, pad, active, b)
        for out in (None, torch.empty(new, device=DEV).contiguous(memory_format=cl)):
            o = abi.forward(xd, wd, pad, active, b, out=out)
            count[abi.last_kernel() + "/crop"] += 1
            assert np.array_equal(o.cpu().numpy(), ref), ("cl crop fwd", shape, crop, pad, active, abi.last_kernel())
        gd = torch.from_numpy(go).to(DEV)
        if rs.randint(2):
            gd = gd.contiguous(memory_format=cl)
        gx, gw = abi.backward(gd, wd, xd, pad, active, b, grad_x=torch.empty(shape, device=DEV).contiguous(memory_format=cl))
        count[abi.last_kernel() + "/crop"] += 1
        gx_o, _ = O.backward(go, w, x, pad, active, b)
        _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active, b)
        assert np.array_equal(gx.cpu().numpy(), gx_o), ("cl crop gx", shape, crop, pad, active, abi.last_kernel())
        tol = max(1e-5, 2 * rel_err(O.backward(go, w, x, pad, active, b)[1], gw64))
        assert rel_err(gw.cpu().numpy(), gw64) < tol, ("cl crop gw", shape, crop, pad, active, abi.last_kernel())
        if C % 16 == 0:
            xq = rs.randint(0, 256, size=shape).astype(np.uint8); wq = rs.randint(122, 135, size=(C, 2)).astype(np.uint8)
            oq = torch.empty(new, dtype=torch.uint8, device=DEV).contiguous(memory_format=cl)
            abi.forward_quantized(torch.from_numpy(xq).to(DEV).contiguous(memory_format=cl), torch.from_numpy(wq).to(DEV), 128, 3, pad, b, out=oq)
            count[abi.last_kernel() + "/crop/u8"] += 1
            assert np.array_equal(oq.cpu().numpy(), O.forward_q(xq, wq, 128, 3, pad, b)), ("cl crop u8", shape, crop, pad)
    finally:
        abi.set_tuning(21, 0)


def case_cl3d(rs):
    """round 4: NDHWC forwards (cl_tiled_forward_3d: sparse / quantized, cl_tiled_active_forward_3d: the plane blend at staging time);
    fp32 bit-exact both ways, int32 quantized, random windows in all three dims, any depth shift"""
    C = int(rs.choice([4, 8, 12, 32, 36, 64]))
    D = int(rs.choice([1, 2, 3, 5, 8])); H = int(rs.choice([1, 5, 6, 9, 17])); W = int(rs.randint(1, 30))
    N = int(rs.randint(1, 3))
    shape = (N, C, D, H, W)
    crop = _random_crop(rs, (D, H, W)) if rs.randint(2) else None
    b, new = abi.check_borders(list(shape), crop, 3)
    pad = int(rs.randint(0, 5)); active = int(rs.randint(0, 2))
    x = rs.uniform(-1, 1, size=shape).astype(np.float32)
    w = weights(rs, C, 3, shape[2:], 3.9).astype(np.float32)
    cl3 = torch.channels_last_3d
    xd = torch.from_numpy(x).to(DEV).contiguous(memory_format=cl3)
    wd = torch.from_numpy(w).to(DEV)
    abi.set_tuning(21, int(rs.choice([0, 0, 3, 7])))
    try:
        ref = O.forward(x, w, pad, active, b)
        for out in (None, torch.empty(new, device=DEV).contiguous(memory_format=cl3)):
            o = abi.forward(xd, wd, pad, active, b, out=out)
            count[abi.last_kernel()] += 1
            if D * H * W > 1:   # (a 1 x 1 x 1 volume is contiguous and channels-last at once: the contiguous kernels take it)
                assert abi.last_kernel() == ("cl_tiled_active_forward_3d" if active else "cl_tiled_forward_3d"), (shape, crop, abi.last_kernel())
            assert np.array_equal(o.cpu().numpy(), ref), ("cl3d fwd", shape, crop, pad, active)
        xq = rs.randint(-500, 500, size=shape).astype(np.int32); wq = rs.randint(122, 135, size=(C, 3)).astype(np.uint8)
        oq = torch.empty(new, dtype=torch.int32, device=DEV).contiguous(memory_format=cl3)
        abi.forward_quantized(torch.from_numpy(xq).to(DEV).contiguous(memory_format=cl3), torch.from_numpy(wq).to(DEV), 128, 3, pad, b, out=oq)
        count[abi.last_kernel() + "/i32"] += 1
        assert np.array_equal(oq.cpu().numpy(), O.forward_q(xq, wq, 128, 3, pad, b)), ("cl3d i32", shape, crop, pad)
    finally:
        abi.set_tuning(21, 0)


CASES = [case_cl, case_ragged, case_ragged, case_bytes, case_step, case_step, case_walk, case_walk, case_crop, case_crop, case_cl_crop, case_cl3d]



def case_ragged3d(rs):
    """round 5: 3-D volumes whose rows are not whole 16-byte pieces, beyond the small-plane kernels (plane_ragged_*: the plane kernels
    with 4- / 8-byte chunks) and below them (small_plane_*): interpolating forward, both backwards, every padding, windows"""
    from test_hip_parity import _ulp_close
    tdt = [torch.float32, torch.float64, torch.float16, torch.bfloat16][rs.randint(4)]
    es = torch.empty(0, dtype=tdt).element_size()
    while True:
        D = int(rs.choice([2, 3, 5, 9, 16])); H = int(rs.choice([3, 7, 14, 20, 30])); W = int(rs.choice([6, 7, 10, 14, 15, 22, 28, 30, 31, 62, 110]))
        if (W * es) % 16 and D * H * W * es <= 96 * 1024:   # (fp64: the odd widths)
            break
    N, C = int(rs.randint(1, 3)), int(rs.randint(1, 4))
    shape = (N, C, D, H, W)
    crop = _random_crop(rs, shape[2:]) if rs.rand() < 0.3 else None
    b, new = abi.check_borders(list(shape), crop, 3)
    pad = int(rs.randint(0, 5))
    xt = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt); gt = torch.from_numpy(rs.uniform(-1, 1, size=new)).to(tdt)
    wt = torch.from_numpy(weights(rs, C, 3, shape[2:], 3.5)).to(tdt)
    wide = np.float64 if tdt == torch.float64 else np.float32
    x, go, w = (t.to(torch.float64).numpy().astype(wide) for t in (xt, gt, wt))
    xd, gd, wd = xt.to(DEV), gt.to(DEV), wt.to(DEV)
    exact = tdt in (torch.float32, torch.float64)
    last.update(shape=shape, tdt=tdt, pad=pad, xt=xt, gt=gt, wt=wt)
    floor16 = 32 * 2.0 ** -24
    o = abi.forward(xd, wd, pad, 1, b)
    count[abi.last_kernel()] += 1
    ref = torch.from_numpy(O.forward(x, w, pad, 1, b)).to(tdt)
    assert torch.equal(o.cpu(), ref) if exact else _ulp_close(o.cpu(), ref, tdt, floor16), ("ragged3d fwd", shape, crop, tdt, pad, abi.last_kernel())
    for active in (1, 0):
        gx, gw = abi.backward(gd, wd, xd, pad, active, b)
        k = abi.last_kernel()
        count[k] += 1
        gx_o = torch.from_numpy(O.backward(go, w, x, pad, active, b)[0]).to(tdt)
        ok = torch.equal(gx.cpu(), gx_o) if (exact or not active) else _ulp_close(gx.cpu(), gx_o, tdt, floor16)
        assert ok, ("ragged3d gx", shape, crop, tdt, pad, active, k)
        _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active, b)
        tol = {torch.float64: 1e-12, torch.float32: 1e-5}.get(tdt, 0.51 * float(torch.finfo(tdt).eps))
        assert rel_err(gw.to(torch.float64).cpu().numpy(), gw64) < tol, ("ragged3d gw", shape, crop, tdt, pad, active, k)


CASES += [case_ragged3d, case_ragged3d]


def case_crop3d(rs):
    """round 6: cropped 3-D volumes through crop_forward3 / crop_backward3 -- x rows of whole 16-byte pieces, every float dtype, every
    padding, both shifts, random asymmetric cuts per dim (shifts beyond the volume included)"""
    tdt = [torch.float32, torch.float64, torch.float16, torch.bfloat16][rs.randint(4)]
    es = torch.empty(0, dtype=tdt).element_size()
    per16 = 16 // es
    sp = (int(rs.choice([2, 3, 5, 8, 16])), int(rs.choice([2, 5, 9, 18, 37, 70])), per16 * int(rs.choice([1, 2, 3, 7, 14, 28, 56])))
    N, C = int(rs.randint(1, 3)), int(rs.randint(1, 5))
    shape = (N, C) + sp
    crop = _random_crop(rs, sp)
    b, new = abi.check_borders(list(shape), crop, 3)
    pad = int(rs.randint(0, 5)); active = int(rs.randint(0, 2))
    xt = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt); gt = torch.from_numpy(rs.uniform(-1, 1, size=new)).to(tdt)
    wt = torch.from_numpy(weights(rs, C, 3, sp, 4.5)).to(tdt)
    wide = np.float64 if tdt == torch.float64 else np.float32
    x, go, w = (t.to(torch.float64).numpy().astype(wide) for t in (xt, gt, wt))
    xd, gd, wd = xt.to(DEV), gt.to(DEV), wt.to(DEV)
    exact = tdt in (torch.float32, torch.float64) or not active
    o = abi.forward(xd, wd, pad, active, b)
    count[abi.last_kernel()] += 1
    _check_float(("crop3 fwd", shape, crop, tdt, pad, active, abi.last_kernel()), tdt, o, O.forward(x, w, pad, active, b), exact, 32 * 2.0 ** -24)
    gx, gw = abi.backward(gd, wd, xd, pad, active, b)
    count[abi.last_kernel()] += 1
    _check_float(("crop3 gx", shape, crop, tdt, pad, active, abi.last_kernel()), tdt, gx, O.backward(go, w, x, pad, active, b)[0], exact, 32 * 2.0 ** -24)
    _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active, b)
    tol = {torch.float64: 1e-12, torch.float32: 1e-5}.get(tdt, 0.51 * float(torch.finfo(tdt).eps))
    if tdt == torch.float32:
        tol = max(tol, 2 * rel_err(O.backward(go, w, x, pad, active, b)[1], gw64))
    e = rel_err(gw.to(torch.float64).cpu().numpy(), gw64)
    assert e < tol, ("crop3 gw", shape, crop, tdt, pad, active, abi.last_kernel(), e)


CASES += [case_crop3d, case_crop3d]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rs = np.random.RandomState(a.seed)
    t0 = time.time()
    n = 0
    while time.time() - t0 < a.seconds:
        CASES[n % len(CASES)](rs)
        n += 1
    print("cases", n, dict(count))
    print("fuzz ok")


if __name__ == "__main__":
    main()
