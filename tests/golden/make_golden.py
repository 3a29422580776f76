#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Runs only in the build container: it loads oracle/_ref/_C.so (the reference's own C++ CPU,
QuantizedCPU, Autograd and composite ops, built in place from /root/reference by
oracle/build_ref.sh) and records inputs + outputs as small .npz files.  The fixtures are data
(inputs and expected outputs); no reference source travels.

    bash oracle/build_ref.sh && python tests/golden/make_golden.py

Fixtures
  g1_float.npz     nD{1,2,3} x pad{0..4} x active{F,T} x borders{none,crop} x {f32,f64}:
                   out, grad_x, grad_w for fixed x / grad_out / w (SURVEY.md section 8c "G1").
                   Inputs are multiples of 1/8 and weight fractions multiples of 1/4, so every
                   product and partial sum is exact: grad_w is independent of summation order and
                   a parallel reduction must reproduce it bit for bit.
  g1_random.npz    same grid on uniform-random fp64/fp32 data (order-dependent sums: fp64 results
                   are the truth for tolerance checks), single-threaded reference.
  g2_quant.npz     quint8/qint8/qint32 inputs, quint8(zp128)/qint8(zp0) weights, 5 paddings,
                   NCHW and channels-last, with and without crop ("G2").
  ka_1d.npz        the 1-D known-answer vectors of SURVEY.md section 8c.
  borders.npz      check_borders edge cases (Q12) observed through output windows.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "_C.so")

torch.set_num_threads(1)  # the reference's CPU backward races on grad_w with >1 thread
torch.ops.load_library(REF_SO)
OPS = torch.ops.torchshifts

SHAPES = {1: (2, 8, 7), 2: (2, 8, 7, 5), 3: (2, 8, 7, 5, 4)}
CROPS = {1: [[1, 2]], 2: [[1, 2], [0, 1]], 3: [[1, 2], [0, 1], [1, 0]]}
# per channel x dim: 0, -0.0, halves (round-half-even), quarters, shifts beyond the dim (multi-wrap)
WTABLE = np.array([[0.0, -0.0, 0.5],
                   [0.5, -0.5, 1.5],
                   [-1.5, 2.5, -2.5],
                   [0.25, -1.75, 1.75],
                   [9.25, -7.5, 6.0],
                   [-0.25, 3.0, -11.75],
                   [1.0, -1.0, 2.0],
                   [-3.5, 4.5, 0.75]])


def op(nd):
    return getattr(OPS, "shift%dd" % nd)


def run_float(x, w, pad, active, crop, go_full):
    """x, w numpy -> out, grad_x, grad_w through the reference's public op + autograd."""
    xt = torch.from_numpy(x.copy()).requires_grad_(True)
    wt = torch.from_numpy(w.copy()).requires_grad_(True)
    b = torch.Tensor() if crop is None else torch.tensor(crop, dtype=torch.long)
    out = op(x.ndim - 2)(xt, wt, b, pad, active)
    sl = tuple(slice(0, s) for s in out.shape)
    go = torch.from_numpy(np.ascontiguousarray(go_full[sl]))
    out.backward(go)
    return out.detach().numpy(), xt.grad.numpy(), wt.grad.numpy(), go.numpy()


def gen_float(exact, seed):
    rs = np.random.RandomState(seed)
    d = {}
    for nd in (1, 2, 3):
        shape = SHAPES[nd]
        if exact:
            x64 = rs.randint(-64, 65, size=shape) / 8.0
            go64 = rs.randint(-32, 33, size=shape) / 8.0
            w64 = WTABLE[:, :nd].copy()
        else:
            x64 = rs.uniform(-1, 1, size=shape)
            go64 = rs.uniform(-1, 1, size=shape)
            w64 = rs.uniform(-4, 4, size=(shape[1], nd))
            w64[0, :] = [0.5, 1.5, -2.5][:nd]  # keep exact halves in the random set too
        for dt in ("f32", "f64"):
            npdt = np.float32 if dt == "f32" else np.float64
            x, go_full, w = x64.astype(npdt), go64.astype(npdt), w64.astype(npdt)
            d["x_%dd_%s" % (nd, dt)] = x
            d["go_%dd_%s" % (nd, dt)] = go_full
            d["w_%dd_%s" % (nd, dt)] = w
            for pad in range(5):
                for active in (0, 1):
                    for ci, crop in enumerate((None, CROPS[nd])):
                        out, gx, gw, _ = run_float(x, w, pad, bool(active), crop, go_full)
                        key = "%dd_%s_p%d_a%d_c%d" % (nd, dt, pad, active, ci)
                        d["out_" + key] = out
                        d["gx_" + key] = gx
                        d["gw_" + key] = gw
    d["crops_1d"] = np.array(CROPS[1])
    d["crops_2d"] = np.array(CROPS[2])
    d["crops_3d"] = np.array(CROPS[3])
    return d


def gen_quant(seed):
    rs = np.random.RandomState(seed)
    d = {}
    qdt = {"quint8": (torch.quint8, 0, 255, 7), "qint8": (torch.qint8, -128, 127, -3),
           "qint32": (torch.qint32, -100000, 100000, 11)}
    for nd in (1, 2, 3):
        shape = SHAPES[nd]
        wf = np.round(WTABLE[:, :nd] * 2) / 2  # .0 / .5 values: quantize_per_tensor rounds half-even
        wq_defs = {"wu8": (torch.quint8, 128), "wi8": (torch.qint8, 0)}
        wq_t = {}
        for wname, (wdt, wzp) in wq_defs.items():
            wq = torch.quantize_per_tensor(torch.from_numpy(wf).float(), 1.0, wzp, wdt)
            wq_t[wname] = wq
            d["wq_%dd_%s" % (nd, wname)] = wq.int_repr().numpy()
            d["wzp_%s" % wname] = np.array(wzp)
        for xname, (xdt, lo, hi, xzp) in qdt.items():
            xi = rs.randint(lo, hi + 1, size=shape)
            xq = torch._make_per_tensor_quantized_tensor(
                torch.from_numpy(xi).to({torch.quint8: torch.uint8, torch.qint8: torch.int8,
                                         torch.qint32: torch.int32}[xdt]), 0.05, xzp)
            d["xq_%dd_%s" % (nd, xname)] = xq.int_repr().numpy()
            d["xzp_%s" % xname] = np.array(xzp)
            layouts = ["nchw"]
            if nd == 2:
                layouts.append("cl")
            if nd == 3:
                layouts.append("cl3d")
            for layout in layouts:
                xin = xq
                if layout == "cl":
                    xin = xq.contiguous(memory_format=torch.channels_last)
                elif layout == "cl3d":
                    xin = xq.contiguous(memory_format=torch.channels_last_3d)
                for wname, wq in wq_t.items():
                    for pad in range(5):
                        for ci, crop in enumerate((None, CROPS[nd])):
                            b = torch.Tensor() if crop is None else torch.tensor(crop, dtype=torch.long)
                            out = op(nd)(xin, wq, b, pad, False)
                            assert out.q_zero_point() == xzp and abs(out.q_scale() - 0.05) < 1e-9
                            key = "%dd_%s_%s_%s_p%d_c%d" % (nd, xname, layout, wname, pad, ci)
                            d["out_" + key] = out.int_repr().contiguous().numpy()
                            d["outcl_" + key] = np.array(
                                out.is_contiguous() if layout == "nchw" else
                                out.is_contiguous(memory_format=torch.channels_last if nd == 2
                                                  else torch.channels_last_3d))
    return d


def gen_known_answer():
    d = {}
    x = np.array([1, 2, 4, 8, 16, 32], np.float32).reshape(1, 1, 6)
    g = np.array([1, 10, 100, 1e3, 1e4, 1e5], np.float32).reshape(1, 1, 6)
    d["x"], d["g"] = x, g
    ws = [0.25, -0.25, 1.25, -1.75, 0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 7.25, -13.5]
    d["ws"] = np.array(ws, np.float32)
    for wi, wv in enumerate(ws):
        w = np.array([[wv]], np.float32)
        for pad in range(5):
            for active in (0, 1):
                out, gx, gw, _ = run_float(x, w, pad, bool(active), None, g)
                key = "w%d_p%d_a%d" % (wi, pad, active)
                d["out_" + key], d["gx_" + key], d["gw_" + key] = out, gx, gw
    # index maps: source index for idx = i - shift, observed with an identity ramp
    for length in (1, 2, 3, 5):
        ramp = np.arange(1, length + 1, dtype=np.float64).reshape(1, 1, length)  # 0 = fill
        for pad in range(5):
            rows = []
            for s in range(-12, 13):
                o = op(1)(torch.from_numpy(ramp), torch.tensor([[float(s)]], dtype=torch.float64),
                          torch.Tensor(), pad, False).numpy().ravel()
                rows.append(o - 1)  # -1 = fill, else source index
            d["map_len%d_p%d" % (length, pad)] = np.array(rows, np.int64)
    return d


def gen_borders():
    d = {}
    cases = [[[3, 3], [0, 0]], [[4, 4], [1, 1]], [[6, 0], [0, 0]], [[0, 6], [0, 0]], [[1, 2], [0, 1]],
             [[0, 0], [5, 5]], [[2, 9], [0, 0]], [[0, 0], [0, 0]], [[5, 0], [0, 5]]]
    x = np.arange(1, 2 * 3 * 6 * 6 + 1, dtype=np.float64).reshape(2, 3, 6, 6)
    w = np.zeros((3, 2))
    d["x"] = x
    d["cases"] = np.array(cases)
    for i, c in enumerate(cases):
        out = OPS.shift2d(torch.from_numpy(x), torch.from_numpy(w), torch.tensor(c, dtype=torch.long), 0, False)
        d["out_%d" % i] = out.numpy()
    # 1-D and 3-D, rank-(dim+1) input (no batch dim is NOT supported by kernels; shape logic only)
    x3 = np.arange(1, 2 * 2 * 4 * 5 * 6 + 1, dtype=np.float64).reshape(2, 2, 4, 5, 6)
    c3 = [[1, 1], [5, 0], [0, 6]]
    d["x3"], d["case3"] = x3, np.array(c3)
    d["out3"] = OPS.shift3d(torch.from_numpy(x3), torch.zeros(2, 3, dtype=torch.float64),
                            torch.tensor(c3, dtype=torch.long), 0, False).numpy()
    return d


def main():
    out = {
        "g1_float.npz": gen_float(True, 1234),
        "g1_random.npz": gen_float(False, 4321),
        "g2_quant.npz": gen_quant(99),
        "ka_1d.npz": gen_known_answer(),
        "borders.npz": gen_borders(),
    }
    for name, d in out.items():
        path = os.path.join(HERE, name)
        np.savez_compressed(path, **d)
        print("%-16s %4d arrays %8.1f KB" % (name, len(d), os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    sys.exit(main())
