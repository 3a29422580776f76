"""GPU parity of the flat-stream kernels (csrc/shiftnd_flat.hip, round 5: contiguous 1-D / 2-D problems whose rows are not whole
16-byte pieces, as one-step workgroups over the tensor's flat chunk stream) against the CPU oracle, through the C ABI.

The shapes: 14 x 14 / 7 x 7 planes (the deep stages of an ImageNet network), odd sizes, 62 x 62 and 222 x 222 (the output of the
reference's cropped shift as the next layer's input: modules/shifts.py:41-46, tests/shifts_test.py:9-28), windows on ragged planes,
rows of one element, planes smaller and larger than a 4 KB step, tensors that are not a whole number of chunks, Shift1d rows of any
length.  Bars as everywhere: fp32 / fp64 forward and grad_x bit-exact, grad_w <= 1e-5 / 1e-12 of the fp64
evaluation; 16-bit: the sparse shift bit-exact, interpolation within 1 ulp, grad_w within half a unit of the type."""
import numpy as np
import pytest
import torch

from cases import rel_err, gw16_tol
from oracle import oracle as O
from test_hip_parity import _ulp_close, _weights

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TDT = {"f32": torch.float32, "f64": torch.float64, "f16": torch.float16, "bf16": torch.bfloat16}


@pytest.fixture()
def abi():
    from torchshifts import abi as A
    assert torch.cuda.is_available()
    A.set_path_policy(0)
    A.set_tuning(27, 2)   # the flat-stream kernels whenever eligible (the automatic choice: ragged rows only)
    yield A
    A.set_tuning(27, 0)


# (shape, cut): small planes (many per step), planes around the step size, large planes (row-range covers), windows, one-element
# rows / columns, tensors whose last chunk is partial, 1-D
CASES = [((3, 5, 14, 14), None), ((2, 7, 7, 7), None), ((5, 3, 3, 5), None), ((2, 3, 31, 33), None), ((1, 2, 62, 62), None),
         ((1, 2, 113, 113), None), ((1, 1, 225, 225), None), ((2, 2, 222, 222), None), ((1, 3, 62, 62), [[1, 1], [1, 1]]),
         ((2, 3, 14, 14), [[1, 1], [1, 1]]), ((1, 2, 64, 64), [[1, 2], [3, 0]]), ((3, 2, 9, 1), None), ((2, 3, 1, 23), None),
         ((1, 1, 7, 3), None), ((2, 2, 40, 100), [[0, 0], [1, 0]]), ((1, 2, 300, 6), [[100, 150], [1, 1]]), ((1, 2, 130, 12), [[1, 1], [1, 1]]),
         ((2, 4, 45, 45), [[2, 2], [2, 2]]), ((1, 1, 500, 300), None), ((4, 3, 27, 27), None),
         ((3, 5, 77), None), ((2, 2, 4100), None), ((2, 3, 1000), [[3, 5]]), ((1, 2, 33), [[0, 1]]),
         #  planes of two very wide ragged rows: covers within budget, covers + a whole row of slack beyond the 64 KiB
         # a launch may ask for -- the plan declines and the older kernels serve them
         ((2, 2, 2, 1501), None), ((1, 2, 2, 701), None)]


def flat_serves(shape, new, es, backward):
    """csrc/shiftnd_flat.hip: flat_plan restated -- small planes: the source planes a 4 KB step touches (at most 72) fit 24 KB of LDS
    per staged tensor; large planes: at most two per step and the source rows of a step fit.  Tinier planes and rows of several
    KB keep the older kernels (checked against the oracle all the same)."""
    S1, S2 = (shape[-2] if len(shape) == 4 else 1), shape[-1]
    O1, O2 = (new[-2] if len(new) == 4 else 1), new[-1]
    E = 16 // es
    xpb, opb = S1 * S2 * es, O1 * O2 * es
    spb = xpb if backward else opb
    nplanes = (4096 + spb - 1) // spb + 1
    cover = lambda payload: ((payload + 30) // 16) * 16
    rec = 56 if es == 8 else 48
    slack = ((max(0 if S1 == 1 else S2, 0 if O1 == 1 else O2) + 2) * es + 15) // 16 * 16
    partials = 256 * 4 * (8 if es == 8 else 4) if backward else 0

    def lds(planes_in_table, cx, cg):   # table + per-thread partials + covers + slack on both sides: at most 64 KiB per launch
        return 16 + (planes_in_table * rec + 15) // 16 * 16 + partials + cx + (cg if backward else 0) + 2 * slack
    if nplanes <= 72 and cover(nplanes * xpb) <= 24576 and (not backward or cover(nplanes * opb) <= 24576):
        return lds(72, cover(nplanes * xpb), cover(nplanes * opb)) <= 65536
    if spb < 4096:
        return False
    sr = S2 if backward else O2
    rows = (256 * E + sr - 2) // sr + 1
    cx, cg = cover((rows + 5) * S2 * es) + 32, cover((rows + 5) * O2 * es) + 32
    return cx <= 24576 and (not backward or cg <= 24576) and lds(2, cx, cg) <= 65536


@pytest.mark.parametrize("dt", ["f32", "f64", "f16", "bf16"])
@pytest.mark.parametrize("shape,crop", CASES)
def test_flat_forward_vs_oracle(abi, shape, crop, dt):
    tdt = TDT[dt]
    nd = len(shape) - 2
    b, new = abi.check_borders(list(shape), crop, nd)
    rs = np.random.RandomState(sum(shape) * 13 + 1)
    xt = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt)
    wt = torch.from_numpy(_weights(rs, shape[1], nd, shape[2:])).to(tdt)
    wide = np.float64 if tdt == torch.float64 else np.float32
    x, w = (t.to(torch.float64).numpy().astype(wide) for t in (xt, wt))
    xd, wd = xt.to(DEV), wt.to(DEV)
    es = xt.element_size()
    for pad in range(5):
        for active in (0, 1):
            out = abi.forward(xd, wd, pad, active, b)
            if flat_serves(shape, new, es, False):
                assert abi.last_kernel() == ("flat_active_forward" if active else "flat_gather_forward"), (shape, crop, abi.last_kernel())
            ref = torch.from_numpy(O.forward(x, w, pad, active, b)).to(tdt)
            if es >= 4 or not active:
                assert torch.equal(out.cpu(), ref), (shape, crop, dt, pad, active)
            else:
                assert _ulp_close(out.cpu(), ref, tdt), (shape, crop, dt, pad, active)


@pytest.mark.parametrize("dt", ["f32", "f64", "f16", "bf16"])
@pytest.mark.parametrize("shape,crop", CASES)
def test_flat_backward_vs_oracle(abi, shape, crop, dt):
    tdt = TDT[dt]
    nd = len(shape) - 2
    b, new = abi.check_borders(list(shape), crop, nd)
    rs = np.random.RandomState(sum(shape) * 7 + 3)
    xt = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt)
    gt = torch.from_numpy(rs.uniform(-1, 1, size=new)).to(tdt)
    wt = torch.from_numpy(_weights(rs, shape[1], nd, shape[2:])).to(tdt)
    wide = np.float64 if tdt == torch.float64 else np.float32
    x, go, w = (t.to(torch.float64).numpy().astype(wide) for t in (xt, gt, wt))
    xd, god, wd = xt.to(DEV), gt.to(DEV), wt.to(DEV)
    es = xt.element_size()
    for pad in range(5):
        for active in (0, 1):
            gx, gw = abi.backward(god, wd, xd, pad, active, b)
            if flat_serves(shape, new, es, True):
                assert abi.last_kernel() == "flat_backward", (shape, crop, abi.last_kernel())
            gx_ref = torch.from_numpy(O.backward(go, w, x, pad, active, b)[0]).to(tdt)
            if es >= 4 or not active:
                assert torch.equal(gx.cpu(), gx_ref), ("gx", shape, crop, dt, pad, active)
            else:
                assert _ulp_close(gx.cpu(), gx_ref, tdt), ("gx", shape, crop, dt, pad, active)
            _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active, b)
            tol = {"f32": 1e-5, "f64": 1e-12}.get(dt, gw16_tol(torch.finfo(tdt).eps))
            if dt == "f32":   # a window of a few hundred elements can cancel to a small sum: never looser than the reference's own fp32 (test_span_gpu.py)
                own = 2 * rel_err(O.backward(go, w, x, pad, active, b)[1], gw64)
                if own > tol:
                    print("relaxed grad_w bar", shape, crop, pad, active, own)
                    tol = own
            assert rel_err(gw.to(torch.float64).cpu().numpy(), gw64) < tol, ("gw", shape, crop, dt, pad, active)
            gx2, gw2 = abi.backward(god, wd, xd, pad, active, b)
            assert torch.equal(gx, gx2) and torch.equal(gw, gw2)  # deterministic


def test_flat_default_route_for_ragged_rows():
    """untouched knobs: ragged rows take the flat-stream kernels, aligned rows keep the chunk kernels"""
    from torchshifts import abi
    abi.set_path_policy(0)
    abi.set_tuning(27, 0)
    # (shape, the forward's family, the backward's kernel: fp32 planes with rows of at least 8 chunks take the row-relative crop_backward)
    for shape, flat, bwd in (((4, 8, 14, 14), True, "flat_backward"), ((2, 4, 62, 62), True, "crop_backward_ragged"), ((1, 2, 222, 222), True, "crop_backward_ragged"),
                             ((2, 4, 64, 64), False, "step_backward"), ((2, 3, 1001), True, "flat_backward"), ((2, 3, 1024), False, None)):
        nd = len(shape) - 2
        x = torch.rand(shape, device=DEV)
        w = torch.rand(shape[1], nd, device=DEV) * 4 - 2
        for active in (0, 1):
            abi.forward(x, w, 0, active)
            big = bwd == "crop_backward_ragged"   # (fp32 rows of at least 8 chunks: the row-relative kernels of shiftnd_span.hip)
            assert abi.last_kernel().startswith("ragged_" if big else "flat_") == flat, (shape, active, abi.last_kernel())
            abi.backward(torch.rand(shape, device=DEV), w, x, 0, active)
            if bwd is not None:
                assert abi.last_kernel() == bwd, (shape, active, abi.last_kernel())
    # 16-bit ragged rows: an even number of elements per row (rows at 4-byte boundaries) take the row-relative kernels too, an odd
    # number the flat stream (2-byte-aligned 16-byte stores are slow)
    for width, name in ((70, "crop_backward_ragged"), (67, "flat_backward")):
        xh = torch.rand(2, 4, 62, width, device=DEV).half()
        wh = (torch.rand(4, 2, device=DEV) * 4 - 2).half()
        abi.backward(torch.rand(2, 4, 62, width, device=DEV).half(), wh, xh, 0, 0)
        assert abi.last_kernel() == name, (width, abi.last_kernel())
