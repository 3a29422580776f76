"""GPU parity of the span kernels (csrc/shiftnd_span.hip: cropped 2-D windows and 1-D rows of any length as one-step workgroups
over row spans) against the CPU oracle, through the C ABI.  The shapes are the reference's everyday cases: the depthwise-conv
emulation's cut of one element per side (modules/shifts.py:41-46; the reference's own test script, tests/shifts_test.py:9-28, is
N512 C16 64x64 cut to 62x62), asymmetric and one-sided cuts, windows one element wide, and Shift1d rows shorter and longer than
one workgroup pass.  Bars as everywhere: fp32 / fp64 grad_x bit-exact, grad_w <= 1e-5 / 1e-12 of the fp64
evaluation; 16-bit: the sparse shift bit-exact, interpolation within 1 ulp, grad_w within half a unit of the type."""
import numpy as np
import pytest
import torch

from cases import rel_err, gw16_tol
from oracle import oracle as O
from test_hip_parity import _ulp_close, _weights
from test_flat_gpu import flat_serves

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture()
def abi():
    from torchshifts import abi as A
    assert torch.cuda.is_available()
    A.set_path_policy(0)
    yield A
    for k in (32, 33, 34, 35):
        A.set_tuning(k, 0)


CASES_2D = [((3, 5, 9, 24), [[1, 1], [1, 1]]), ((2, 3, 64, 64), [[1, 1], [1, 1]]), ((2, 2, 40, 224), [[1, 1], [1, 1]]),
            ((2, 3, 17, 32), [[2, 0], [0, 3]]), ((1, 2, 12, 16), [[0, 0], [5, 6]]), ((2, 2, 7, 1000), [[3, 3], [1, 0]]),
            ((1, 3, 33, 8), [[1, 2], [4, 3]]), ((2, 2, 6, 2048), [[0, 1], [7, 9]]), ((1, 2, 5, 12), [[4, 0], [0, 0]]),
            ((1, 1, 300, 16), [[100, 150], [2, 2]]), ((2, 2, 16, 16), [[0, 0], [0, 1]])]


@pytest.mark.parametrize("dt", ["f32", "f64", "f16", "bf16"])
@pytest.mark.parametrize("shape,crop", CASES_2D)
def test_cropped_backward_vs_oracle(abi, shape, crop, dt):
    tdt = {"f32": torch.float32, "f64": torch.float64, "f16": torch.float16, "bf16": torch.bfloat16}[dt]
    es = torch.empty(0, dtype=tdt).element_size()
    b, new = abi.check_borders(list(shape), crop, 2)
    total = 1
    for v in new:
        total *= v
    # x rows that are not whole 16-byte pieces, or a gradient that is not a whole number of pieces: the flat-stream kernels (round 5)
    ragged = bool((shape[-1] * es) % 16)   # (a gradient that is not a whole number of pieces is crop_backward's too since round 5)
    rs = np.random.RandomState(sum(shape) * 11 + 5)
    xt = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt)
    gt = torch.from_numpy(rs.uniform(-1, 1, size=new)).to(tdt)
    wt = torch.from_numpy(_weights(rs, shape[1], 2, shape[2:])).to(tdt)
    wide = np.float64 if tdt == torch.float64 else np.float32
    x, go, w = (t.to(torch.float64).numpy().astype(wide) for t in (xt, gt, wt))
    xd, god, wd = xt.to(DEV), gt.to(DEV), wt.to(DEV)
    for pad in range(5):
        for active in (0, 1):
            gx, gw = abi.backward(god, wd, xd, pad, active, b)
            # crop_backward: rows of at most 256 chunks (wider cropped rows keep the per-channel kernels; checked all the same)
            # (... and a window ONE column wide with zeros padding: its gradient map ignores the shift and is not the affine column
            #  state crop_backward<.., PAD = 0> reads through)
            lean = shape[-1] * es <= 4064 and not (pad == 0 and new[-1] == 1)
            if ragged:
                assert abi.last_kernel() in ("flat_backward", "crop_backward_ragged", "plane_backward", "small_plane_backward", "band_plane_backward"), (shape, crop, pad, abi.last_kernel())
            elif lean:
                assert abi.last_kernel() == "crop_backward", (shape, crop, pad, abi.last_kernel())
            else:   # (what crop_backward leaves: the flat-stream kernels when they serve the geometry, else the per-channel ones)
                assert abi.last_kernel() == ("flat_backward" if flat_serves(shape, new, es, True) else "plane_backward"), (shape, crop, pad, abi.last_kernel())
            gx_ref = torch.from_numpy(O.backward(go, w, x, pad, active, b)[0]).to(tdt)
            if es >= 4 or not active:
                assert torch.equal(gx.cpu(), gx_ref), ("gx", shape, crop, dt, pad, active)
            else:
                assert _ulp_close(gx.cpu(), gx_ref, tdt), ("gx", shape, crop, dt, pad, active)
            _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active, b)
            tol = {"f32": 1e-5, "f64": 1e-12}.get(dt, gw16_tol(torch.finfo(tdt).eps))
            if dt == "f32":
                # a window of a few hundred elements can cancel to a small sum (case 9: 600 terms of ~0.3 add up to 0.14): there the
                # reference's own fp32 CPU evaluation is 1.2e-5 .. 2e-5 from the fp64 one.  The bar stays 1e-5, or twice the fp32
                # oracle's own distance where that is larger -- never looser than the reference's fp32 path on the same data.
                own = 2 * rel_err(O.backward(go, w, x, pad, active, b)[1], gw64)
                if own > tol:   # (logged: `pytest -rP` / -s shows which cases used the wider bar)
                    print("relaxed grad_w bar", shape, crop, pad, active, "%.3g" % own)
                    tol = own
            assert rel_err(gw.to(torch.float64).cpu().numpy(), gw64) < tol, ("gw", shape, crop, dt, pad, active)
            gx2, gw2 = abi.backward(god, wd, xd, pad, active, b)
            assert torch.equal(gx, gx2) and torch.equal(gw, gw2)  # deterministic


CASES_1D = [((3, 5, 512), None), ((4, 3, 64), None), ((2, 2, 8), None), ((2, 3, 96), [[8, 24]]), ((2, 3, 4096), None), ((2, 2, 1024 + 256), None), ((2, 3, 2048), [[3, 5]]), ((1, 2, 4096 + 64), [[0, 16]]),
            ((2, 2, 640), [[100, 28]]), ((1, 3, 1032), None),
            # round 6 -- output rows that are not whole 16-byte pieces (Shift1d behind emulate_dw with padding 0: cut 1 / 1): the element-
            # aligned stores of row_forward; an odd width (16-bit: not served, the older kernels), a one-sided cut
            ((2, 3, 4096), [[1, 1]]), ((1, 2, 1032), [[0, 3]]), ((2, 2, 2048), [[2, 0]])]


@pytest.mark.parametrize("dt", ["f32", "f64", "f16", "bf16"])
@pytest.mark.parametrize("shape,crop", CASES_1D)
def test_1d_backward_vs_oracle(abi, shape, crop, dt):
    tdt = {"f32": torch.float32, "f64": torch.float64, "f16": torch.float16, "bf16": torch.bfloat16}[dt]
    es = torch.empty(0, dtype=tdt).element_size()
    b, new = abi.check_borders(list(shape), crop, 1)
    total = 1
    for v in new:
        total *= v
    abi.set_tuning(32, 2)   # whenever eligible (the automatic choice leaves rows of fewer than 128 chunks to the per-channel kernels)
    rs = np.random.RandomState(sum(shape) * 3 + 2)
    xt = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt)
    gt = torch.from_numpy(rs.uniform(-1, 1, size=new)).to(tdt)
    wt = torch.from_numpy(_weights(rs, shape[1], 1, shape[2:])).to(tdt)
    wide = np.float64 if tdt == torch.float64 else np.float32
    x, go, w = (t.to(torch.float64).numpy().astype(wide) for t in (xt, gt, wt))
    xd, god, wd = xt.to(DEV), gt.to(DEV), wt.to(DEV)
    for pad in range(5):
        for active in (0, 1):
            gx, gw = abi.backward(god, wd, xd, pad, active, b)
            assert abi.last_kernel() in ("row_backward", "crop_backward", "flat_backward", "plane_backward"), (shape, crop, abi.last_kernel())
            gx_ref = torch.from_numpy(O.backward(go, w, x, pad, active, b)[0]).to(tdt)
            if es >= 4 or not active:
                assert torch.equal(gx.cpu(), gx_ref), ("gx", shape, crop, dt, pad, active)
            else:
                assert _ulp_close(gx.cpu(), gx_ref, tdt), ("gx", shape, crop, dt, pad, active)
            _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active, b)
            tol = {"f32": 1e-5, "f64": 1e-12}.get(dt, gw16_tol(torch.finfo(tdt).eps))
            assert rel_err(gw.to(torch.float64).cpu().numpy(), gw64) < tol, ("gw", shape, crop, dt, pad, active)


CASES_FWD_2D = CASES_2D + [((2, 3, 62, 62), None), ((2, 2, 62, 62), [[1, 1], [1, 1]]), ((1, 2, 113, 113), [[0, 1], [0, 1]]), ((2, 2, 30, 20), None), ((2, 5, 8, 1), None), ((2, 3, 16, 2), None), ((1, 4, 12, 3), None), ((2, 2, 40, 8), [[1, 1], [3, 4]]),
                           ((1, 2, 9, 4100), [[1, 0], [2, 2]]),
                           # tall, narrow windows: a step of 256 chunks crosses more than 64 output-row boundaries (the
                           # straddling chunks beyond the 64th boundary were never written)
                           ((1, 2, 130, 12), [[1, 1], [1, 1]]), ((1, 2, 200, 24), [[0, 0], [1, 1]]), ((2, 2, 260, 8), [[2, 2], [1, 1]]),
                           ((1, 3, 400, 4), [[0, 0], [1, 0]]), ((1, 2, 514, 8), [[1, 1], [0, 1]])]


@pytest.mark.parametrize("dt", ["f32", "f64", "f16", "bf16"])
@pytest.mark.parametrize("shape,crop", CASES_FWD_2D)
def test_cropped_forward_vs_oracle(abi, shape, crop, dt):
    """crop_forward (cropped windows on source rows of whole pieces: ragged output rows, chunks that straddle two output rows, steps
    that straddle many) and -- ragged source rows (62 x 62, 113 x 113 inputs), planes that are not whole pieces -- the flat-stream
    kernels of round 5; every padding, both shifts"""
    tdt = {"f32": torch.float32, "f64": torch.float64, "f16": torch.float16, "bf16": torch.bfloat16}[dt]
    es = torch.empty(0, dtype=tdt).element_size()
    b, new = abi.check_borders(list(shape), crop, 2)
    xtotal = 1
    for v in shape:
        xtotal *= v
    rs = np.random.RandomState(sum(shape) * 17 + 3)
    xt = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt)
    wt = torch.from_numpy(_weights(rs, shape[1], 2, shape[2:])).to(tdt)
    wide = np.float64 if tdt == torch.float64 else np.float32
    x, w = (t.to(torch.float64).numpy().astype(wide) for t in (xt, wt))
    xd, wd = xt.to(DEV), wt.to(DEV)
    abi.set_tuning(34, 2)   # whenever eligible (the automatic choice leaves aligned, uncropped planes to the step kernels)
    abi.set_tuning(33, 1)
    for pad in range(5):
        for active in (0, 1):
            out = abi.forward(xd, wd, pad, active, b)
            # (aligned rows / tiny planes keep their kernels, windows far smaller than their planes the strided ones; checked all the same)
            if not abi.last_kernel().startswith(("step_", "plane_", "sweep_", "small_", "band_")) and flat_serves(shape, new, es, False):
                assert abi.last_kernel() in (("flat_active_forward", "crop_active_forward", "row_active_forward", "ragged_active_forward", "crop_active_forward_rows") if active else ("flat_gather_forward", "crop_gather_forward", "row_gather_forward", "ragged_gather_forward", "crop_gather_forward_rows")), (shape, crop, abi.last_kernel())
            ref = torch.from_numpy(O.forward(x, w, pad, active, b)).to(tdt)
            if es >= 4 or not active:
                assert torch.equal(out.cpu(), ref), (shape, crop, dt, pad, active)
            else:
                assert _ulp_close(out.cpu(), ref, tdt), (shape, crop, dt, pad, active)


@pytest.mark.parametrize("dt", ["f32", "f64", "f16", "bf16"])
@pytest.mark.parametrize("shape,crop", CASES_1D)
def test_1d_forward_vs_oracle(abi, shape, crop, dt):
    tdt = {"f32": torch.float32, "f64": torch.float64, "f16": torch.float16, "bf16": torch.bfloat16}[dt]
    es = torch.empty(0, dtype=tdt).element_size()
    b, new = abi.check_borders(list(shape), crop, 1)
    xtotal = 1
    for v in shape:
        xtotal *= v
    rs = np.random.RandomState(sum(shape) * 19 + 7)
    xt = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt)
    wt = torch.from_numpy(_weights(rs, shape[1], 1, shape[2:])).to(tdt)
    wide = np.float64 if tdt == torch.float64 else np.float32
    x, w = (t.to(torch.float64).numpy().astype(wide) for t in (xt, wt))
    xd, wd = xt.to(DEV), wt.to(DEV)
    abi.set_tuning(34, 2)   # whenever eligible (the automatic choice keeps 1-D forwards on the per-channel kernels)
    abi.set_tuning(33, 1)
    for pad in range(5):
        for active in (0, 1):
            out = abi.forward(xd, wd, pad, active, b)
            if not abi.last_kernel().startswith(("step_", "plane_", "sweep_")):
                assert abi.last_kernel() in (("flat_active_forward", "crop_active_forward", "row_active_forward", "ragged_active_forward", "crop_active_forward_rows") if active else ("flat_gather_forward", "crop_gather_forward", "row_gather_forward", "ragged_gather_forward", "crop_gather_forward_rows")), (shape, crop, abi.last_kernel())
            ref = torch.from_numpy(O.forward(x, w, pad, active, b)).to(tdt)
            if es >= 4 or not active:
                assert torch.equal(out.cpu(), ref), (shape, crop, dt, pad, active)
            else:
                assert _ulp_close(out.cpu(), ref, tdt), (shape, crop, dt, pad, active)


RAGGED_2D = [((1, 2, 62, 62), None), ((2, 3, 40, 33), None), ((1, 1, 225, 225), None), ((2, 2, 222, 222), None), ((1, 2, 113, 113), None),
             ((1, 3, 62, 62), [[1, 1], [1, 1]]), ((1, 2, 64, 70), [[1, 2], [3, 0]]), ((2, 2, 17, 1001), None), ((1, 2, 300, 35), [[100, 150], [2, 2]]),
             ((3, 2, 16, 33), None), ((1, 2, 50, 62), [[0, 0], [0, 61]]), ((2, 3, 30, 126), None), ((1, 2, 224, 222), [[1, 1], [0, 0]]),
             ((1, 4, 40, 70), [[0, 0], [1, 1]])]


@pytest.mark.parametrize("dt", ["f32", "f64", "f16", "bf16"])
@pytest.mark.parametrize("shape,crop", RAGGED_2D)
def test_ragged_rows_backward_vs_oracle(abi, shape, crop, dt):
    """crop_backward<.., XRAG> (round 5): x rows that are not whole 16-byte pieces -- 62 x 62 / 222 x 222 fp32, the output of the
    reference's cropped shift as the next layer's input (modules/shifts.py:41-46) -- in the row-relative form: covers with a phase
    for the x rows too, element-aligned 16-byte stores, a partial last chunk per row; with and without a window, every padding,
    both shifts.  grad_x bit-exact, grad_w <= 1e-5 / 1e-12 of the fp64 evaluation, deterministic."""
    tdt = {"f32": torch.float32, "f64": torch.float64, "f16": torch.float16, "bf16": torch.bfloat16}[dt]
    es = torch.empty(0, dtype=tdt).element_size()
    b, new = abi.check_borders(list(shape), crop, 2)
    rs = np.random.RandomState(sum(shape) * 5 + 9)
    xt = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt)
    gt = torch.from_numpy(rs.uniform(-1, 1, size=new)).to(tdt)
    wt = torch.from_numpy(_weights(rs, shape[1], 2, shape[2:])).to(tdt)
    wide = np.float64 if tdt == torch.float64 else np.float32
    x, go, w = (t.to(torch.float64).numpy().astype(wide) for t in (xt, gt, wt))
    xd, god, wd = xt.to(DEV), gt.to(DEV), wt.to(DEV)
    ragged = (shape[-1] * es) % 16 != 0
    for pad in range(5):
        for active in (0, 1):
            gx, gw = abi.backward(god, wd, xd, pad, active, b)
            # (2-byte elements: rows of an even number of them, so that every row starts at a 4-byte boundary; the others: the flat stream)
            served = ragged and 128 <= shape[-1] * es <= 254 * 16 and shape[-2] >= 16 and (es >= 4 or (shape[-1] % 2 == 0 and new[-1] % 2 == 0))
            if served and not (pad == 0 and new[-1] == 1):
                assert abi.last_kernel() == "crop_backward_ragged", (shape, crop, dt, abi.last_kernel())
            gx_ref = torch.from_numpy(O.backward(go, w, x, pad, active, b)[0]).to(tdt)
            if es >= 4 or not active:
                assert torch.equal(gx.cpu(), gx_ref), ("gx", shape, crop, dt, pad, active)
            else:
                assert _ulp_close(gx.cpu(), gx_ref, tdt), ("gx", shape, crop, dt, pad, active)
            _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active, b)
            tol = {"f32": 1e-5, "f64": 1e-12}.get(dt, gw16_tol(torch.finfo(tdt).eps))
            if dt == "f32":
                own = 2 * rel_err(O.backward(go, w, x, pad, active, b)[1], gw64)
                if own > tol:   # (logged: `pytest -rP` / -s shows which cases used the wider bar)
                    print("relaxed grad_w bar", shape, crop, pad, active, "%.3g" % own)
                    tol = own
            assert rel_err(gw.to(torch.float64).cpu().numpy(), gw64) < tol, ("gw", shape, crop, dt, pad, active)
            gx2, gw2 = abi.backward(god, wd, xd, pad, active, b)
            assert torch.equal(gx, gx2) and torch.equal(gw, gw2)  # deterministic


@pytest.mark.parametrize("dt", ["f32", "f64", "f16", "bf16"])
@pytest.mark.parametrize("shape,crop", RAGGED_2D)
def test_ragged_rows_forward_vs_oracle(abi, shape, crop, dt):
    """ragged_forward (round 5): the forward twin of crop_backward<.., XRAG> -- source rows that are not whole 16-byte pieces,
    row-relative output chunks, element-aligned stores; bit-exact"""
    tdt = {"f32": torch.float32, "f64": torch.float64, "f16": torch.float16, "bf16": torch.bfloat16}[dt]
    es = torch.empty(0, dtype=tdt).element_size()
    b, new = abi.check_borders(list(shape), crop, 2)
    rs = np.random.RandomState(sum(shape) * 3 + 4)
    xt = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt)
    wt = torch.from_numpy(_weights(rs, shape[1], 2, shape[2:])).to(tdt)
    wide = np.float64 if tdt == torch.float64 else np.float32
    x, w = (t.to(torch.float64).numpy().astype(wide) for t in (xt, wt))
    xd, wd = xt.to(DEV), wt.to(DEV)
    served = ((shape[-1] * es) % 16 != 0 and 128 <= shape[-1] * es <= 254 * 16 and shape[-2] >= 16 and
              (es >= 4 or (shape[-1] % 2 == 0 and new[-1] % 2 == 0)))
    for pad in range(5):
        for active in (0, 1):
            out = abi.forward(xd, wd, pad, active, b)
            if served:
                assert abi.last_kernel() == ("ragged_active_forward" if active else "ragged_gather_forward"), (shape, crop, dt, abi.last_kernel())
            ref = torch.from_numpy(O.forward(x, w, pad, active, b)).to(tdt)
            if es >= 4 or not active:
                assert torch.equal(out.cpu(), ref), (shape, crop, dt, pad, active)
            else:
                assert _ulp_close(out.cpu(), ref, tdt), (shape, crop, dt, pad, active)


# round 6 -- crop_backward3: cropped 3-D volumes (Shift3d behind emulate_dw with padding < kernel / 2: cut 1 / 1 per dim), x rows of
# whole 16-byte pieces: symmetric and one-sided cuts, a cut in the planes only / in the rows only, several steps per plane, the
# ragged last step, shifts beyond the volume (the weights of _weights), windows of two planes
CASES_3D = [((2, 3, 5, 6, 8), [[1, 1], [1, 1], [1, 1]]), ((1, 4, 4, 9, 16), [[0, 1], [2, 0], [1, 2]]), ((2, 2, 6, 7, 32), [[2, 2], [0, 0], [0, 0]]),
            ((1, 2, 3, 70, 16), [[0, 0], [1, 1], [0, 0]]), ((1, 3, 8, 5, 24), [[3, 3], [1, 2], [5, 6]]), ((2, 2, 4, 12, 64), [[1, 0], [0, 3], [7, 9]]),
            ((1, 2, 16, 20, 112), [[1, 1], [1, 1], [1, 1]]), ((1, 2, 4, 6, 16), [[0, 1], [1, 0], [2, 2]]), ((2, 2, 3, 40, 32), [[0, 0], [0, 0], [2, 0]]),
            ((1, 2, 4, 5, 16), [[0, 0], [1, 0], [3, 3]]), ((2, 2, 5, 33, 24), [[1, 1], [0, 0], [1, 5]])]   # (10- / 18-column windows: a last piece of one dword)


@pytest.mark.parametrize("dt", ["f32", "f64", "f16", "bf16"])
@pytest.mark.parametrize("shape,crop", CASES_3D)
def test_cropped_3d_backward_vs_oracle(abi, shape, crop, dt):
    tdt = {"f32": torch.float32, "f64": torch.float64, "f16": torch.float16, "bf16": torch.bfloat16}[dt]
    es = torch.empty(0, dtype=tdt).element_size()
    b, new = abi.check_borders(list(shape), crop, 3)
    rs = np.random.RandomState(sum(shape) * 17 + 1)
    xt = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt)
    gt = torch.from_numpy(rs.uniform(-1, 1, size=new)).to(tdt)
    wt = torch.from_numpy(_weights(rs, shape[1], 3, shape[2:])).to(tdt)
    wide = np.float64 if tdt == torch.float64 else np.float32
    x, go, w = (t.to(torch.float64).numpy().astype(wide) for t in (xt, gt, wt))
    xd, god, wd = xt.to(DEV), gt.to(DEV), wt.to(DEV)
    served = (shape[-1] * es) % 16 == 0 and min(shape[2:]) >= 2 and min(new[2:]) >= 2
    # 16-bit tensors under zeros padding: the walk through the planes with the window inside (walk_backward16<.., CROP>) -- window rows of
    # an even number of elements that begin at most two columns into the volume's rows
    # ... and 4-byte elements (walk_backward<.., CROP>, shiftnd_walk3.hip: the reference's blend order, bit for bit)
    walk = served and crop[2][0] <= 2 and ((es == 2 and new[-1] % 2 == 0) or es == 4)
    wname = "walk_backward16_crop" if es == 2 else "walk_backward_crop"
    for pad in range(5):
        for active in (0, 1):
            gx, gw = abi.backward(god, wd, xd, pad, active, b)
            if walk and pad == 0:
                assert abi.last_kernel() == (wname if active else wname + "_sparse"), (shape, crop, pad, abi.last_kernel())
                abi.set_tuning(35, 2048)   # (knob 35 bit 11: the one-step kernel it replaces -- the same bits, but for 16-bit interpolation)
                gx1, _ = abi.backward(god, wd, xd, pad, active, b)
                abi.set_tuning(35, 0)
                assert abi.last_kernel() == "crop_backward3" and ((active and es == 2) or torch.equal(gx1, gx)), (shape, crop, dt, active)
            elif served:
                assert abi.last_kernel() == "crop_backward3", (shape, crop, pad, abi.last_kernel())
            gx_ref = torch.from_numpy(O.backward(go, w, x, pad, active, b)[0]).to(tdt)
            if es >= 4 or not active:
                assert torch.equal(gx.cpu(), gx_ref), ("gx", shape, crop, dt, pad, active)
            else:
                assert _ulp_close(gx.cpu(), gx_ref, tdt), ("gx", shape, crop, dt, pad, active)
            _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active, b)
            tol = {"f32": 1e-5, "f64": 1e-12}.get(dt, gw16_tol(torch.finfo(tdt).eps))
            if dt == "f32":   # (as above: never looser than the reference's own fp32 evaluation on the same data)
                own = 2 * rel_err(O.backward(go, w, x, pad, active, b)[1], gw64)
                if own > tol:
                    print("relaxed grad_w bar", shape, crop, pad, active, "%.3g" % own)
                    tol = own
            assert rel_err(gw.to(torch.float64).cpu().numpy(), gw64) < tol, ("gw", shape, crop, dt, pad, active)
            gx2, gw2 = abi.backward(god, wd, xd, pad, active, b)
            assert torch.equal(gx, gx2) and torch.equal(gw, gw2)  # deterministic
            # the plane kernels it replaces give the same grad_x (knob 35 bit 10)
            abi.set_tuning(35, 1024 + 2048)
            gx3, _ = abi.backward(god, wd, xd, pad, active, b)
            abi.set_tuning(35, 0)
            assert "crop" not in abi.last_kernel(), abi.last_kernel()
            assert torch.equal(gx3, gx) or (walk and es == 2 and pad == 0 and active and _ulp_close(gx3.cpu(), gx.cpu(), tdt)), (shape, crop, dt, pad, active)


@pytest.mark.parametrize("dt", ["f32", "f64", "f16", "bf16"])
@pytest.mark.parametrize("shape,crop", CASES_3D)
def test_cropped_3d_forward_vs_oracle(abi, shape, crop, dt):
    """crop_forward3 (round 6): the forward of the same cropped volumes, both shifts, every padding, against the oracle"""
    tdt = {"f32": torch.float32, "f64": torch.float64, "f16": torch.float16, "bf16": torch.bfloat16}[dt]
    es = torch.empty(0, dtype=tdt).element_size()
    b, new = abi.check_borders(list(shape), crop, 3)
    rs = np.random.RandomState(sum(shape) * 19 + 7)
    xt = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt)
    wt = torch.from_numpy(_weights(rs, shape[1], 3, shape[2:])).to(tdt)
    wide = np.float64 if tdt == torch.float64 else np.float32
    x, w = (t.to(torch.float64).numpy().astype(wide) for t in (xt, wt))
    xd, wd = xt.to(DEV), wt.to(DEV)
    # served: source rows of whole pieces, dims of at least 2, 16-bit windows of an even width; output rows that ARE whole pieces at
    # the sources' own columns may go to the aligned one-step forwards instead (asked first)
    served = (shape[-1] * es) % 16 == 0 and min(shape[2:]) >= 2 and min(new[2:]) >= 2 and (es != 2 or new[-1] % 2 == 0)
    hit = 0
    for pad in range(5):
        for active in (0, 1):
            out = abi.forward(xd, wd, pad, active, b)
            name = abi.last_kernel()
            # (16-bit, zeros padding, interpolating: the walk through the planes with the window inside, walk_forward16<.., CROP>)
            fwalk = served and es == 2 and pad == 0 and new[-1] % 2 == 0   # (both shifts: the sparse one is the same walk without blends)
            if fwalk:
                assert name == ("walk_forward16_crop" if active else "walk_forward16_crop_sparse"), (shape, crop, pad, active, name)
            elif served and (new[-1] * es) % 16 != 0:
                assert name == ("crop_active_forward3" if active else "crop_gather_forward3"), (shape, crop, pad, active, name)
            hit += name.endswith("forward3") or name.startswith("walk_forward16_crop")
            ref = torch.from_numpy(O.forward(x, w, pad, active, b)).to(tdt)
            if es >= 4 or not active:
                assert torch.equal(out.cpu(), ref), (shape, crop, dt, pad, active, name)
            else:
                assert _ulp_close(out.cpu(), ref, tdt), (shape, crop, dt, pad, active, name)
            abi.set_tuning(35, 1024 + 2048)   # the kernels it replaces give the same bits
            out2 = abi.forward(xd, wd, pad, active, b)
            abi.set_tuning(35, 0)
            assert not abi.last_kernel().endswith("forward3") and not abi.last_kernel().startswith("walk_forward16_crop")
            if es >= 4 or not active:
                assert torch.equal(out2, out), (shape, crop, dt, pad, active)
    if served and (new[-1] * es) % 16 != 0:
        assert hit == 10


# round 6 -- crop_forward3<.., ND = 2>: cropped 2-D windows whose output planes are not whole 16-byte pieces (110 x 110 bf16: the
# mixed-precision network behind emulate_dw with padding 0), row-relative; fp32 windows of an odd area take it too
ROWS_2D = [((2, 3, 112, 112), [[1, 1], [1, 1]]), ((1, 4, 20, 32), [[1, 1], [1, 1]]), ((2, 2, 9, 24), [[0, 2], [3, 1]]), ((1, 2, 33, 64), [[1, 1], [0, 2]]),
           ((1, 3, 12, 16), [[2, 3], [1, 0]]), ((2, 2, 7, 256), [[1, 1], [1, 1]])]


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("shape,crop", ROWS_2D)
def test_cropped_rows_forward_vs_oracle(abi, shape, crop, dt):
    tdt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[dt]
    es = torch.empty(0, dtype=tdt).element_size()
    b, new = abi.check_borders(list(shape), crop, 2)
    rs = np.random.RandomState(sum(shape) * 23 + 3)
    xt = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt)
    wt = torch.from_numpy(_weights(rs, shape[1], 2, shape[2:])).to(tdt)
    x, w = (t.to(torch.float64).numpy().astype(np.float32) for t in (xt, wt))
    xd, wd = xt.to(DEV), wt.to(DEV)
    xcpr = shape[-1] * es // 16
    fill = min(new[-2], 256 // xcpr) * -(-new[-1] * es // 16)   # threads of a step that have a chunk: at least half a workgroup
    rows = ((new[-2] * new[-1] * es) % 16 != 0 and min(new[2:]) >= 2 and (es != 2 or new[-1] % 2 == 0) and fill >= 128)
    for pad in range(5):
        for active in (0, 1):
            out = abi.forward(xd, wd, pad, active, b)
            if rows:
                assert abi.last_kernel() == ("crop_active_forward_rows" if active else "crop_gather_forward_rows"), (shape, crop, dt, abi.last_kernel())
            ref = torch.from_numpy(O.forward(x, w, pad, active, b)).to(tdt)
            if es >= 4 or not active:
                assert torch.equal(out.cpu(), ref), (shape, crop, dt, pad, active, abi.last_kernel())
            else:
                assert _ulp_close(out.cpu(), ref, tdt), (shape, crop, dt, pad, active, abi.last_kernel())


def _guarded(shape, tdt, fill=0.0):
    """a contiguous tensor of `shape` inside a larger buffer, 16-byte aligned, with 512 sentinel bytes on either side"""
    es = torch.empty(0, dtype=tdt).element_size()
    n = int(np.prod(shape))
    pad = 512 // es
    big = torch.full((n + 2 * pad,), 7.0, dtype=tdt, device=DEV)
    view = big[pad:pad + n].view(shape)
    view.fill_(fill)
    assert view.data_ptr() % 16 == 0
    return big, view, pad


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("shape,crop", [((2, 3, 5, 6, 16), [[1, 1], [1, 1], [1, 1]]), ((1, 2, 4, 9, 24), [[0, 1], [2, 0], [1, 5]]),
                                        ((1, 2, 16, 20, 112), [[1, 1], [1, 1], [1, 1]]), ((2, 2, 3, 7, 8), [[0, 1], [1, 1], [0, 2]])])
def test_cropped_3d_kernels_stay_inside_their_tensors(abi, shape, crop, dt):
    """the cropped walks store partial pieces at the ends of window rows (forward) and mask grad_x (backward): nothing may be written
    outside the output tensors -- sentinel bytes around `out` and `grad_x` survive every padding and both shifts"""
    tdt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[dt]
    b, new = abi.check_borders(list(shape), crop, 3)
    rs = np.random.RandomState(sum(shape) + 3)
    xd = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt).to(DEV)
    gd = torch.from_numpy(rs.uniform(-1, 1, size=new)).to(tdt).to(DEV)
    wd = torch.from_numpy(_weights(rs, shape[1], 3, shape[2:])).to(tdt).to(DEV)
    for pad in range(5):
        for active in (0, 1):
            big_o, out, po = _guarded(new, tdt)
            abi.forward(xd, wd, pad, active, b, out=out)
            torch.cuda.synchronize()
            assert bool((big_o[:po] == 7).all()) and bool((big_o[po + out.numel():] == 7).all()), ("fwd", shape, crop, dt, pad, active, abi.last_kernel())
            assert torch.equal(out, abi.forward(xd, wd, pad, active, b)), ("fwd", shape, crop, dt, pad, active)
            big_g, gx, pg = _guarded(shape, tdt)
            gx1, gw1 = abi.backward(gd, wd, xd, pad, active, b, grad_x=gx)
            torch.cuda.synchronize()
            assert bool((big_g[:pg] == 7).all()) and bool((big_g[pg + gx.numel():] == 7).all()), ("bwd", shape, crop, dt, pad, active, abi.last_kernel())
            gx2, gw2 = abi.backward(gd, wd, xd, pad, active, b)
            assert torch.equal(gx1, gx2) and torch.equal(gw1, gw2)
