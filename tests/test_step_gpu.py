"""GPU parity of the one-step kernels (csrc/shiftnd_step.hip: the backward pass of contiguous 2-D problems as a linear
sweep of one-step workgroups) against the CPU oracle, through the C ABI.

Bars as everywhere: fp32 / fp64 grad_x bit-exact, grad_w <= 1e-5 relative to the fp64 oracle
(bit-exact on the dyadic golden fixture); 16-bit: the sparse shift bit-exact, interpolation within 1 ulp of the 16-bit
type, grad_w within the 16-bit epsilon.  Shapes cover one step per plane, ragged last steps, one-row planes, rows of one
chunk and of 256 chunks, shifts beyond the dim (the special rows of `_weights`)."""
import numpy as np
import pytest
import torch

from cases import rel_err, gw16_tol
from oracle import oracle as O
from test_hip_parity import _ulp_close, _weights

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# 16-bit 3-D interpolation (the walk kernels): 1 ulp of the 16-bit type at the result, plus 8 fp32 ulps of the OPERANDS' scale
# (|x|, |grad| <= 1 in these tests) for results that cancel to almost nothing.  Seven nested blends, evaluated here inner-first
# with mul + fma and by the oracle plane-first with mul, mul, add (the reference's nesting): each blend rounds once at the
# magnitude of its operands, so the two fp32 evaluations differ by a few 2^-24 of the operand scale whatever the result is --
# more than one 16-bit ulp of a result below 2^-16 of that scale.  Among millions of outputs a few are that small.
FLOOR16 = 8 * 2.0 ** -24


@pytest.fixture()
def abi():
    from torchshifts import abi as A
    assert torch.cuda.is_available()
    A.set_path_policy(0)
    A.set_tuning(32, 2)  # whenever eligible
    A.set_tuning(33, 2)
    A.set_tuning(34, 2)
    yield A
    for k in (32, 33, 34, 35):
        A.set_tuning(k, 0)


SHAPES = [(3, 5, 9, 24), (2, 3, 40, 224), (5, 2, 33, 64), (1, 2, 300, 8), (2, 3, 1, 16), (7, 2, 6, 56), (1, 2, 7, 1000),
          (2, 1, 5, 1024), (2, 4, 224, 224), (1, 3, 17, 4), (2, 2, 64, 12), (3, 2, 2, 512)]


@pytest.mark.parametrize("shape", SHAPES)
def test_fp32_vs_oracle(abi, shape):
    rs = np.random.RandomState(sum(shape) * 13 + 1)
    x = rs.uniform(-1, 1, size=shape).astype(np.float32)
    go = rs.uniform(-1, 1, size=shape).astype(np.float32)
    w = _weights(rs, shape[1], 2, shape[2:]).astype(np.float32)
    xd, wd, god = (torch.from_numpy(a).to(DEV) for a in (x, w, go))
    abi.set_tuning(35, 16)  # (not the walk through the planes: test_3d_walk_backward_vs_oracle)
    for pad in range(5):
        for active in (0, 1):
            gx, gw = abi.backward(god, wd, xd, pad, active)
            assert abi.last_kernel() == "step_backward", (shape, abi.last_kernel())
            gx_o, _ = O.backward(go, w, x, pad, active)
            assert np.array_equal(gx.cpu().numpy(), gx_o), ("gx", shape, pad, active)
            _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active)
            assert rel_err(gw.cpu().numpy(), gw64) < 1e-5, ("gw", shape, pad, active)
            gx2, gw2 = abi.backward(god, wd, xd, pad, active)
            assert torch.equal(gx, gx2) and torch.equal(gw, gw2)  # deterministic


@pytest.mark.parametrize("shape", [(3, 5, 9, 24), (2, 3, 40, 224), (1, 2, 300, 8), (2, 3, 1, 16), (2, 1, 5, 512), (1, 2, 33, 2)])
def test_fp64_vs_oracle(abi, shape):
    rs = np.random.RandomState(sum(shape) * 7 + 3)
    x = rs.uniform(-1, 1, size=shape)
    go = rs.uniform(-1, 1, size=shape)
    w = _weights(rs, shape[1], 2, shape[2:]).astype(np.float64)
    xd, wd, god = (torch.from_numpy(a).to(DEV) for a in (x, w, go))
    abi.set_tuning(35, 16)  # (not the walk through the planes: test_3d_walk_backward_vs_oracle)
    for pad in range(5):
        for active in (0, 1):
            gx, gw = abi.backward(god, wd, xd, pad, active)
            assert abi.last_kernel() == "step_backward", (shape, abi.last_kernel())
            gx_o, gw_o = O.backward(go, w, x, pad, active)
            assert np.array_equal(gx.cpu().numpy(), gx_o), ("gx", shape, pad, active)
            assert rel_err(gw.cpu().numpy(), gw_o) < 1e-12, ("gw", shape, pad, active)


@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(3, 5, 9, 24), (2, 3, 40, 224), (5, 2, 33, 64), (1, 2, 300, 8), (2, 2, 1, 16), (2, 1, 5, 2048)])
def test_16bit_vs_oracle(abi, shape, tdt):
    rs = np.random.RandomState(sum(shape) + 11)
    x16 = torch.from_numpy(rs.uniform(-1, 1, size=shape).astype(np.float32)).to(tdt)
    go16 = torch.from_numpy(rs.uniform(-1, 1, size=shape).astype(np.float32)).to(tdt)
    w16 = torch.from_numpy(_weights(rs, shape[1], 2, shape[2:]).astype(np.float32)).to(tdt)
    x, w, go = x16.float().numpy(), w16.float().numpy(), go16.float().numpy()
    xd, wd, god = x16.to(DEV), w16.to(DEV), go16.to(DEV)
    abi.set_tuning(35, 16)  # (not the walk through the planes)
    for pad in range(5):
        for active in (0, 1):
            gx, gw = abi.backward(god, wd, xd, pad, active)
            assert abi.last_kernel() == "step_backward"
            gx_ref = torch.from_numpy(O.backward(go, w, x, pad, active)[0]).to(tdt)
            if active:
                assert _ulp_close(gx.cpu(), gx_ref, tdt), ("gx", shape, pad)
            else:
                assert torch.equal(gx.cpu(), gx_ref), ("gx", shape, pad)
            _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active)
            assert rel_err(gw.float().cpu().numpy(), gw64) < gw16_tol(torch.finfo(tdt).eps), ("gw", shape, pad, active)


def test_dyadic_bit_exact(abi):
    """dyadic data (multiples of 1/8, weights multiples of 1/4): every product and partial sum is exact in fp32, so grad_w
    must equal the fp64 oracle's bit for bit whatever the summation order (the golden fixture's rows are 5 wide, not whole
    16-byte pieces: tests/test_hip_parity.py runs it through the other families)"""
    rs = np.random.RandomState(7)
    for shape in ((2, 3, 8, 16), (1, 4, 13, 32), (3, 2, 5, 8)):
        x = (rs.randint(-8, 9, size=shape) / 8.0).astype(np.float32)
        go = (rs.randint(-8, 9, size=shape) / 8.0).astype(np.float32)
        w = (rs.randint(-14, 15, size=(shape[1], 2)) / 4.0).astype(np.float32)
        xd, wd, god = (torch.from_numpy(a).to(DEV) for a in (x, w, go))
        for pad in range(5):
            for active in (0, 1):
                gx, gw = abi.backward(god, wd, xd, pad, active)
                assert abi.last_kernel() == "step_backward"
                gx_o, _ = O.backward(go, w, x, pad, active)
                _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active)
                assert np.array_equal(gx.cpu().numpy(), gx_o), (shape, pad, active)
                assert np.array_equal(gw.cpu().numpy(), gw64.astype(np.float32)), (shape, pad, active)


def test_agrees_with_the_band_walk_kernels(abi):
    """same grad_x bits as plane_backward_lds, grad_w within the fp32 bar, on a shape both families take"""
    torch.manual_seed(3)
    for shape, tdt in (((4, 6, 48, 96), torch.float32), ((4, 6, 48, 96), torch.float16), ((2, 8, 224, 224), torch.float32)):
        x = torch.rand(shape, device=DEV).to(tdt)
        go = torch.rand(shape, device=DEV).to(tdt)
        w = ((torch.rand(shape[1], 2, device=DEV) - 0.5) * 7).to(tdt)
        for pad in range(5):
            for active in (0, 1):
                abi.set_tuning(32, 1)
                gx0, gw0 = abi.backward(go, w, x, pad, active)
                assert abi.last_kernel().startswith("plane_backward")
                abi.set_tuning(32, 2)
                gx1, gw1 = abi.backward(go, w, x, pad, active)
                assert abi.last_kernel() == "step_backward"
                if active and tdt != torch.float32:
                    assert _ulp_close(gx1.cpu(), gx0.cpu(), tdt)
                else:
                    assert torch.equal(gx0, gx1)
                assert rel_err(gw1.float().cpu().numpy(), gw0.float().cpu().numpy()) < (1e-5 if tdt == torch.float32 else 2e-2)


FWD_SHAPES = [((3, 5, 9, 24), None), ((2, 3, 40, 224), None), ((5, 2, 33, 64), [[1, 2], [4, 8]]), ((1, 2, 300, 8), [[3, 0], [0, 0]]),
              ((2, 3, 1, 16), None), ((2, 2, 7, 1000), [[0, 1], [8, 16]]), ((2, 1, 5, 1024), None), ((2, 4, 224, 224), None),
              ((1, 3, 17, 4), None), ((2, 3, 30, 40), [[2, 3], [4, 4]])]


@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("shape,crop", FWD_SHAPES)
def test_gather_forward_vs_oracle(abi, shape, crop, dt):
    """sparse-shift forward of 4- / 8-byte elements, crops included: bit-exact (a pure gather)"""
    rs = np.random.RandomState(sum(shape) * 5 + 2)
    npdt = np.float32 if dt == "f32" else np.float64
    x = rs.uniform(-1, 1, size=shape).astype(npdt)
    w = _weights(rs, shape[1], 2, shape[2:]).astype(npdt)
    b, new = abi.check_borders(list(shape), crop, 2)
    if (new[-1] * x.itemsize) % 16 or new[-1] * x.itemsize > 4096:
        pytest.skip("output rows are not whole 16-byte chunks, or wider than one workgroup pass")
    xd, wd = torch.from_numpy(x).to(DEV), torch.from_numpy(w).to(DEV)
    for pad in range(5):
        out = abi.forward(xd, wd, pad, 0, b)
        assert abi.last_kernel() == "step_gather_forward", (shape, abi.last_kernel())
        assert np.array_equal(out.cpu().numpy(), O.forward(x, w, pad, 0, b)), (shape, crop, pad)


def test_gather_forward_quantized_int32_and_huge_weights(abi):
    """qint32 tensors with quantized weights (the generic shift path), and float weights beyond 2^30 (the 64-bit path)"""
    rs = np.random.RandomState(5)
    xq = rs.randint(-1000, 1000, size=(2, 4, 24, 32)).astype(np.int32)
    wq = rs.randint(120, 136, size=(4, 2)).astype(np.uint8)
    for pad in range(5):
        out = abi.forward_quantized(torch.from_numpy(xq).to(DEV), torch.from_numpy(wq).to(DEV), 128, -3, pad)
        assert abi.last_kernel() == "step_gather_forward"
        assert np.array_equal(out.cpu().numpy(), O.forward_q(xq, wq, 128, -3, pad)), pad
    x = rs.uniform(-1, 1, size=(1, 3, 12, 16)).astype(np.float32)
    w = np.array([[3e9, -2.0], [1.0, -5e9], [2.5e9, 2.5e9]], dtype=np.float32)
    for pad in range(5):
        out = abi.forward(torch.from_numpy(x).to(DEV), torch.from_numpy(w).to(DEV), pad, 0)
        assert abi.last_kernel() == "step_gather_forward"
        assert np.array_equal(out.cpu().numpy(), O.forward(x, w, pad, 0)), pad


@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("shape,crop", FWD_SHAPES)
def test_active_forward_vs_oracle(abi, shape, crop, dt):
    """interpolating forward through LDS, crops included: fp32 / fp64 bit-exact (the reference's expression order)"""
    rs = np.random.RandomState(sum(shape) * 3 + 4)
    npdt = np.float32 if dt == "f32" else np.float64
    x = rs.uniform(-1, 1, size=shape).astype(npdt)
    w = _weights(rs, shape[1], 2, shape[2:]).astype(npdt)
    b, new = abi.check_borders(list(shape), crop, 2)
    if (new[-1] * x.itemsize) % 16 or (shape[-1] * x.itemsize) % 16 or new[-1] * x.itemsize > 4096:
        pytest.skip("rows are not whole 16-byte pieces, or wider than one workgroup pass")
    xd, wd = torch.from_numpy(x).to(DEV), torch.from_numpy(w).to(DEV)
    for pad in range(5):
        ref = O.forward(x, w, pad, 1, b)
        for groups in (2, 3):  # through LDS, one / two row groups per thread
            abi.set_tuning(34, groups)
            out = abi.forward(xd, wd, pad, 1, b)
            assert abi.last_kernel() == "step_active_forward", (shape, abi.last_kernel())
            assert np.array_equal(out.cpu().numpy(), ref), ("lds", shape, crop, pad, groups)
        abi.set_tuning(34, 2)


@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape,crop", [((3, 5, 9, 24), None), ((2, 3, 40, 224), None), ((5, 2, 33, 64), [[1, 2], [8, 8]]),
                                        ((1, 2, 300, 8), None), ((2, 2, 1, 16), None), ((2, 1, 5, 2048), None),
                                        ((2, 3, 30, 48), [[2, 3], [8, 16]])])
def test_16bit_forwards_vs_oracle(abi, shape, crop, tdt):
    """2-byte elements: the sparse shift is a raw copy (bit-exact), interpolation within 1 ulp of the 16-bit type"""
    rs = np.random.RandomState(sum(shape) + 29)
    x16 = torch.from_numpy(rs.uniform(-1, 1, size=shape).astype(np.float32)).to(tdt)
    w16 = torch.from_numpy(_weights(rs, shape[1], 2, shape[2:]).astype(np.float32)).to(tdt)
    x, w = x16.float().numpy(), w16.float().numpy()
    b, new = abi.check_borders(list(shape), crop, 2)
    xd, wd = x16.to(DEV), w16.to(DEV)
    for pad in range(5):
        for groups in (2, 3):  # one / two row groups per thread
            abi.set_tuning(34, groups)
            abi.set_tuning(33, 1)  # (step_gather_forward_small would take the sparse shift first: tested below)
            out = abi.forward(xd, wd, pad, 0, b)
            abi.set_tuning(33, 2)
            assert abi.last_kernel() == "step_gather_forward_lds", (shape, abi.last_kernel())
            assert torch.equal(out.cpu(), torch.from_numpy(O.forward(x, w, pad, 0, b)).to(tdt)), ("ssl", shape, pad, groups)
            out = abi.forward(xd, wd, pad, 1, b)
            assert abi.last_kernel() == "step_active_forward", (shape, abi.last_kernel())
            assert _ulp_close(out.cpu(), torch.from_numpy(O.forward(x, w, pad, 1, b)).to(tdt), tdt), ("active", shape, pad, groups)
        abi.set_tuning(34, 2)


SHAPES3 = [(2, 3, 5, 6, 16), (1, 2, 20, 9, 64), (2, 2, 3, 40, 112), (1, 3, 1, 5, 8), (1, 2, 6, 1, 32), (1, 1, 3, 37, 512),
           (2, 2, 4, 19, 4)]


@pytest.mark.parametrize("npdt", [np.uint8, np.int8, np.uint16])
@pytest.mark.parametrize("shape,crop", [((3, 5, 9, 32), None), ((2, 3, 40, 224), None), ((5, 2, 33, 64), [[1, 2], [16, 16]]),
                                        ((1, 2, 300, 16), None), ((2, 2, 1, 48), None), ((2, 1, 5, 2048), None),
                                        ((2, 3, 30, 96), [[2, 3], [16, 32]]), ((2, 2, 17, 80), [[0, 0], [5, 11]])])
def test_small_element_gather_forward_vs_oracle(abi, shape, crop, npdt):
    """1- and 2-byte elements through two aligned 16-byte loads and a uniform byte funnel: bit-exact (a pure gather),
    every padding, crops that keep or break the 16-byte phase, quantized weights (uint8 / int8 tensors) and float16 bit
    patterns (uint16 stands for fp16 / bf16: the kernel moves raw elements)"""
    rs = np.random.RandomState(sum(shape) * 7 + 9)
    b, new = abi.check_borders(list(shape), crop, 2)
    es = np.dtype(npdt).itemsize
    if (new[-1] * es) % 16 or (shape[-1] * es) % 16 or new[-1] * es > 4096:
        pytest.skip("rows are not whole 16-byte pieces, or wider than one workgroup pass")
    if npdt == np.uint16:
        x16 = torch.from_numpy(rs.uniform(-1, 1, size=shape).astype(np.float32)).to(torch.float16)
        w16 = torch.from_numpy(_weights(rs, shape[1], 2, shape[2:]).astype(np.float32)).to(torch.float16)
        for pad in range(5):
            out = abi.forward(x16.to(DEV), w16.to(DEV), pad, 0, b)
            assert abi.last_kernel() == "step_gather_forward_small", (shape, abi.last_kernel())
            ref = torch.from_numpy(O.forward(x16.float().numpy(), w16.float().numpy(), pad, 0, b)).to(torch.float16)
            assert torch.equal(out.cpu(), ref), (shape, crop, pad)
        return
    info = np.iinfo(npdt)
    xq = rs.randint(info.min, info.max + 1, size=shape).astype(npdt)
    wq = rs.randint(118, 139, size=(shape[1], 2)).astype(np.uint8)
    wq[0] = [128 + 100, 128 - 90]  # far beyond small dims
    for pad in range(5):
        out = abi.forward_quantized(torch.from_numpy(xq).to(DEV), torch.from_numpy(wq).to(DEV), 128, 7, pad, b)
        assert abi.last_kernel() == "step_gather_forward_small", (shape, abi.last_kernel())
        assert np.array_equal(out.cpu().numpy(), O.forward_q(xq, wq, 128, 7, pad, b)), (shape, crop, pad)


@pytest.mark.parametrize("dt", ["f32", "f64", "f16", "bf16"])
@pytest.mark.parametrize("shape,crop", [((2, 3, 5, 6, 16), None), ((1, 2, 20, 9, 64), None), ((2, 2, 3, 40, 112), None),
                                        ((1, 3, 1, 5, 8), None), ((1, 2, 6, 1, 32), None), ((1, 1, 3, 37, 512), None),
                                        ((2, 2, 4, 7, 24), [[1, 0], [0, 2], [8, 0]]), ((2, 3, 6, 11, 32), [[0, 1], [2, 1], [0, 8]])])
def test_3d_forwards_through_lds_vs_oracle(abi, shape, crop, dt):
    """3-D interpolating forward (two source planes per step, blends shared along the reference's nesting: same bits as
    interp_nd) and 3-D sparse forward of 2-byte elements through the one-step LDS kernel; crops included"""
    tdt = {"f32": torch.float32, "f64": torch.float64, "f16": torch.float16, "bf16": torch.bfloat16}[dt]
    rs = np.random.RandomState(sum(shape) * 3 + 11)
    xt = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt)
    wt = torch.from_numpy(_weights(rs, shape[1], 3, shape[2:])).to(tdt)
    es = xt.element_size()
    b, new = abi.check_borders(list(shape), crop, 3)
    if (new[-1] * es) % 16 or (shape[-1] * es) % 16:
        pytest.skip("rows are not whole 16-byte pieces")
    wide = np.float64 if tdt == torch.float64 else np.float32
    x, w = xt.to(torch.float64).numpy().astype(wide), wt.to(torch.float64).numpy().astype(wide)
    xd, wd = xt.to(DEV), wt.to(DEV)
    abi.set_tuning(35, 16)  # (not the walk through the planes: test_3d_walk_forward_vs_oracle)
    for groups in (2, 3):
        abi.set_tuning(34, groups)
        for pad in range(5):
            if es >= 4:   # (16-bit volumes interpolate in walk_forward16 / the sliding window: test_3d_walk_forward_vs_oracle, test_slide_gpu.py)
                out = abi.forward(xd, wd, pad, 1, b)
                assert abi.last_kernel() == "step_active_forward", (shape, abi.last_kernel())
                ref = torch.from_numpy(O.forward(x, w, pad, 1, b)).to(tdt)
                assert torch.equal(out.cpu(), ref), ("active", shape, crop, pad, groups)
            if es == 2:
                out = abi.forward(xd, wd, pad, 0, b)
                assert abi.last_kernel() == "step_gather_forward_lds", (shape, abi.last_kernel())
                assert torch.equal(out.cpu(), torch.from_numpy(O.forward(x, w, pad, 0, b)).to(tdt)), ("ssl", shape, crop, pad)
    abi.set_tuning(34, 2)


@pytest.mark.parametrize("dt", ["f32", "f64", "f16", "bf16"])
@pytest.mark.parametrize("shape", [(2, 3, 5, 6, 16), (1, 2, 20, 9, 64), (2, 2, 3, 40, 112), (1, 2, 2, 1, 32), (1, 1, 3, 37, 512),
                                   (1, 2, 16, 112, 112), (2, 2, 7, 300, 8), (1, 3, 4, 33, 2048 // 8), (2, 160, 5, 48, 112)])
def test_3d_walk_forward_vs_oracle(abi, shape, dt):
    """walk_forward (csrc/shiftnd_step.hip): the 3-D interpolating forward as a walk through the planes -- one plane staged
    per step, the other plane's corner rows carried in registers; blends nested as the reference nests them (fp32 / fp64
    bit-exact with the oracle, 16-bit within 1 ulp); every padding; ragged last row steps, one-row planes, rows of one and of
    256 pieces, shifts beyond every dim"""
    tdt = {"f32": torch.float32, "f64": torch.float64, "f16": torch.float16, "bf16": torch.bfloat16}[dt]
    rs = np.random.RandomState(sum(shape) * 5 + 3)
    xt = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt)
    wt = torch.from_numpy(_weights(rs, shape[1], 3, shape[2:])).to(tdt)
    es = xt.element_size()
    if (shape[-1] * es) % 16 or shape[-1] * es // 16 > 128:
        pytest.skip("rows are not whole 16-byte pieces / wider than half a workgroup pass (R + 1 rows of pieces, one per thread)")
    wide = np.float64 if tdt == torch.float64 else np.float32
    x, w = xt.to(torch.float64).numpy().astype(wide), wt.to(torch.float64).numpy().astype(wide)
    xd, wd = xt.to(DEV), wt.to(DEV)
    abi.set_tuning(34, 0)
    abi.set_tuning(35, 32)  # every float dtype (automatic: 16-bit only)
    for pad in range(5):
        out = abi.forward(xd, wd, pad, 1)
        assert abi.last_kernel() == ("walk_forward16" if es == 2 else "walk_forward"), (shape, abi.last_kernel())   # 16-bit: shiftnd_walk.hip
        ref = torch.from_numpy(O.forward(x, w, pad, 1)).to(tdt)
        if es >= 4:
            assert torch.equal(out.cpu(), ref), (shape, dt, pad)
        else:
            assert _ulp_close(out.cpu(), ref, tdt, FLOOR16), (shape, dt, pad)


@pytest.mark.parametrize("dt", ["f32", "f64", "f16", "bf16"])
@pytest.mark.parametrize("shape", [(2, 3, 5, 6, 16), (1, 2, 20, 9, 64), (2, 2, 3, 40, 112), (1, 2, 2, 1, 32), (1, 1, 3, 37, 512),
                                   (1, 2, 16, 112, 112), (2, 2, 7, 300, 8), (1, 3, 9, 33, 256),
                                   (2, 160, 5, 48, 112)])   # 960 workgroups: several per CU (the race of walk_barrier, shiftnd_walk.hip)
def test_3d_walk_backward_vs_oracle(abi, shape, dt):
    """walk_backward (csrc/shiftnd_step.hip): the 3-D interpolating backward as a walk through the planes -- one plane of the
    saved input and one of the gradient staged per step, the other corner planes carried in registers, the weight-gradient
    sums accumulated over the walk.  grad_x: fp32 / fp64 bit-exact with the oracle, 16-bit within 1 ulp; grad_w within
    1e-5 / 1e-12 / the 16-bit epsilon of the fp64 evaluation; every padding, ragged row steps, shifts beyond every dim"""
    tdt = {"f32": torch.float32, "f64": torch.float64, "f16": torch.float16, "bf16": torch.bfloat16}[dt]
    rs = np.random.RandomState(sum(shape) * 7 + 1)
    xt = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt)
    gt = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt)
    wt = torch.from_numpy(_weights(rs, shape[1], 3, shape[2:])).to(tdt)
    es = xt.element_size()
    if (shape[-1] * es) % 16 or shape[-1] * es // 16 > 128:
        pytest.skip("rows are not whole 16-byte pieces / wider than half a workgroup pass (R + 1 rows of pieces, one per thread)")
    wide = np.float64 if tdt == torch.float64 else np.float32
    x, go, w = (t.to(torch.float64).numpy().astype(wide) for t in (xt, gt, wt))
    xd, god, wd = xt.to(DEV), gt.to(DEV), wt.to(DEV)
    abi.set_tuning(35, 32)  # every float dtype (automatic: 2- and 4-byte elements)
    for pad in range(5):
        for active in (1, 0):   # the sparse shift: one gradient tap copied (bit-exact in every dtype), the same corner sums
            gx, gw = abi.backward(god, wd, xd, pad, active)
            want = ("walk_backward16" if es == 2 else "walk_backward") + ("" if active else "_sparse")   # 16-bit data: shiftnd_walk.hip (round 4)
            assert abi.last_kernel() == want, (shape, abi.last_kernel())
            gx_ref = torch.from_numpy(O.backward(go, w, x, pad, active)[0]).to(tdt)
            if es >= 4 or not active:
                assert torch.equal(gx.cpu(), gx_ref), ("gx", shape, dt, pad, active)
            else:
                assert _ulp_close(gx.cpu(), gx_ref, tdt, FLOOR16), ("gx", shape, dt, pad)
            _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active)
            tol = {"f32": 1e-5, "f64": 1e-12}.get(dt, gw16_tol(torch.finfo(tdt).eps))
            assert rel_err(gw.to(torch.float64).cpu().numpy(), gw64) < tol, ("gw", shape, dt, pad, active)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("shape", [(2, 20, 5, 6, 16), (1, 20, 3, 9, 64), (2, 20, 4, 40, 112), (1, 20, 2, 3, 8), (2, 160, 5, 48, 112)])
def test_3d_walk_small_shifts_vs_oracle(abi, shape, dt):
    """The folded row ends of the generic-padding walk kernels (shiftnd_walk.hip, WalkWindow / walk_park_guards: for |column shift|
    <= 6 the threads that stage a row's first and last piece park a permuted copy in the row's guard pieces and every window is five
    plain dwords): weights as the reference initialises them, uniform in (-1, 1), plus every integral column shift in -8 .. 8 and
    the half-way ones around the limit (7 and beyond keep the 9-address gather path); paddings 1 .. 4, forward and both backwards;
    rows of one chunk (no guards), two chunks (the first and the last piece are neighbours), many; 960 workgroups.  16-bit
    interpolation within 1 ulp, the sparse shift bit-exact, grad_w within half an ulp of the largest entry"""
    tdt = {"f16": torch.float16, "bf16": torch.bfloat16}[dt]
    rs = np.random.RandomState(sum(shape) * 3 + 11)
    C = shape[1]
    w = rs.uniform(-0.999, 0.999, size=(C, 3))
    w[0] = [1.0, -1.0, 1.0]
    w[1] = [0.0, 0.5, -1.0]
    w[2] = [-0.75, 1.0, 0.0]
    for k, col in enumerate([2.5, -2.0, 3.0, -4.5, 5.0, -5.75, 6.0, -6.0, 6.5, -6.5, 7.0, -7.0, 8.0, -8.25, 5.5, -3.25, 4.0]):
        w[3 + k] = [0.25 * ((k % 5) - 2), -0.5 * ((k % 3) - 1), col]
    xt = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt)
    gt = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt)
    wt = torch.from_numpy(w).to(tdt)
    x, go, wn = (t.float().numpy() for t in (xt, gt, wt))
    xd, god, wd = xt.to(DEV), gt.to(DEV), wt.to(DEV)
    for pad in (1, 2, 3, 4):
        out = abi.forward(xd, wd, pad, 1)
        assert abi.last_kernel() == "walk_forward16", abi.last_kernel()
        assert _ulp_close(out.cpu(), torch.from_numpy(O.forward(x, wn, pad, 1)).to(tdt), tdt, FLOOR16), ("fwd", shape, dt, pad)
        for active in (1, 0):
            gx, gw = abi.backward(god, wd, xd, pad, active)
            assert abi.last_kernel() == "walk_backward16" + ("" if active else "_sparse"), abi.last_kernel()
            gx_ref = torch.from_numpy(O.backward(go, wn, x, pad, active)[0]).to(tdt)
            if active:
                assert _ulp_close(gx.cpu(), gx_ref, tdt, FLOOR16), ("gx", shape, dt, pad)
            else:
                assert torch.equal(gx.cpu(), gx_ref), ("gx sparse", shape, dt, pad)
            _, gw64 = O.backward(go.astype(np.float64), wn.astype(np.float64), x.astype(np.float64), pad, active)
            assert rel_err(gw.to(torch.float64).cpu().numpy(), gw64) < gw16_tol(torch.finfo(tdt).eps), ("gw", shape, dt, pad, active)


@pytest.mark.parametrize("dt", ["f32", "f64", "i32"])
@pytest.mark.parametrize("shape,crop", [((2, 3, 5, 6, 16), None), ((1, 2, 4, 40, 112), None), ((2, 2, 6, 9, 32), [[1, 0], [0, 2], [4, 8]]),
                                        ((1, 3, 1, 5, 8), None), ((1, 2, 3, 1, 1024), None)])
def test_3d_gather_forward_vs_oracle(abi, shape, crop, dt):
    """step_gather_forward<..., 3>: the 3-D sparse shift of 4- / 8-byte elements as the linear sweep of one-step workgroups (float
    weights; quantized int32 tensors take it when their weights are floats' stand-ins -- here: the float op on int32 bit
    patterns is not a thing, so int32 goes through forward_quantized and must NOT take this kernel); crops, every padding,
    shifts beyond every dim; bit-exact"""
    rs = np.random.RandomState(sum(shape) + 77)
    b, new = abi.check_borders(list(shape), crop, 3)
    if dt == "i32":
        xq = rs.randint(-1000, 1000, size=shape).astype(np.int32)
        wq = rs.randint(124, 133, size=(shape[1], 3)).astype(np.uint8)
        for pad in range(5):
            out = abi.forward_quantized(torch.from_numpy(xq).to(DEV), torch.from_numpy(wq).to(DEV), 128, 5, pad, b)
            assert abi.last_kernel() != "step_gather_forward", abi.last_kernel()   # (integer weights: the sweep / plane kernels)
            assert np.array_equal(out.cpu().numpy(), O.forward_q(xq, wq, 128, 5, pad, b)), (shape, crop, pad)
        return
    npdt = np.float32 if dt == "f32" else np.float64
    if (new[-1] * np.dtype(npdt).itemsize) % 16 or new[-1] * np.dtype(npdt).itemsize // 16 > 256:
        pytest.skip("output rows are not whole 16-byte pieces / wider than one workgroup pass")
    x = rs.uniform(-1, 1, size=shape).astype(npdt)
    w = _weights(rs, shape[1], 3, shape[2:]).astype(npdt)
    xd, wd = torch.from_numpy(x).to(DEV), torch.from_numpy(w).to(DEV)
    for pad in range(5):
        out = abi.forward(xd, wd, pad, 0, b)
        assert abi.last_kernel() == "step_gather_forward", (shape, abi.last_kernel())
        assert np.array_equal(out.cpu().numpy(), O.forward(x, w, pad, 0, b)), (shape, crop, pad)
