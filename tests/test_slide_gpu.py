"""GPU parity of the sliding-window kernels (csrc/shiftnd_slide.hip: backward pass and interpolating forward of
contiguous 2-D / 3-D problems without crop) against the CPU oracle, through the C ABI.

Bars as everywhere: fp32 forward / grad_x bit-exact (the kernels nest the blends like
interpolation.h:34-40), grad_w <= 1e-5 relative to the fp64 oracle; 16-bit: the sparse shift bit-exact,
interpolation within 1 ulp of the 16-bit type.  Launch-planning knobs (12: which problems slide, 13: workgroups
wanted, 14: minimum rows per band) are swept so that ragged bands, ragged plane groups and ragged unit groups run."""
import numpy as np
import pytest
import torch

from cases import rel_err, gw16_tol
from oracle import oracle as O
from test_hip_parity import _ulp_close, _weights

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture()
def abi():
    from torchshifts import abi as A
    assert torch.cuda.is_available()
    A.set_path_policy(0)
    A.set_tuning(12, 3)  # 2-D and 3-D problems slide
    A.set_tuning(32, 1)  # (the one-step kernels, which take 2-D problems first, have their own file: test_step_gpu.py)
    A.set_tuning(34, 1)
    yield A
    for k, v in ((12, -1), (13, 0), (14, 16), (32, 0), (34, 0)):
        A.set_tuning(k, v)


SHAPES = [
    (2, 3, 5, 6, 16), (1, 2, 20, 9, 64), (2, 2, 3, 40, 112), (1, 3, 1, 5, 8), (1, 2, 6, 1, 32), (1, 1, 3, 37, 512),
    (3, 5, 9, 24), (2, 3, 40, 224), (5, 2, 33, 64), (1, 2, 300, 8), (2, 3, 1, 16), (7, 2, 6, 56), (1, 2, 7, 1000), (2, 1, 5, 1024),
]
PLANS = [(0, 16), (100000, 2), (64, 5)]  # (knob 13, knob 14)


@pytest.mark.parametrize("plan", PLANS)
@pytest.mark.parametrize("shape", SHAPES)
def test_fp32_vs_oracle(abi, shape, plan):
    rs = np.random.RandomState(sum(shape) * 17 + plan[1])
    nd = len(shape) - 2
    x = rs.uniform(-1, 1, size=shape).astype(np.float32)
    go = rs.uniform(-1, 1, size=shape).astype(np.float32)
    w = _weights(rs, shape[1], nd, shape[2:]).astype(np.float32)
    xd, wd, god = (torch.from_numpy(a).to(DEV) for a in (x, w, go))
    abi.set_tuning(13, plan[0])
    abi.set_tuning(14, plan[1])
    for pad in range(5):
        for active in (0, 1):
            if active:
                out = abi.forward(xd, wd, pad, active)
                assert abi.last_kernel() == "slide_forward", (shape, abi.last_kernel())
                assert np.array_equal(out.cpu().numpy(), O.forward(x, w, pad, active)), ("fwd", shape, pad)
            gx, gw = abi.backward(god, wd, xd, pad, active)
            assert abi.last_kernel() == "slide_backward", (shape, abi.last_kernel())
            gx_o, _ = O.backward(go, w, x, pad, active)
            assert np.array_equal(gx.cpu().numpy(), gx_o), ("gx", shape, pad, active)
            _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active)
            assert rel_err(gw.cpu().numpy(), gw64) < 1e-5, ("gw", shape, pad, active)


@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 3, 5, 6, 16), (1, 2, 20, 9, 64), (2, 2, 3, 40, 112), (3, 5, 9, 24), (2, 3, 40, 224),
                                   (5, 2, 33, 64), (1, 2, 300, 8)])
def test_16bit_vs_oracle(abi, shape, tdt):
    rs = np.random.RandomState(sum(shape) + 5)
    nd = len(shape) - 2
    x16 = torch.from_numpy(rs.uniform(-1, 1, size=shape).astype(np.float32)).to(tdt)
    go16 = torch.from_numpy(rs.uniform(-1, 1, size=shape).astype(np.float32)).to(tdt)
    w16 = torch.from_numpy(_weights(rs, shape[1], nd, shape[2:]).astype(np.float32)).to(tdt)
    x, w, go = x16.float().numpy(), w16.float().numpy(), go16.float().numpy()
    xd, wd, god = x16.to(DEV), w16.to(DEV), go16.to(DEV)
    for plan in PLANS[:2]:
        abi.set_tuning(13, plan[0])
        abi.set_tuning(14, plan[1])
        for pad in range(5):
            for active in (0, 1):
                gx, gw = abi.backward(god, wd, xd, pad, active)
                assert abi.last_kernel() == "slide_backward"
                gx_ref = torch.from_numpy(O.backward(go, w, x, pad, active)[0]).to(tdt)
                if active:
                    out = abi.forward(xd, wd, pad, active)
                    assert abi.last_kernel() == "slide_forward"
                    ref = torch.from_numpy(O.forward(x, w, pad, active)).to(tdt)
                    assert _ulp_close(out.cpu(), ref, tdt), ("fwd", shape, pad)
                    assert _ulp_close(gx.cpu(), gx_ref, tdt), ("gx", shape, pad)
                else:
                    assert torch.equal(gx.cpu(), gx_ref), ("gx", shape, pad)
                _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active)
                assert rel_err(gw.float().cpu().numpy(), gw64) < gw16_tol(torch.finfo(tdt).eps), ("gw", shape, pad, active)


def test_agrees_with_the_step_tiled_kernels(abi):
    """same bits as the LDS-staged step kernels of shiftnd_plane.hip (forward, grad_x), deterministic grad_w"""
    torch.manual_seed(1)
    for shape, tdt in (((2, 8, 6, 24, 64), torch.float32), ((2, 8, 6, 24, 64), torch.bfloat16), ((4, 6, 48, 96), torch.float16)):
        nd = len(shape) - 2
        x = torch.rand(shape, device=DEV).to(tdt)
        go = torch.rand(shape, device=DEV).to(tdt)
        w = ((torch.rand(shape[1], nd, device=DEV) - 0.5) * 7).to(tdt)
        for pad in (0, 3):
            for active in (0, 1):
                abi.set_tuning(12, 0)
                gx0, gw0 = abi.backward(go, w, x, pad, active)
                assert abi.last_kernel().startswith("plane_backward")
                out0 = abi.forward(x, w, pad, active)
                abi.set_tuning(12, 3)
                gx1, gw1 = abi.backward(go, w, x, pad, active)
                gx2, gw2 = abi.backward(go, w, x, pad, active)
                assert abi.last_kernel() == "slide_backward"
                assert torch.equal(gx1, gx2) and torch.equal(gw1, gw2)  # deterministic
                if active and tdt != torch.float32:
                    # 16-bit interpolation: each family is within 1 ulp of the oracle (the compiler may fuse the final
                    # rounding into the last multiply-add in one kernel and not in the other)
                    assert _ulp_close(gx1.cpu(), gx0.cpu(), tdt)
                else:
                    assert torch.equal(gx0, gx1)
                assert rel_err(gw1.float().cpu().numpy(), gw0.float().cpu().numpy()) < (1e-5 if tdt == torch.float32 else 2e-2)
                if active:
                    out1 = abi.forward(x, w, pad, active)
                    assert torch.equal(out1, out0) if tdt == torch.float32 else _ulp_close(out1.cpu(), out0.cpu(), tdt)


@pytest.mark.parametrize("shape,tdt,npdt,zp", [((4, 6, 56, 56), torch.uint8, np.uint8, 3), ((9, 5, 8, 16), torch.int8, np.int8, -5),
                                               ((5, 3, 9, 5, 16), torch.uint8, np.uint8, 128), ((3, 4, 28, 28), torch.int8, np.int8, 0),
                                               ((130, 2, 4, 4), torch.uint8, np.uint8, 7), ((2, 3, 1, 112, 32), torch.uint8, np.uint8, 255)])
def test_small_plane_byte_kernel_vs_oracle(shape, tdt, npdt, zp):
    """bytes_gather_forward (csrc/shiftnd_bytes.hip): the quantized forward on planes that are whole 16-byte pieces;
    bit-exact against the oracle's quantized kernel (shifts_kernels.h:532-571) for every padding, shifts beyond the
    dims, ragged last workgroups (knob 17 = planes per workgroup)"""
    from torchshifts import abi
    rs = np.random.RandomState(sum(shape))
    nd = len(shape) - 2
    info = np.iinfo(npdt)
    xq = rs.randint(info.min, info.max + 1, size=shape).astype(npdt)
    wq = rs.randint(121, 136, size=(shape[1], nd)).astype(np.uint8)
    wq[0, :] = 128 + shape[-1] + 3  # a shift beyond the dim
    if shape[1] > 1:
        wq[1, :] = 128 - 2 * shape[-1] - 1
    x = torch.from_numpy(xq).to(tdt).to(DEV)
    w = torch.from_numpy(wq).to(DEV)
    try:
        for ppw in (0, 1, 3):
            abi.set_tuning(17, ppw)
            for pad in range(5):
                out = abi.forward_quantized(x, w, 128, zp, pad)
                assert abi.last_kernel() == "bytes_gather_forward", (shape, abi.last_kernel())
                assert np.array_equal(out.cpu().numpy(), O.forward_q(xq, wq, 128, zp, pad)), (shape, pad, ppw)
        abi.set_tuning(16, 0)  # the row-chunk kernel serves the same problem
        out = abi.forward_quantized(x, w, 128, zp, 0)
        assert abi.last_kernel() != "bytes_gather_forward"
        assert np.array_equal(out.cpu().numpy(), O.forward_q(xq, wq, 128, zp, 0))
    finally:
        abi.set_tuning(16, 1)
        abi.set_tuning(17, 0)


@pytest.mark.parametrize("shape", [(2, 8, 9, 12), (3, 300, 6, 5), (2, 64, 40, 70), (1, 36, 100, 33), (2, 4, 1, 50), (1, 32, 64, 1)])
def test_tiled_channels_last_forward_vs_oracle(shape):
    """cl_tiled_forward (csrc/shiftnd_cl_tiled.hip): dense channels-last fp32 / int32 input, channels-last or
    NCHW-contiguous output, every padding (periodic: the wrapped edge pixels / rows through the element pass), shifts beyond the
    ring (gathered from memory), ragged tiles / channel blocks / bands; bit-exact"""
    from torchshifts import abi
    rs = np.random.RandomState(sum(shape) + 3)
    x = rs.uniform(-1, 1, size=shape).astype(np.float32)
    w = rs.uniform(-3.4, 3.4, size=(shape[1], 2)).astype(np.float32)
    w[0] = [0.5, -1.5]
    w[1] = [shape[2] + 2.25, -7.0]       # beyond the dim / beyond the ring
    w[2] = [-5.0, 2.5]
    xd = torch.from_numpy(x).to(DEV).contiguous(memory_format=torch.channels_last)
    wd = torch.from_numpy(w).to(DEV)
    try:
        for band_rows in (0, 5):
            abi.set_tuning(21, band_rows)
            for pad in (0, 1, 2, 3, 4):
                ref = O.forward(x, w, pad, False)
                out = abi.forward(xd, wd, pad, False)  # NCHW-contiguous output
                assert abi.last_kernel() == "cl_tiled_forward" and out.is_contiguous()
                assert np.array_equal(out.cpu().numpy(), ref), (shape, pad, "nchw")
                out_cl = torch.empty(shape, device=DEV).contiguous(memory_format=torch.channels_last)
                abi.forward(xd, wd, pad, False, out=out_cl)
                assert abi.last_kernel() == "cl_tiled_forward"
                assert np.array_equal(out_cl.cpu().numpy(), ref), (shape, pad, "cl")
        abi.set_tuning(21, 0)
        # int32 quantized: fill = the input's zero point
        xq = rs.randint(-1000, 1000, size=shape).astype(np.int32)
        wq = rs.randint(124, 133, size=(shape[1], 2)).astype(np.uint8)
        xqd = torch.from_numpy(xq).to(DEV).contiguous(memory_format=torch.channels_last)
        outq = torch.empty(shape, dtype=torch.int32, device=DEV).contiguous(memory_format=torch.channels_last)
        abi.forward_quantized(xqd, torch.from_numpy(wq).to(DEV), 128, -7, 0, out=outq)
        assert abi.last_kernel() == "cl_tiled_forward"
        assert np.array_equal(outq.cpu().numpy(), O.forward_q(xq, wq, 128, -7, 0))
    finally:
        abi.set_tuning(21, 0)


CL_CROPS = [((2, 8, 9, 12), [[1, 1], [1, 1]]), ((3, 300, 6, 5), [[0, 2], [1, 0]]), ((2, 64, 40, 70), [[1, 1], [1, 1]]),
            ((1, 36, 100, 33), [[7, 30], [0, 5]]), ((2, 4, 12, 50), [[11, 0], [3, 3]]), ((1, 32, 64, 9), [[0, 0], [4, 4]]),
            ((2, 16, 33, 40), [[5, 3], [2, 6]])]


@pytest.mark.parametrize("shape,crop", CL_CROPS)
def test_tiled_channels_last_cropped_forward_vs_oracle(shape, crop):
    """cl_tiled_forward with a window (round 4: the emulated `valid` cut of modules/shifts.py:41-46 on a channels-last
    input): the ring follows the source rows / pixels of the window, the output has the window's sizes; fp32 (channels-last
    and NCHW-contiguous outputs), quantized uint8 (channels-last, shifts_quantized.cpp:119-121) and bf16; every padding,
    shifts beyond the ring and beyond the dim, ragged bands; bit-exact vs the oracle"""
    from torchshifts import abi
    b, new = abi.check_borders(list(shape), crop, 2)
    rs = np.random.RandomState(sum(shape) + 29)
    x = rs.uniform(-1, 1, size=shape).astype(np.float32)
    w = rs.uniform(-3.4, 3.4, size=(shape[1], 2)).astype(np.float32)
    w[0] = [0.5, -1.5]
    w[1] = [shape[2] + 2.25, -7.0]       # beyond the dim / beyond the ring
    w[2] = [-5.0, 2.5]
    xd = torch.from_numpy(x).to(DEV).contiguous(memory_format=torch.channels_last)
    wd = torch.from_numpy(w).to(DEV)
    try:
        for band_rows in (0, 5):
            abi.set_tuning(21, band_rows)
            for pad in (0, 1, 2, 3, 4):
                ref = O.forward(x, w, pad, False, b)
                assert list(ref.shape) == list(new)
                out = abi.forward(xd, wd, pad, False, b)  # NCHW-contiguous output
                assert abi.last_kernel() == "cl_tiled_forward" and out.is_contiguous(), abi.last_kernel()
                assert np.array_equal(out.cpu().numpy(), ref), (shape, crop, pad, "nchw")
                out_cl = torch.empty(new, device=DEV).contiguous(memory_format=torch.channels_last)
                abi.forward(xd, wd, pad, False, b, out=out_cl)
                assert abi.last_kernel() == "cl_tiled_forward"
                assert np.array_equal(out_cl.cpu().numpy(), ref), (shape, crop, pad, "cl")
                if shape[2] >= 5:   # cl_tiled_active_forward (rows folded once: H >= 5)
                    ref = O.forward(x, w, pad, True, b)
                    out = abi.forward(xd, wd, pad, True, b)
                    assert abi.last_kernel() == "cl_tiled_active_forward" and out.is_contiguous(), abi.last_kernel()
                    assert np.array_equal(out.cpu().numpy(), ref), (shape, crop, pad, "active nchw")
                    abi.forward(xd, wd, pad, True, b, out=out_cl)
                    assert abi.last_kernel() == "cl_tiled_active_forward"
                    assert np.array_equal(out_cl.cpu().numpy(), ref), (shape, crop, pad, "active cl")
        abi.set_tuning(21, 0)
        if shape[1] % 16 == 0:
            xq = rs.randint(0, 255, size=shape).astype(np.uint8)
            wq = rs.randint(124, 133, size=(shape[1], 2)).astype(np.uint8)
            wq[1] = [128 + 9, 128 - 6]
            xqd = torch.from_numpy(xq).to(DEV).contiguous(memory_format=torch.channels_last)
            for pad in (0, 1, 2, 3, 4):
                outq = torch.empty(new, dtype=torch.uint8, device=DEV).contiguous(memory_format=torch.channels_last)
                abi.forward_quantized(xqd, torch.from_numpy(wq).to(DEV), 128, 7, pad, b, out=outq)
                assert abi.last_kernel() == "cl_tiled_forward"
                assert np.array_equal(outq.cpu().numpy(), O.forward_q(xq, wq, 128, 7, pad, b)), (shape, crop, pad, "u8")
        if shape[1] % 8 == 0:
            xb = torch.from_numpy(x).to(torch.bfloat16)
            wb = torch.from_numpy(w).to(torch.bfloat16)
            xbd = xb.to(DEV).contiguous(memory_format=torch.channels_last)
            for pad in (0, 1, 2, 3, 4):
                ref = O.forward(xb.float().numpy(), wb.float().numpy(), pad, False, b)
                out_cl = torch.empty(new, dtype=torch.bfloat16, device=DEV).contiguous(memory_format=torch.channels_last)
                abi.forward(xbd, wb.to(DEV), pad, False, b, out=out_cl)
                assert abi.last_kernel() == "cl_tiled_forward"
                assert np.array_equal(out_cl.float().cpu().numpy(), ref), (shape, crop, pad, "bf16")
                if shape[2] >= 5:
                    ref = torch.from_numpy(O.forward(xb.float().numpy(), wb.float().numpy(), pad, True, b)).to(torch.bfloat16)
                    abi.forward(xbd, wb.to(DEV), pad, True, b, out=out_cl)
                    assert abi.last_kernel() == "cl_tiled_active_forward"
                    assert _ulp_close(out_cl.cpu(), ref, torch.bfloat16), (shape, crop, pad, "bf16 active")
    finally:
        abi.set_tuning(21, 0)


CL3_CASES = [((2, 8, 4, 9, 12), None), ((1, 36, 3, 10, 33), None), ((2, 64, 5, 12, 40), [[1, 1], [1, 1], [1, 1]]),
             ((1, 32, 1, 6, 7), None), ((2, 4, 6, 1, 50), None), ((1, 12, 7, 8, 1), [[2, 3], [0, 0], [0, 0]]),
             ((1, 16, 16, 20, 35), [[0, 1], [3, 0], [2, 2]])]


@pytest.mark.parametrize("shape,crop", CL3_CASES)
def test_tiled_channels_last_3d_forward_vs_oracle(shape, crop):
    """cl_tiled_forward<..., ND3> (round 4, NDHWC): dense channels_last_3d fp32 / bf16 / int32 input, output channels_last_3d or
    NCDHW-contiguous; every padding, depth shifts of any size (through the staging address), row / column shifts beyond the
    ring, windows in all three dims, size-1 dims; bit-exact vs the oracle"""
    from torchshifts import abi
    cl3 = torch.channels_last_3d
    b, new = abi.check_borders(list(shape), crop, 3)
    rs = np.random.RandomState(sum(shape) + 57)
    x = rs.uniform(-1, 1, size=shape).astype(np.float32)
    w = rs.uniform(-3.4, 3.4, size=(shape[1], 3)).astype(np.float32)
    w[0] = [0.5, -1.5, 1.0]
    w[1] = [shape[2] + 2.25, shape[3] + 1.0, -7.0]       # beyond the dims / beyond the ring
    w[2] = [-5.0, 2.5, -2.0]
    w[3] = [1.0, 0.0, 0.0]
    xd = torch.from_numpy(x).to(DEV).contiguous(memory_format=cl3)
    wd = torch.from_numpy(w).to(DEV)
    try:
        for band_rows in (0, 5):
            abi.set_tuning(21, band_rows)
            for pad in (0, 1, 2, 3, 4):
                ref = O.forward(x, w, pad, False, b)
                out = abi.forward(xd, wd, pad, False, b)  # NCDHW-contiguous output
                assert abi.last_kernel() == "cl_tiled_forward_3d" and out.is_contiguous(), abi.last_kernel()
                assert np.array_equal(out.cpu().numpy(), ref), (shape, crop, pad, "ncdhw")
                out_cl = torch.empty(new, device=DEV).contiguous(memory_format=cl3)
                abi.forward(xd, wd, pad, False, b, out=out_cl)
                assert abi.last_kernel() == "cl_tiled_forward_3d"
                assert np.array_equal(out_cl.cpu().numpy(), ref), (shape, crop, pad, "cl")
                if shape[3] == 1 or shape[3] >= 5:   # the interpolating shift: the plane blend at staging time, bit-exact in fp32
                    ref = O.forward(x, w, pad, True, b)
                    out = abi.forward(xd, wd, pad, True, b)
                    assert abi.last_kernel() == "cl_tiled_active_forward_3d" and out.is_contiguous(), abi.last_kernel()
                    assert np.array_equal(out.cpu().numpy(), ref), (shape, crop, pad, "active ncdhw")
                    abi.forward(xd, wd, pad, True, b, out=out_cl)
                    assert abi.last_kernel() == "cl_tiled_active_forward_3d"
                    assert np.array_equal(out_cl.cpu().numpy(), ref), (shape, crop, pad, "active cl")
        abi.set_tuning(21, 0)
        # int32 quantized: fill = the input's zero point, format kept (shifts_quantized.cpp:119-121)
        xq = rs.randint(-1000, 1000, size=shape).astype(np.int32)
        wq = rs.randint(124, 133, size=(shape[1], 3)).astype(np.uint8)
        xqd = torch.from_numpy(xq).to(DEV).contiguous(memory_format=cl3)
        for pad in (0, 2, 3):
            outq = torch.empty(new, dtype=torch.int32, device=DEV).contiguous(memory_format=cl3)
            abi.forward_quantized(xqd, torch.from_numpy(wq).to(DEV), 128, -7, pad, b, out=outq)
            assert abi.last_kernel() == "cl_tiled_forward_3d"
            assert np.array_equal(outq.cpu().numpy(), O.forward_q(xq, wq, 128, -7, pad, b)), (shape, crop, pad, "i32")
        if shape[1] % 8 == 0:
            xb, wb = torch.from_numpy(x).to(torch.bfloat16), torch.from_numpy(w).to(torch.bfloat16)
            xbd = xb.to(DEV).contiguous(memory_format=cl3)
            for pad in (0, 1, 2, 3, 4):
                ref = O.forward(xb.float().numpy(), wb.float().numpy(), pad, False, b)
                out_cl = torch.empty(new, dtype=torch.bfloat16, device=DEV).contiguous(memory_format=cl3)
                abi.forward(xbd, wb.to(DEV), pad, False, b, out=out_cl)
                assert abi.last_kernel() == "cl_tiled_forward_3d"
                assert np.array_equal(out_cl.float().cpu().numpy(), ref), (shape, crop, pad, "bf16 cl")
                if (new[4] * 2) % 4 == 0:
                    out = abi.forward(xbd, wb.to(DEV), pad, False, b)
                    assert abi.last_kernel() == "cl_tiled_forward_3d" and out.is_contiguous()
                    assert np.array_equal(out.float().cpu().numpy(), ref), (shape, crop, pad, "bf16 ncdhw")
                if shape[3] == 1 or shape[3] >= 5:
                    refa = torch.from_numpy(O.forward(xb.float().numpy(), wb.float().numpy(), pad, True, b)).to(torch.bfloat16)
                    abi.forward(xbd, wb.to(DEV), pad, True, b, out=out_cl)
                    assert abi.last_kernel() == "cl_tiled_active_forward_3d"
                    assert _ulp_close(out_cl.cpu(), refa, torch.bfloat16, 8 * 2.0 ** -24), (shape, crop, pad, "bf16 active")
    finally:
        abi.set_tuning(21, 0)


CL3B_CASES = CL3_CASES + [((1, 8, 3, 5, 20), None), ((1, 40, 4, 30, 17), [[1, 0], [2, 3], [0, 1]]), ((2, 16, 2, 64, 16), None)]


@pytest.mark.parametrize("go_layout", ["ndhwc", "ncdhw"])
@pytest.mark.parametrize("shape,crop", CL3B_CASES)
def test_tiled_channels_last_3d_backward_vs_oracle(shape, crop, go_layout):
    """cl_tiled_backward_3d (round 5, shiftnd_cl_tiled3.hip): saved input and grad_x dense channels_last_3d, the incoming gradient
    channels_last_3d or NCDHW-contiguous (what follows the reference's float forward, cpu/shifts_cpu.cpp:221); both shifts, every
    padding, depth shifts of any size (through the staging address), row / column shifts beyond the ring, windows in all three
    dims, size-1 dims.  fp32: grad_x bit-exact, grad_w <= 1e-5 of the fp64 evaluation; bf16 / fp16: the sparse shift's grad_x
    bit-exact, interpolation within 1 ulp, grad_w within half a unit of the type."""
    from torchshifts import abi
    if not (shape[3] == 1 or shape[3] >= 5) or (shape[1] * 4) % 16:
        pytest.skip("the tiled kernels fold their source rows once (H == 1 or H >= 5); pixel lines of whole 16-byte pieces")
    cl3 = torch.channels_last_3d
    b, new = abi.check_borders(list(shape), crop, 3)
    if not (new[3] == 1 or new[3] >= 5):
        pytest.skip("window rows")
    rs = np.random.RandomState(sum(shape) + 91)
    x = rs.uniform(-1, 1, size=shape).astype(np.float32)
    go = rs.uniform(-1, 1, size=new).astype(np.float32)
    w = rs.uniform(-3.4, 3.4, size=(shape[1], 3)).astype(np.float32)
    w[0] = [0.5, -1.5, 1.0]
    w[1] = [shape[2] + 2.25, shape[3] + 1.0, -7.0]       # beyond the dims / beyond the ring
    w[2] = [-5.0, 2.5, -2.0]
    w[3] = [1.0, 0.0, 0.0]
    w[4 % shape[1]] = [-0.25, -3.0, 3.0]                 # the ring's edges (reflect: the corner one step beyond it)
    fmt = cl3 if go_layout == "ndhwc" else torch.contiguous_format
    name = "cl_tiled_backward_3d" if go_layout == "ndhwc" else "cl_tiled_backward_3d_ncdhw_grad"
    for tdt in (torch.float32, torch.bfloat16, torch.float16):
        es = 4 if tdt == torch.float32 else 2
        if (shape[1] * es) % 16:
            continue
        xt, gt, wt = (torch.from_numpy(a).to(tdt) for a in (x, go, w))
        xr, gr, wr = (t.float().numpy() for t in (xt, gt, wt))
        xd = xt.to(DEV).contiguous(memory_format=cl3)
        gd = gt.to(DEV).contiguous(memory_format=fmt)
        wd = wt.to(DEV)
        for pad in (0, 1, 2, 3, 4):
            for active in (0, 1):
                gx = torch.empty_like(xd)
                gx.fill_(float("nan"))   # (an element no kernel wrote fails the comparison)
                gx, gw = abi.backward(gd, wd, xd, pad, active, b, grad_x=gx)
                assert abi.last_kernel() == name, (abi.last_kernel(), shape, crop, tdt)
                gx_ref, _ = O.backward(gr, wr, xr, pad, active, b)
                _, gw64 = O.backward(gr.astype(np.float64), wr.astype(np.float64), xr.astype(np.float64), pad, active, b)
                tag = (shape, crop, go_layout, str(tdt), pad, active)
                if tdt == torch.float32:
                    assert np.array_equal(gx.cpu().numpy(), gx_ref), ("gx",) + tag
                    tol = max(1e-5, 2 * rel_err(O.backward(gr, wr, xr, pad, active, b)[1], gw64))   # (never looser than the reference's own fp32)
                    assert rel_err(gw.cpu().numpy(), gw64) < tol, ("gw",) + tag
                else:
                    ref16 = torch.from_numpy(gx_ref).to(tdt)
                    if active:
                        assert _ulp_close(gx.cpu(), ref16, tdt, 8 * 2.0 ** -24), ("gx",) + tag
                    else:
                        assert torch.equal(gx.cpu(), ref16), ("gx",) + tag
                    assert rel_err(gw.float().cpu().numpy(), gw64) < gw16_tol(torch.finfo(tdt).eps), ("gw",) + tag
                gx2, gw2 = abi.backward(gd, wd, xd, pad, active, b, grad_x=torch.empty_like(xd))
                assert torch.equal(gx2, gx) and torch.equal(gw2, gw), ("deterministic",) + tag


@pytest.mark.parametrize("shape", [(2, 16, 9, 12), (1, 144, 20, 37), (2, 64, 40, 70), (1, 32, 5, 6)])
def test_tiled_channels_last_forward_small_elements(shape):
    """cl_tiled_forward for 1- and 2-byte elements (a dword of output = 4 / 2 elements with their own shifts): quantized
    uint8 / int8 keeping the channels-last format (shifts_quantized.cpp:119-121), fp16 / bf16 sparse shifts to
    channels-last and NCHW-contiguous outputs; shifts beyond the ring; bit-exact vs the oracle"""
    from torchshifts import abi
    rs = np.random.RandomState(sum(shape) + 11)
    C = shape[1]
    try:
        for band_rows in (0, 7):
            abi.set_tuning(21, band_rows)
            for npdt, zp in ((np.uint8, 7), (np.int8, -3)):
                xq = rs.randint(0 if npdt == np.uint8 else -128, 127, size=shape).astype(npdt)
                wq = rs.randint(124, 133, size=(C, 2)).astype(np.uint8)
                wq[1] = [128 + 9, 128 - 6]   # beyond the ring
                wq[2] = [128 - 4, 128 + 3]
                xqd = torch.from_numpy(xq).to(DEV).contiguous(memory_format=torch.channels_last)
                wqd = torch.from_numpy(wq).to(DEV)
                for pad in (0, 1, 2, 3, 4):
                    outq = torch.empty(shape, dtype=xqd.dtype, device=DEV).contiguous(memory_format=torch.channels_last)
                    abi.forward_quantized(xqd, wqd, 128, zp, pad, out=outq)
                    assert abi.last_kernel() == "cl_tiled_forward", (shape, npdt, pad)
                    assert np.array_equal(outq.cpu().numpy(), O.forward_q(xq, wq, 128, zp, pad)), (shape, npdt, pad)
            if (C * 2) % 16 == 0:
                for tdt in (torch.float16, torch.bfloat16):
                    x = torch.from_numpy(rs.uniform(-1, 1, size=shape).astype(np.float32)).to(tdt)
                    w = torch.from_numpy(rs.uniform(-3.4, 3.4, size=(C, 2)).astype(np.float32)).to(tdt)
                    w[0, 0], w[0, 1] = 0.5, -1.5
                    w[1, 0], w[1, 1] = shape[2] + 2.25, -7.0
                    xd = x.to(DEV).contiguous(memory_format=torch.channels_last)
                    wd = w.to(DEV)
                    for pad in (0, 1, 2, 3, 4):
                        ref = abi.forward(x.to(DEV), wd, pad, False)   # the NCHW kernels (checked against the oracle elsewhere)
                        out = abi.forward(xd, wd, pad, False)  # NCHW-contiguous output
                        nchw_tiled = (shape[3] * 2) % 4 == 0
                        assert (abi.last_kernel() == "cl_tiled_forward") == nchw_tiled and out.is_contiguous()
                        assert torch.equal(out, ref), (shape, tdt, pad, "nchw")
                        out_cl = torch.empty(shape, dtype=tdt, device=DEV).contiguous(memory_format=torch.channels_last)
                        abi.forward(xd, wd, pad, False, out=out_cl)
                        assert abi.last_kernel() == "cl_tiled_forward"
                        assert torch.equal(out_cl, ref), (shape, tdt, pad, "cl")
    finally:
        abi.set_tuning(21, 0)


@pytest.mark.parametrize("shape", [(2, 8, 9, 12), (3, 300, 6, 5), (2, 64, 40, 70), (1, 36, 100, 33), (2, 4, 1, 50), (1, 32, 64, 1)])
def test_tiled_channels_last_backward_vs_oracle(shape):
    """cl_tiled_backward (csrc/shiftnd_cl_tiled.hip): fp32, saved input / grad_out / grad_x all dense channels-last;
    sparse and active shifts, every padding it serves, shifts beyond the ring, ragged tiles / channel blocks / bands;
    grad_x bit-exact, grad_w within 1e-5 of the fp64 oracle"""
    from torchshifts import abi
    rs = np.random.RandomState(sum(shape) + 5)
    x = rs.uniform(-1, 1, size=shape).astype(np.float32)
    go = rs.uniform(-1, 1, size=shape).astype(np.float32)
    w = rs.uniform(-3.4, 3.4, size=(shape[1], 2)).astype(np.float32)
    w[0] = [0.5, -1.5]
    w[1] = [shape[2] + 2.25, -7.0]       # beyond the dim / beyond the ring
    w[2] = [-5.0, 2.5]
    w[3] = [3.0, -3.0]
    cl = torch.channels_last
    xd = torch.from_numpy(x).to(DEV).contiguous(memory_format=cl)
    god = torch.from_numpy(go).to(DEV).contiguous(memory_format=cl)
    wd = torch.from_numpy(w).to(DEV)
    try:
        for band_rows in (0, 5):
            abi.set_tuning(21, band_rows)
            for pad in (0, 1, 2, 3, 4):
                for active in (0, 1):
                    gx_o, _ = O.backward(go, w, x, pad, active)
                    _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active)
                    gxd = torch.empty(shape, device=DEV).contiguous(memory_format=cl)
                    gx, gw = abi.backward(god, wd, xd, pad, active, grad_x=gxd)
                    assert abi.last_kernel() == "cl_tiled_backward", (shape, pad, active)
                    assert np.array_equal(gx.cpu().numpy(), gx_o), (shape, pad, active, band_rows)
                    assert rel_err(gw.cpu().numpy(), gw64) < 1e-5, (shape, pad, active, band_rows)
        abi.set_tuning(21, 0)
    finally:
        abi.set_tuning(21, 0)


CL_CROPS_BWD = [((2, 8, 9, 12), [[1, 1], [1, 1]]), ((3, 300, 8, 5), [[0, 2], [1, 0]]), ((2, 64, 40, 70), [[1, 1], [1, 1]]),
                ((1, 36, 100, 33), [[7, 30], [0, 5]]), ((2, 4, 12, 50), [[6, 0], [3, 3]]), ((1, 32, 64, 9), [[0, 0], [4, 4]]),
                ((2, 16, 33, 40), [[5, 3], [2, 6]]), ((1, 8, 7, 21), [[3, 3], [0, 20]]), ((2, 12, 30, 16), [[0, 25], [15, 0]]),
                ((2, 8, 12, 44), [[1, 1], [3, 5]])]   # (a window of 36 columns at column 3: NCHW gradient rows of whole pieces)


@pytest.mark.parametrize("go_layout", ["cl", "nchw"])
@pytest.mark.parametrize("shape,crop", CL_CROPS_BWD)
def test_tiled_channels_last_cropped_backward_vs_oracle(shape, crop, go_layout):
    """cl_tiled_backward with a window (round 4): saved input / grad_x channels-last at the full size, grad_out (channels-last or
    NCHW-contiguous) with the window's sizes; the gradient's maps fold in the WINDOW's sizes (shifts_kernels.h:402-527 through
    shifts.cpp:93-135), elements outside the window get zero and count nothing.  Windows one row / one column wide, shifts
    beyond the ring and beyond the dim, every padding, both shifts, ragged bands.  fp32: grad_x bit-exact, grad_w within 1e-5 of
    the fp64 oracle; bf16 sparse: grad_x bit-exact"""
    from torchshifts import abi
    b, new = abi.check_borders(list(shape), crop, 2)
    rs = np.random.RandomState(sum(shape) + 41)
    x = rs.uniform(-1, 1, size=shape).astype(np.float32)
    go = rs.uniform(-1, 1, size=new).astype(np.float32)
    w = rs.uniform(-3.4, 3.4, size=(shape[1], 2)).astype(np.float32)
    w[0] = [0.5, -1.5]
    w[1] = [shape[2] + 2.25, -7.0]       # beyond the dim / beyond the ring
    w[2] = [-5.0, 2.5]
    w[3] = [3.0, -3.0]
    cl = torch.channels_last
    xd = torch.from_numpy(x).to(DEV).contiguous(memory_format=cl)
    god = torch.from_numpy(go).to(DEV)
    if go_layout == "cl":
        god = god.contiguous(memory_format=cl)
    wd = torch.from_numpy(w).to(DEV)
    name = "cl_tiled_backward" if go_layout == "cl" else "cl_tiled_backward_nchw_grad"
    if new[2] * new[3] == 1:   # a 1 x 1 window is both layouts at once: either reading is right
        name = ("cl_tiled_backward", "cl_tiled_backward_nchw_grad")
    tiled = new[2] == 1 or new[2] >= 5   # (the kernel folds the gradient's rows once)
    try:
        for band_rows in (0, 5):
            abi.set_tuning(21, band_rows)
            for pad in (0, 1, 2, 3, 4):
                for active in (0, 1):
                    gx_o, _ = O.backward(go, w, x, pad, active, b)
                    _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active, b)
                    gxd = torch.empty(shape, device=DEV).contiguous(memory_format=cl)
                    gx, gw = abi.backward(god, wd, xd, pad, active, b, grad_x=gxd)
                    assert (abi.last_kernel() in name if isinstance(name, tuple) else abi.last_kernel() == name) == tiled, (shape, crop, pad, active, abi.last_kernel())
                    assert np.array_equal(gx.cpu().numpy(), gx_o), (shape, crop, pad, active, band_rows)
                    assert rel_err(gw.cpu().numpy(), gw64) < 1e-5, (shape, crop, pad, active, band_rows)
        abi.set_tuning(21, 0)
        if (shape[1] * 2) % 16 == 0:
            tdt = torch.bfloat16
            x16, go16, w16 = torch.from_numpy(x).to(tdt), torch.from_numpy(go).to(tdt), torch.from_numpy(w).to(tdt)
            xn, gn, wn = x16.float().numpy(), go16.float().numpy(), w16.float().numpy()
            g16 = go16.to(DEV).contiguous(memory_format=cl) if go_layout == "cl" else go16.to(DEV)
            for pad in (0, 1, 2, 3, 4):
                gx_o = torch.from_numpy(O.backward(gn, wn, xn, pad, 0, b)[0]).to(tdt)
                _, gw64 = O.backward(gn.astype(np.float64), wn.astype(np.float64), xn.astype(np.float64), pad, 0, b)
                gxd = torch.empty(shape, dtype=tdt, device=DEV).contiguous(memory_format=cl)
                gx, gw = abi.backward(g16, w16.to(DEV), x16.to(DEV).contiguous(memory_format=cl), pad, 0, b, grad_x=gxd)
                assert (abi.last_kernel() in name if isinstance(name, tuple) else abi.last_kernel() == name) == tiled, (shape, crop, pad, abi.last_kernel())
                assert torch.equal(gx.cpu(), gx_o), (shape, crop, pad, "bf16")
                assert rel_err(gw.float().cpu().numpy(), gw64) < gw16_tol(float(torch.finfo(tdt).eps)), (shape, crop, pad, "bf16")
    finally:
        abi.set_tuning(21, 0)


@pytest.mark.parametrize("shape", [(2, 8, 9, 12), (3, 300, 6, 5), (2, 64, 40, 70), (1, 36, 100, 33), (2, 4, 1, 50), (1, 32, 64, 1),
                                   (2, 64, 40, 72), (1, 36, 21, 36), (2, 12, 9, 16)])   # (rows of whole 16-byte pieces: the piece staging)
def test_tiled_backward_nchw_gradient_vs_oracle(shape):
    """cl_tiled_backward<GO_NCHW>: saved input channels-last, incoming gradient NCHW-contiguous (what an op downstream of
    the reference's float forward returns, cpu/shifts_cpu.cpp:221), grad_x in the input's layout -- one pass, no transpose.
    fp32: grad_x bit-exact with the oracle, grad_w within 1e-5 of its fp64 evaluation; fp16 / bf16 (sparse shift): grad_x
    bit-exact, grad_w within the type's epsilon"""
    from torchshifts import abi
    rs = np.random.RandomState(sum(shape) + 31)
    x = rs.uniform(-1, 1, size=shape).astype(np.float32)
    go = rs.uniform(-1, 1, size=shape).astype(np.float32)
    w = rs.uniform(-3.4, 3.4, size=(shape[1], 2)).astype(np.float32)
    w[0] = [0.5, -1.5]
    w[1] = [shape[2] + 2.25, -7.0]       # beyond the dim / beyond the ring
    w[2] = [-5.0, 2.5]
    w[3] = [3.0, -3.0]
    cl = torch.channels_last
    xd = torch.from_numpy(x).to(DEV).contiguous(memory_format=cl)
    god = torch.from_numpy(go).to(DEV)   # NCHW
    wd = torch.from_numpy(w).to(DEV)
    try:
        for band_rows in (0, 5):
            abi.set_tuning(21, band_rows)
            for pad in (0, 1, 2, 3, 4):
                for active in (0, 1):
                    gx_o, _ = O.backward(go, w, x, pad, active)
                    _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active)
                    gxd = torch.empty(shape, device=DEV).contiguous(memory_format=cl)
                    gx, gw = abi.backward(god, wd, xd, pad, active, grad_x=gxd)
                    assert abi.last_kernel() == "cl_tiled_backward_nchw_grad", (shape, pad, active)
                    assert np.array_equal(gx.cpu().numpy(), gx_o), (shape, pad, active, band_rows)
                    assert rel_err(gw.cpu().numpy(), gw64) < 1e-5, (shape, pad, active, band_rows)
        abi.set_tuning(21, 0)
        for tdt in (torch.float16, torch.bfloat16):
            if (shape[1] * 2) % 16:
                continue
            x16, go16, w16 = torch.from_numpy(x).to(tdt), torch.from_numpy(go).to(tdt), torch.from_numpy(w).to(tdt)
            xn, gn, wn = x16.float().numpy(), go16.float().numpy(), w16.float().numpy()
            eps = float(torch.finfo(tdt).eps)
            for pad in (0, 1, 2, 3, 4):
                gx_o = torch.from_numpy(O.backward(gn, wn, xn, pad, 0)[0]).to(tdt)
                _, gw64 = O.backward(gn.astype(np.float64), wn.astype(np.float64), xn.astype(np.float64), pad, 0)
                gxd = torch.empty(shape, dtype=tdt, device=DEV).contiguous(memory_format=cl)
                gx, gw = abi.backward(go16.to(DEV), w16.to(DEV), x16.to(DEV).contiguous(memory_format=cl), pad, 0, grad_x=gxd)
                assert abi.last_kernel() == "cl_tiled_backward_nchw_grad", (shape, tdt, pad)
                assert torch.equal(gx.cpu(), gx_o), (shape, tdt, pad)
                assert rel_err(gw.float().cpu().numpy(), gw64) < gw16_tol(eps), (shape, tdt, pad)
    finally:
        abi.set_tuning(21, 0)


def test_channels_last_module_trains_without_a_transpose():
    """Shift2d on a channels-last input through the dispatcher ops: the forward returns NCHW (as the reference's does), the
    gradient that comes back is NCHW, and the backward runs as ONE kernel that reads both as they lie and writes grad_x in
    the input's layout -- same values as the contiguous run"""
    import torchshifts
    from torchshifts import abi
    torch.manual_seed(3)
    for active in (False, True):
        m = torchshifts.Shift2d(64, init_shift=2, sparsity_term=0., active_flag=active).to(DEV)
        x = torch.rand(4, 64, 40, 36, device=DEV)
        t = torch.rand(4, 64, 40, 36, device=DEV)
        xc = x.clone().requires_grad_(True)
        out_c, _ = m(xc)
        torch.nn.functional.mse_loss(out_c, t).backward()
        gw_c, gx_c = m.weight.grad.clone(), xc.grad.clone()
        m.zero_grad()
        xl = x.clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        out_l, _ = m(xl)
        assert out_l.is_contiguous() and torch.equal(out_l, out_c)
        torch.nn.functional.mse_loss(out_l, t).backward()
        assert xl.grad.is_contiguous(memory_format=torch.channels_last)
        assert torch.equal(xl.grad, gx_c), active
        assert rel_err(m.weight.grad.cpu().numpy(), gw_c.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("shape", [(2, 8, 9, 12), (3, 300, 6, 5), (2, 64, 40, 70), (1, 36, 100, 33), (2, 4, 1, 50), (1, 32, 64, 1)])
def test_tiled_channels_last_active_forward_vs_oracle(shape):
    """cl_tiled_active_forward: fp32 channels-last input, interpolated output channels-last or NCHW-contiguous, every
    padding it serves, shifts beyond the ring, the reflected corner of the last row / column; bit-exact"""
    from torchshifts import abi
    rs = np.random.RandomState(sum(shape) + 9)
    x = rs.uniform(-1, 1, size=shape).astype(np.float32)
    w = rs.uniform(-3.9, 3.9, size=(shape[1], 2)).astype(np.float32)
    w[0] = [0.5, -1.5]
    w[1] = [shape[2] + 2.25, -7.0]       # beyond the dim / beyond the ring
    w[2] = [-5.0, 2.5]
    w[3] = [-2.75, -2.25]                # floor = -3: the + 1 corner at distance 4
    xd = torch.from_numpy(x).to(DEV).contiguous(memory_format=torch.channels_last)
    wd = torch.from_numpy(w).to(DEV)
    try:
        for band_rows in (0, 5):
            abi.set_tuning(21, band_rows)
            for pad in (0, 1, 2, 3, 4):
                ref = O.forward(x, w, pad, True)
                out = abi.forward(xd, wd, pad, True)  # NCHW-contiguous output
                assert abi.last_kernel() == "cl_tiled_active_forward" and out.is_contiguous()
                assert np.array_equal(out.cpu().numpy(), ref), (shape, pad, "nchw", band_rows)
                out_cl = torch.empty(shape, device=DEV).contiguous(memory_format=torch.channels_last)
                abi.forward(xd, wd, pad, True, out=out_cl)
                assert abi.last_kernel() == "cl_tiled_active_forward"
                assert np.array_equal(out_cl.cpu().numpy(), ref), (shape, pad, "cl", band_rows)
    finally:
        abi.set_tuning(21, 0)


@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 8, 9, 12), (2, 72, 40, 70), (1, 64, 100, 33), (2, 16, 1, 50)])
def test_tiled_channels_last_16bit_active_and_backward(shape, tdt):
    """cl_tiled_active_forward / cl_tiled_backward for fp16 and bf16 (64 channels per workgroup, 2-byte stores) against the
    ORACLE (widened inputs, one rounding: interpolation within 1 ulp of the 16-bit type, the sparse shift's grad_x bit-exact,
    grad_w within the 16-bit epsilon of the fp64 evaluation) -- and the same bits as the contiguous kernels (one definition
    of the 16-bit interpolation, shiftnd_common.hpp interp_t)"""
    from torchshifts import abi
    rs = np.random.RandomState(sum(shape) + 13)
    cl = torch.channels_last
    x = torch.from_numpy(rs.uniform(-1, 1, size=shape).astype(np.float32)).to(tdt).to(DEV)
    go = torch.from_numpy(rs.uniform(-1, 1, size=shape).astype(np.float32)).to(tdt).to(DEV)
    w = rs.uniform(-3.9, 3.9, size=(shape[1], 2)).astype(np.float32)
    w[0] = [0.5, -1.5]
    w[1] = [shape[2] + 2.25, -7.0]       # beyond the dim / beyond the ring
    w[2] = [-5.0, 2.5]
    w[3] = [-2.75, -2.25]
    wd = torch.from_numpy(w).to(tdt).to(DEV)
    xc, goc = x.contiguous(memory_format=cl), go.contiguous(memory_format=cl)
    xn, gn, wn = x.float().cpu().numpy(), go.float().cpu().numpy(), wd.float().cpu().numpy()
    eps = float(torch.finfo(tdt).eps)
    try:
        for band_rows in (0, 7):
            abi.set_tuning(21, band_rows)
            for pad in (0, 1, 2, 3, 4):
                ref = abi.forward(x, wd, pad, True)
                # the oracle on the widened values, one rounding to the 16-bit type: the bar for every 16-bit kernel
                ref_o = torch.from_numpy(O.forward(xn, wn, pad, True)).to(tdt)
                out = abi.forward(xc, wd, pad, True)   # NCHW-contiguous output
                nchw_tiled = (shape[3] * 2) % 4 == 0
                # (against the contiguous kernel family: within 1 ulp -- the compiler may fuse the final rounding to 16 bits into the last
                #  multiply-add of one family (v_fma_mixlo: one rounding) and not of the other (two); the oracle is the bar, below)
                assert (abi.last_kernel() == "cl_tiled_active_forward") == nchw_tiled and _ulp_close(out.cpu(), ref.cpu(), tdt), (shape, pad)
                assert _ulp_close(out.cpu(), ref_o, tdt), ("oracle", shape, pad)
                out_cl = torch.empty(shape, dtype=tdt, device=DEV).contiguous(memory_format=cl)
                abi.forward(xc, wd, pad, True, out=out_cl)
                assert abi.last_kernel() == "cl_tiled_active_forward" and _ulp_close(out_cl.cpu(), ref.cpu(), tdt), (shape, pad)
                assert _ulp_close(out_cl.cpu(), ref_o, tdt), ("oracle", shape, pad)
                for active in (0, 1):
                    gx_r, gw_r = abi.backward(go, wd, x, pad, active)
                    gx, gw = abi.backward(goc, wd, xc, pad, active, grad_x=torch.empty(shape, dtype=tdt, device=DEV).contiguous(memory_format=cl))
                    assert abi.last_kernel() == "cl_tiled_backward", (shape, pad, active)
                    assert (_ulp_close(gx.cpu(), gx_r.cpu(), tdt) if active else torch.equal(gx, gx_r)), (shape, pad, active, band_rows)
                    assert rel_err(gw.float().cpu().numpy(), gw_r.float().cpu().numpy()) < 2 * gw16_tol(eps), (shape, pad, active)   # (two roundings)
                    gx_o = torch.from_numpy(O.backward(gn, wn, xn, pad, active)[0]).to(tdt)
                    if active:
                        assert _ulp_close(gx.cpu(), gx_o, tdt), ("oracle gx", shape, pad)
                    else:
                        assert torch.equal(gx.cpu(), gx_o), ("oracle gx", shape, pad)
                    _, gw64 = O.backward(gn.astype(np.float64), wn.astype(np.float64), xn.astype(np.float64), pad, active)
                    assert rel_err(gw.float().cpu().numpy(), gw64) < gw16_tol(eps), ("oracle gw", shape, pad, active)
    finally:
        abi.set_tuning(21, 0)


def test_channels_last_input_through_the_op_uses_the_tiled_kernel():
    """torch.ops.torchshifts.shift2d with a channels-last fp32 input: one pass (no layout change first), NCHW result
    like the reference (cpu/shifts_cpu.cpp:221), same bits as the contiguous input"""
    import torchshifts  # noqa: F401
    from torchshifts import abi
    torch.manual_seed(4)
    x = torch.rand(4, 64, 56, 56, device=DEV)
    w = (torch.rand(64, 2, device=DEV) - 0.5) * 6
    ref = torch.ops.torchshifts.shift2d(x, w, torch.Tensor(), 3, False)
    out = torch.ops.torchshifts.shift2d(x.contiguous(memory_format=torch.channels_last), w, torch.Tensor(), 3, False)
    assert abi.last_kernel() == "cl_tiled_forward" and out.is_contiguous() and torch.equal(out, ref)


def test_channels_last_3d_input_through_the_op():
    """torch.ops.torchshifts.shift3d with an NDHWC (channels_last_3d) input: fp32 sparse shift in one pass through
    cl_tiled_forward_3d (NCDHW result like the reference, cpu/shifts_cpu.cpp:221); the interpolating shift, bf16 and the whole
    backward through the layout change + the contiguous kernels -- every result the same bits as the contiguous input's, the
    gradients flowing through autograd"""
    import torchshifts  # noqa: F401
    from torchshifts import abi
    torch.manual_seed(6)
    cl3 = torch.channels_last_3d
    x = torch.rand(2, 32, 6, 20, 24, device=DEV)
    w = (torch.rand(32, 3, device=DEV) - 0.5) * 4
    ref = torch.ops.torchshifts.shift3d(x, w, torch.Tensor(), 0, False)
    out = torch.ops.torchshifts.shift3d(x.contiguous(memory_format=cl3), w, torch.Tensor(), 0, False)
    assert abi.last_kernel() == "cl_tiled_forward_3d" and out.is_contiguous() and torch.equal(out, ref), abi.last_kernel()
    for active, tdt in ((True, torch.float32), (False, torch.bfloat16), (True, torch.bfloat16)):
        xa, wa = x.to(tdt), w.to(tdt)
        ref = torch.ops.torchshifts.shift3d(xa, wa, torch.Tensor(), 3, active)
        out = torch.ops.torchshifts.shift3d(xa.contiguous(memory_format=cl3), wa, torch.Tensor(), 3, active)
        assert torch.equal(out, ref), (active, tdt, abi.last_kernel())
    # training step: the same gradients as with a contiguous input
    grads = []
    for fmt in (torch.contiguous_format, cl3):
        xi = x.clone().contiguous(memory_format=fmt).requires_grad_(True)
        wi = w.clone().requires_grad_(True)
        y = torch.ops.torchshifts.shift3d(xi, wi, torch.Tensor(), 0, True)
        y.square().sum().backward()
        grads.append((xi.grad.contiguous(), wi.grad))
    assert torch.equal(grads[0][0], grads[1][0])
    assert torch.allclose(grads[0][1], grads[1][1], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("active", [False, True])
def test_channels_last_backward_through_the_op_uses_the_tiled_kernel(active):
    """torch.ops.torchshifts._shift2d_backward with channels-last fp32 saved input and incoming gradient: one pass (no
    layout change), grad_x channels-last, same values as the contiguous call"""
    import torchshifts  # noqa: F401
    from torchshifts import abi
    torch.manual_seed(5)
    cl = torch.channels_last
    x = torch.rand(4, 64, 56, 56, device=DEV)
    go = torch.rand(4, 64, 56, 56, device=DEV)
    w = (torch.rand(64, 2, device=DEV) - 0.5) * 6
    b6 = torch.tensor([0, 56, 0, 56, 0, 1], dtype=torch.int32)
    op = torch.ops.torchshifts._shift2d_backward
    gx_r, gw_r = op(go, w, x, b6, 3, active)
    gx, gw = op(go.contiguous(memory_format=cl), w, x.contiguous(memory_format=cl), b6, 3, active)
    assert abi.last_kernel() == "cl_tiled_backward" and gx.is_contiguous(memory_format=cl)
    assert torch.equal(gx, gx_r) and rel_err(gw.cpu().numpy(), gw_r.cpu().numpy()) < 1e-5
    # a contiguous incoming gradient: the layout of the saved input is changed once, then the contiguous kernels
    gx2, gw2 = op(go, w, x.contiguous(memory_format=cl), b6, 3, active)
    assert abi.last_kernel() != "cl_tiled_backward" and torch.equal(gx2, gx_r)


def test_byte_kernel_on_112x112_planes():
    """the largest plane the byte kernel takes with zeros padding (12544 bytes: the stem of a 224-pixel network after its
    first stride-2 layer); the other paddings need the byte table as well and take it while that fits LDS.  Bit-exact
    either way, rounds per workgroup 1 and 3 (knob 19)"""
    from torchshifts import abi
    rs = np.random.RandomState(112)
    shape = (5, 4, 112, 112)
    xq = rs.randint(0, 256, size=shape).astype(np.uint8)
    wq = rs.randint(123, 134, size=(4, 2)).astype(np.uint8)
    x, w = torch.from_numpy(xq).to(DEV), torch.from_numpy(wq).to(DEV)
    try:
        for rpw in (0, 1, 3):
            abi.set_tuning(19, rpw)
            for pad in range(5):
                out = abi.forward_quantized(x, w, 128, 9, pad)
                if pad == 0:
                    assert abi.last_kernel() == "bytes_gather_forward"
                assert np.array_equal(out.cpu().numpy(), O.forward_q(xq, wq, 128, 9, pad)), (pad, rpw)
    finally:
        abi.set_tuning(19, 0)
