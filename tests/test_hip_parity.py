"""GPU parity: the HIP kernels, called through the C ABI (torchshifts.abi -> libshiftnd_hip.so),
against (a) the golden fixtures recorded from the real reference and (b) the CPU oracle
(oracle/shift_oracle.c) on seeded random problems that engage every kernel family.

Bars:
  SSL forward, SSL grad_x, quantized ........ bit-exact
  active forward / active grad_x, fp32/fp64 .. bit-exact (same expression order, no FMA contraction);
                                               tolerance stated where used: 1e-5 relative
  grad_w ..................................... <= 1e-5 relative to the fp64 oracle (exact fixtures: bit-exact)
  fp16 / bf16 ................................ oracle in fp32 on widened inputs, rounded once (RNE):
                                               SSL bit-exact; interpolation within 1 ulp of the 16-bit type
"""
import numpy as np
import pytest
import torch

from cases import float_cases, quant_cases, rel_err, gw16_tol
from oracle import oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def abi():
    from torchshifts import abi as A
    assert torch.cuda.is_available(), "the gpu tests need an MI355X"
    yield A
    A.set_path_policy(0)


def _np_dt(dt):
    return np.float32 if dt == "f32" else np.float64


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("policy", [0, 1])
def test_golden_float_grid(abi, policy):
    """120 reference cases x {automatic path (sweep / plane / strided as eligible), forced strided fallback}"""
    abi.set_path_policy(policy)
    for key, nd, dt, pad, active, crop, x, w, go_full, out_r, gx_r, gw_r in float_cases("g1_float.npz"):
        b, _ = abi.check_borders(list(x.shape), crop, nd)
        xd, wd = _dev(x), _dev(w)
        out = abi.forward(xd, wd, pad, active, b)
        assert np.array_equal(out.cpu().numpy(), out_r), "forward " + key
        go = _dev(go_full[tuple(slice(0, s) for s in out_r.shape)])
        gx, gw = abi.backward(go, wd, xd, pad, active, b)
        assert np.array_equal(gx.cpu().numpy(), gx_r), "grad_x " + key
        assert np.array_equal(gw.cpu().numpy(), gw_r), "grad_w (exact data) " + key
    abi.set_path_policy(0)


def test_golden_random_grid(abi):
    for key, nd, dt, pad, active, crop, x, w, go_full, out_r, gx_r, gw_r in float_cases("g1_random.npz"):
        b, _ = abi.check_borders(list(x.shape), crop, nd)
        xd, wd = _dev(x), _dev(w)
        out = abi.forward(xd, wd, pad, active, b)
        assert np.array_equal(out.cpu().numpy(), out_r), "forward " + key
        go = _dev(go_full[tuple(slice(0, s) for s in out_r.shape)])
        gx, gw = abi.backward(go, wd, xd, pad, active, b)
        assert np.array_equal(gx.cpu().numpy(), gx_r), "grad_x " + key
        tol = 1e-12 if dt == "f64" else 1e-5
        assert rel_err(gw.cpu().numpy(), gw_r) < tol, "grad_w " + key


@pytest.mark.parametrize("policy", [0, 1, 2, 3, 4])
def test_golden_quantized(abi, policy):
    abi.set_path_policy(policy)
    tdt = {"quint8": torch.uint8, "qint8": torch.int8, "qint32": torch.int32}
    for key, nd, xname, layout, wname, pad, crop, xq, xzp, wq, wzp, out_r in quant_cases():
        x = torch.from_numpy(xq).to(tdt[xname]).to(DEV)
        if layout == "cl":
            x = x.contiguous(memory_format=torch.channels_last)
        elif layout == "cl3d":
            x = x.contiguous(memory_format=torch.channels_last_3d)
        w = torch.from_numpy(wq).to(torch.uint8 if wname == "wu8" else torch.int8).to(DEV)
        b, _ = abi.check_borders(list(xq.shape), crop, nd)
        if policy in (2, 3) and layout != "nchw":
            continue  # plane / sweep kernels need contiguous NC[spatial] tensors
        if policy == 4 and (layout == "nchw" or x.shape[1] < 2):
            continue  # the channel-fastest kernels serve channels-last inputs
        out = None
        if layout != "nchw":  # the quantized op keeps the input's channels-last format (shifts_quantized.cpp:119-121)
            out = torch.empty(abi.out_shape(x, b), dtype=x.dtype, device=DEV).contiguous(
                memory_format=torch.channels_last if layout == "cl" else torch.channels_last_3d)
        out = abi.forward_quantized(x, w, wzp, xzp, pad, b, out=out)
        assert np.array_equal(out.cpu().numpy(), out_r), key
        if layout != "nchw" and policy in (0, 4) and x.shape[1] > 1:
            # (pixel lines of whole 16-byte pieces, rows folded once: the LDS-tiled kernel; windows too since round 4, and NDHWC
            #  tensors of 2- / 4-byte elements)
            lines = (x.shape[1] * x.element_size()) % 16 == 0 and (x.shape[-2] == 1 or x.shape[-2] > 3)
            want = "cl_gather_forward"
            if lines and nd == 2:
                want = "cl_tiled_forward"
            if lines and nd == 3 and x.element_size() >= 2:
                want = "cl_tiled_forward_3d"
            if not (abi.last_path() == abi.PATH_CL and abi.last_kernel() == want):
                abi.set_path_policy(0)
            assert abi.last_path() == abi.PATH_CL and abi.last_kernel() == want, (key, abi.last_kernel())
        if policy in (0, 2, 3) and layout == "nchw":
            assert abi.last_path() == (abi.PATH_SWEEP if policy == 3 else abi.PATH_PLANE), key
    abi.set_path_policy(0)


# ---- seeded random problems vs the oracle: shapes with 16-byte rows so the plane kernels run ---------
SHAPES = [
    # (shape, crop or None)
    ((3, 5, 40), None), ((2, 3, 64), [[3, 5]]),
    ((3, 5, 9, 24), None), ((2, 4, 13, 32), [[2, 1], [4, 4]]), ((2, 3, 1, 16), None), ((1, 2, 300, 8), None),
    ((2, 3, 5, 6, 16), None), ((2, 2, 4, 7, 24), [[1, 0], [0, 2], [4, 4]]), ((1, 3, 1, 5, 8), None),
]


def _weights(rs, C, nd, sizes):
    w = rs.uniform(-4.5, 4.5, size=(C, nd))
    w[0, :] = [0.5, -1.5, 2.5][:nd]          # exact halves: round-half-even
    if C > 1:
        w[1, :] = [s + 2.25 for s in sizes]  # beyond the dim: multi-wrap
    if C > 2:
        w[2, :] = [-(2 * s + 0.75) for s in sizes]
    return w


@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("shape,crop", SHAPES)
def test_random_vs_oracle(abi, shape, crop, dt):
    rs = np.random.RandomState(sum(shape) * 131 + len(shape) + (dt == "f64"))
    nd = len(shape) - 2
    npdt = _np_dt(dt)
    x = rs.uniform(-1, 1, size=shape).astype(npdt)
    w = _weights(rs, shape[1], nd, shape[2:]).astype(npdt)
    b, new = abi.check_borders(list(shape), crop, nd)
    go = rs.uniform(-1, 1, size=new).astype(npdt)
    xd, wd, god = _dev(x), _dev(w), _dev(go)
    for pad in range(5):
        for active in (0, 1):
            out = abi.forward(xd, wd, pad, active, b)
            inner_bytes = new[-1] * x.itemsize
            if active == 0 or inner_bytes % 16 == 0:
                assert abi.last_path() == abi.PATH_PLANE  # small planes: plane kernels by default
            ref_out = O.forward(x, w, pad, active, b)
            assert np.array_equal(out.cpu().numpy(), ref_out), ("fwd", pad, active)
            if active == 0 or inner_bytes % 16 == 0:  # the sweep kernels serve the same problems
                abi.set_path_policy(3)
                outs = abi.forward(xd, wd, pad, active, b)
                abi.set_path_policy(0)
                assert abi.last_path() == abi.PATH_SWEEP and np.array_equal(outs.cpu().numpy(), ref_out)
            abi.set_path_policy(2)  # the plane kernels serve the same problems
            try:
                outp = abi.forward(xd, wd, pad, active, b)
                assert abi.last_path() == abi.PATH_PLANE and np.array_equal(outp.cpu().numpy(), ref_out)
            except RuntimeError:
                assert active == 1 and inner_bytes % 16 != 0
            abi.set_path_policy(0)
            gx, gw = abi.backward(god, wd, xd, pad, active, b)
            aligned = (shape[-1] * x.itemsize) % 16 == 0
            if aligned:
                assert abi.last_path() == abi.PATH_PLANE
            gx_o, _ = O.backward(go, w, x, pad, active, b)
            assert np.array_equal(gx.cpu().numpy(), gx_o), ("gx", pad, active)
            # grad_w truth: the oracle in fp64 on the same values
            _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active, b)
            assert rel_err(gw.cpu().numpy(), gw64) < (1e-12 if dt == "f64" else 1e-5), ("gw", pad, active)
            if aligned:  # the sweep kernels serve the same problems
                abi.set_path_policy(3)
                gxp, gwp = abi.backward(god, wd, xd, pad, active, b)
                abi.set_path_policy(0)
                assert abi.last_path() == abi.PATH_SWEEP and torch.equal(gxp, gx)
                assert rel_err(gwp.cpu().numpy(), gw64) < (1e-12 if dt == "f64" else 1e-5)
            # the kernel families agree bit for bit on forward / grad_x
            abi.set_path_policy(1)
            out2 = abi.forward(xd, wd, pad, active, b)
            gx2, gw2 = abi.backward(god, wd, xd, pad, active, b)
            abi.set_path_policy(0)
            assert torch.equal(out, out2) and torch.equal(gx, gx2)
            assert rel_err(gw2.cpu().numpy(), gw64) < (1e-12 if dt == "f64" else 1e-5)


def _ulp_close(a, ref, tdt, floor=0.0):
    """|a - ref| <= 1 ulp of the 16-bit type at ref.  `floor` (absolute; 0 unless a caller passes it -- tools/fuzz_round2.py does,
    with the reasoning and the operand scale at the call): the fp32 evaluation error of a result that cancels to almost nothing.
    The kernels evaluate 16-bit blends with mul + fma (and, in 3-D, inner-first), the oracle with mul, mul, add in the
    reference's nesting: a few fp32 ulps OF THE OPERANDS, which exceeds one 16-bit ulp of a result a thousand times smaller
    than they are."""
    a32, r32 = a.float(), ref.float()
    eps = torch.finfo(tdt).eps
    ulp = torch.clamp(r32.abs(), min=torch.finfo(tdt).tiny) * eps
    return bool(((a32 - r32).abs() <= ulp * 1.0001 + floor).all())


@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape,crop", [((3, 5, 48), None), ((2, 4, 11, 32), [[2, 1], [8, 8]]), ((2, 3, 4, 6, 16), None),
                                        ((2, 3, 7, 10), None)])
def test_half_precision_vs_oracle(abi, shape, crop, tdt):
    rs = np.random.RandomState(7)
    nd = len(shape) - 2
    x16 = torch.from_numpy(rs.uniform(-1, 1, size=shape).astype(np.float32)).to(tdt)
    w16 = torch.from_numpy(_weights(rs, shape[1], nd, shape[2:]).astype(np.float32)).to(tdt)
    b, new = abi.check_borders(list(shape), crop, nd)
    go16 = torch.from_numpy(rs.uniform(-1, 1, size=new).astype(np.float32)).to(tdt)
    x, w, go = x16.float().numpy(), w16.float().numpy(), go16.float().numpy()  # widened, exactly
    xd, wd, god = x16.to(DEV), w16.to(DEV), go16.to(DEV)
    for pad in range(5):
        for active in (0, 1):
            out = abi.forward(xd, wd, pad, active, b).cpu()
            ref = torch.from_numpy(O.forward(x, w, pad, active, b)).to(tdt)  # one RNE rounding
            gx, gw = abi.backward(god, wd, xd, pad, active, b)
            gx_o, _ = O.backward(go, w, x, pad, active, b)
            gx_ref = torch.from_numpy(gx_o).to(tdt)
            if active == 0:
                assert torch.equal(out, ref) and torch.equal(gx.cpu(), gx_ref), (pad, active)
            else:
                assert _ulp_close(out, ref, tdt) and _ulp_close(gx.cpu(), gx_ref, tdt), (pad, active)
            _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active, b)
            assert rel_err(gw.float().cpu().numpy(), gw64) < gw16_tol(torch.finfo(tdt).eps), ("gw", pad, active)


def test_quantized_random_vs_oracle(abi):
    rs = np.random.RandomState(11)
    for shape, crop, tdt, npdt, zp in [((4, 6, 56, 56), None, torch.uint8, np.uint8, 3),
                                       ((3, 5, 17, 24), [[1, 2], [8, 0]], torch.int8, np.int8, -5),
                                       ((2, 3, 9, 5, 16), None, torch.uint8, np.uint8, 128),
                                       ((2, 4, 33), None, torch.int32, np.int32, 7),
                                       ((2, 3, 13, 7), None, torch.int8, np.int8, 0)]:
        nd = len(shape) - 2
        info = np.iinfo(npdt)
        xq = rs.randint(max(info.min, -1000), min(info.max, 1000) + 1, size=shape).astype(npdt)
        wq = rs.randint(118, 139, size=(shape[1], nd)).astype(np.uint8)
        wq[0, :] = 128 + shape[-1] + 3  # shift beyond the dim
        b, new = abi.check_borders(list(shape), crop, nd)
        for pad in range(5):
            ref = O.forward_q(xq, wq, 128, zp, pad, b)
            for policy, path in ((3, abi.PATH_SWEEP), (2, abi.PATH_PLANE)):
                abi.set_path_policy(policy)
                out = abi.forward_quantized(torch.from_numpy(xq).to(tdt).to(DEV), torch.from_numpy(wq).to(DEV), 128, zp, pad, b)
                assert abi.last_path() == path
                assert np.array_equal(out.cpu().numpy(), ref), (shape, pad, policy)
            abi.set_path_policy(0)


def test_strided_inputs_and_empty(abi):
    """channels-last inputs take the channel-fastest kernels, sliced inputs the strided path; both equal the
    contiguous result; empty tensors are no-ops"""
    torch.manual_seed(0)
    x = torch.rand(3, 6, 10, 16, device=DEV)
    w = (torch.rand(6, 2, device=DEV) - 0.5) * 6
    go = torch.rand(3, 6, 10, 16, device=DEV)
    for pad in range(5):
        for active in (0, 1):
            ref = abi.forward(x, w, pad, active)
            xcl = x.contiguous(memory_format=torch.channels_last)
            out = abi.forward(xcl, w, pad, active)
            assert abi.last_path() == abi.PATH_STRIDED and torch.equal(out, ref)  # NCHW output: pixel-fastest kernels
            xs = torch.rand(3, 6, 10, 32, device=DEV)[..., ::2]
            assert torch.equal(abi.forward(xs, w, pad, active), abi.forward(xs.contiguous(), w, pad, active))
            assert abi.last_path() == abi.PATH_PLANE  # (the second call; the sliced one ran the strided kernels)
            gref = abi.backward(go, w, x, pad, active)
            gx, gw = abi.backward(go.contiguous(memory_format=torch.channels_last), w, xcl, pad, active)
            assert torch.equal(gx, gref[0]) and rel_err(gw.cpu().numpy(), gref[1].cpu().numpy()) < 1e-5
            # expanded (stride-0) incoming gradient, as produced by out.sum().backward()
            ones = torch.ones(1, device=DEV).expand(3, 6, 10, 16)
            gx1, gw1 = abi.backward(ones, w, x, pad, active)
            gx2, gw2 = abi.backward(torch.ones(3, 6, 10, 16, device=DEV), w, x, pad, active)
            # grad_out == 1: the weight gradient telescopes to ~0 for the wrapping paddings, so compare on the
            # scale of the summed terms (480 per channel, each <= 1), not of the (cancelled) result
            assert torch.equal(gx1, gx2) and float((gw1 - gw2).abs().max()) < 1e-5 * 480
    e = torch.empty(0, 6, 10, 16, device=DEV)
    assert abi.forward(e, w, 0, 0).shape == (0, 6, 10, 16) and abi.last_path() == abi.PATH_EMPTY
    gx, gw = abi.backward(e, w, e, 0, 0)
    assert gx.numel() == 0 and bool((gw == 0).all())


def test_large_plane_band_split_and_wide_rows(abi):
    """few large planes -> row bands; rows wider than one workgroup pass (cpr > 256)"""
    rs = np.random.RandomState(5)
    for shape in [(1, 2, 600, 64), (1, 1, 6, 4400), (2, 1, 3, 2048)]:
        x = rs.uniform(-1, 1, size=shape).astype(np.float32)
        w = rs.uniform(-3, 3, size=(shape[1], 2)).astype(np.float32)
        go = rs.uniform(-1, 1, size=shape).astype(np.float32)
        xd, wd, god = _dev(x), _dev(w), _dev(go)
        for pad in (0, 2, 3):
            for active in (0, 1):
                out = abi.forward(xd, wd, pad, active)
                big_plane = shape[2] * shape[3] * 4 >= 32768
                assert abi.last_path() == (abi.PATH_SWEEP if (not active and big_plane) else abi.PATH_PLANE)
                assert np.array_equal(out.cpu().numpy(), O.forward(x, w, pad, active))
                abi.set_path_policy(3)
                assert torch.equal(abi.forward(xd, wd, pad, active), out) and abi.last_path() == abi.PATH_SWEEP
                abi.set_path_policy(2)
                assert torch.equal(abi.forward(xd, wd, pad, active), out) and abi.last_path() == abi.PATH_PLANE
                abi.set_path_policy(0)
                gx_o, _ = O.backward(go, w, x, pad, active)
                _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active)
                for policy, path in ((3, abi.PATH_SWEEP), (0, abi.PATH_PLANE)):
                    abi.set_path_policy(policy)
                    gx, gw = abi.backward(god, wd, xd, pad, active)
                    assert abi.last_path() == path
                    assert np.array_equal(gx.cpu().numpy(), gx_o) and rel_err(gw.cpu().numpy(), gw64) < 1e-5
                abi.set_path_policy(0)


def test_deterministic_weight_grad(abi):
    torch.manual_seed(3)
    x = torch.rand(16, 8, 32, 32, device=DEV)
    go = torch.rand(16, 8, 32, 32, device=DEV)
    w = (torch.rand(8, 2, device=DEV) - 0.5) * 5
    ref = abi.backward(go, w, x, 3, 1)
    for _ in range(5):
        got = abi.backward(go, w, x, 3, 1)
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])


@pytest.mark.parametrize("tiles", [1, 2])
def test_lds_staged_kernels(abi, tiles):
    """the LDS-staged kernels (default, tuning knob 3 = 2: active forward and backward, 2-D / 3-D, no crop) vs
    the direct-load plane kernels (knob 3 = 1): forward and grad_x bit for bit, grad_w to rounding"""
    rs = np.random.RandomState(33)
    cases = [(3, 5, 9, 24), (2, 4, 13, 32), (2, 3, 1, 16), (1, 2, 300, 8), (2, 3, 40, 224), (1, 2, 7, 1000),
             (2, 3, 5, 6, 16), (2, 2, 4, 7, 24), (1, 3, 1, 5, 8), (1, 2, 3, 40, 112), (1, 2, 6, 1, 32)]
    abi.set_tuning(12, 0)  # (the sliding-window kernels would take the 3-D cases: tests/test_slide_gpu.py)
    try:
        for shape in cases:
            nd = len(shape) - 2
            for dt, tdt in ((np.float32, None), (np.float64, None), (np.float32, torch.bfloat16), (np.float32, torch.float16)):
                x = rs.uniform(-1, 1, size=shape).astype(dt)
                w = _weights(rs, shape[1], nd, shape[2:]).astype(dt)
                go = rs.uniform(-1, 1, size=shape).astype(dt)
                xd, wd, god = _dev(x), _dev(w), _dev(go)
                if tdt is not None:
                    xd, wd, god = xd.to(tdt), wd.to(tdt), god.to(tdt)
                for pad in range(5):
                    for active in (0, 1):
                        abi.set_tuning(3, 1)
                        out0 = abi.forward(xd, wd, pad, active)
                        gx0, gw0 = abi.backward(god, wd, xd, pad, active)
                        abi.set_tuning(3, 2)
                        abi.set_tuning(4, 2 if tiles == 2 else 3)  # 2: two LDS tiles + one barrier per step, 3: one tile
                        out1 = abi.forward(xd, wd, pad, active)
                        gx1, gw1 = abi.backward(god, wd, xd, pad, active)
                        assert abi.last_path() == abi.PATH_PLANE
                        assert torch.equal(out0, out1), (shape, dt, tdt, pad, active)
                        assert torch.equal(gx0, gx1), (shape, dt, tdt, pad, active)
                        assert rel_err(gw1.float().cpu().numpy(), gw0.float().cpu().numpy()) < 1e-5, (shape, pad, active)
    finally:
        abi.set_tuning(3, 2)
        abi.set_tuning(4, 1)  # 1 = automatic tile choice
        abi.set_tuning(12, -1)


def test_lds_staged_gather_forward(abi):
    """LDS-staged gather forward (default for 16-bit rows; every 2/4/8-byte type with tuning knob 2 = 16) vs the
    oracle, with and without crop, all paddings; int32 rows use the quantized entry (fill = zero point)"""
    rs = np.random.RandomState(44)
    cases = [((3, 5, 9, 24), None), ((2, 4, 13, 32), [[2, 1], [8, 8]]), ((2, 3, 5, 6, 16), None),
             ((2, 2, 4, 7, 24), [[1, 0], [0, 2], [8, 0]]), ((1, 2, 300, 8), None), ((2, 3, 40, 224), None)]
    try:
        abi.set_path_policy(2)
        for knob in (4, 16):
            abi.set_tuning(2, knob)
            for shape, crop in cases:
                nd = len(shape) - 2
                b, new = abi.check_borders(list(shape), crop, nd)
                x = rs.uniform(-1, 1, size=shape).astype(np.float32)
                w = _weights(rs, shape[1], nd, shape[2:]).astype(np.float32)
                for tdt in (torch.float32, torch.float64, torch.float16, torch.bfloat16):
                    xd, wd = _dev(x).to(tdt), _dev(w).to(tdt)
                    x_np = xd.cpu().double().numpy() if tdt != torch.float32 else x
                    w_np = wd.cpu().double().numpy() if tdt != torch.float32 else w
                    for pad in range(5):
                        out = abi.forward(xd, wd, pad, 0, b)
                        expect_lds = (knob == 16 or tdt in (torch.float16, torch.bfloat16))
                        if expect_lds and (shape[-1] * xd.element_size()) % 16 == 0 and (new[-1] * xd.element_size()) % 16 == 0:
                            assert abi.last_kernel() == "plane_gather_forward_lds", (shape, tdt, knob, abi.last_kernel())
                        ref = O.forward(x_np, w_np, pad, 0, b)
                        assert np.array_equal(out.cpu().double().numpy(), ref.astype(np.float64)), (shape, tdt, pad, knob)
                # int32 "quantized" rows
                xq = rs.randint(-1000, 1000, size=shape).astype(np.int32)
                wq = rs.randint(120, 137, size=(shape[1], nd)).astype(np.uint8)
                for pad in range(5):
                    outq = abi.forward_quantized(torch.from_numpy(xq).to(DEV), torch.from_numpy(wq).to(DEV), 128, 7, pad, b)
                    assert np.array_equal(outq.cpu().numpy(), O.forward_q(xq, wq, 128, 7, pad, b)), (shape, pad, knob)
    finally:
        abi.set_tuning(2, 4)
        abi.set_path_policy(0)


def test_huge_and_special_shifts_on_device(abi):
    """shifts far beyond the dims (multi-wrap, beyond 2^31: the 64-bit paths of canon_shift / pad_index), exact
    halves and signed zeros, for every kernel family, vs the oracle"""
    x = np.arange(2 * 8 * 12 * 16, dtype=np.float64).reshape(2, 8, 12, 16)
    go = np.cos(np.arange(x.size, dtype=np.float64)).reshape(x.shape)
    w = np.array([[1e6 + 3, -(1e6) - 7], [2.0 ** 31 + 5, -(2.0 ** 33) - 11], [2.0 ** 40 + 1, 2.0 ** 30],
                  [-(2.0 ** 30), 0.5], [-0.0, 0.0], [1.5, -2.5], [12.0, -16.0], [11.5, 15.5]])
    xd, wd, god = _dev(x), _dev(w), _dev(go)
    try:
        for policy in (0, 1, 2, 3):
            abi.set_path_policy(policy)
            for pad in range(5):
                for active in (0, 1):
                    out = abi.forward(xd, wd, pad, active)
                    assert np.array_equal(out.cpu().numpy(), O.forward(x, w, pad, active)), (policy, pad, active)
                    gx, gw = abi.backward(god, wd, xd, pad, active)
                    gx_o, gw_o = O.backward(go, w, x, pad, active)
                    assert np.array_equal(gx.cpu().numpy(), gx_o), (policy, pad, active)
                    assert rel_err(gw.cpu().numpy(), gw_o) < 1e-12, (policy, pad, active)
    finally:
        abi.set_path_policy(0)


@pytest.mark.parametrize("dt", ["f32", "f64"])
def test_channels_last_kernels_vs_oracle(abi, dt):
    """channels-last inputs run the channel-fastest kernels (policy 0 picks them, policy 4 forces them): forward
    (NCHW-contiguous output, like the reference, and channels-last output), backward with channels-last and
    contiguous grad_out / grad_x, every padding mode, crops, huge shifts; bit-exact vs the oracle (grad_w <= 1e-5)"""
    npdt = _np_dt(dt)
    rs = np.random.RandomState(21)
    for shape, crop, fmt in [((2, 8, 9, 12), None, torch.channels_last), ((3, 300, 6, 5), None, torch.channels_last),
                             ((2, 5, 11, 7), [[2, 1], [1, 3]], torch.channels_last),
                             ((2, 6, 4, 5, 6), None, torch.channels_last_3d),
                             ((1, 3, 5, 4, 7), [[1, 0], [0, 1], [2, 2]], torch.channels_last_3d)]:
        nd = len(shape) - 2
        x = rs.uniform(-1, 1, size=shape).astype(npdt)
        w = _weights(rs, shape[1], nd, shape[2:]).astype(npdt)
        b, new = abi.check_borders(list(shape), crop, nd)
        go = rs.uniform(-1, 1, size=new).astype(npdt)
        xd = _dev(x).contiguous(memory_format=fmt)
        wd = _dev(w)
        for pad in range(5):
            for active in (0, 1):
                ref = O.forward(x, w, pad, active, b)
                gx_o, _ = O.backward(go, w, x, pad, active, b)
                _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active, b)
                # the LDS-tiled kernels (shiftnd_cl_tiled.hip) serve 4-byte 2-D tensors whose pixel lines are whole pieces
                # (windows too since round 4; the cropped case of this list has 5 channels)
                tiled = dt == "f32" and nd == 2 and shape[1] % 4 == 0
                for policy in (0, 4):  # 0 picks the channel-fastest kernels when every tensor is channels-last
                    abi.set_path_policy(policy)
                    out = abi.forward(xd, wd, pad, active, b)  # NCHW-contiguous output (shifts_cpu.cpp:221)
                    assert abi.last_path() == (abi.PATH_CL if (policy == 4 or tiled) else abi.PATH_STRIDED), (shape, pad, active)
                    assert np.array_equal(out.cpu().numpy(), ref), (shape, pad, active)
                    out_cl = torch.empty(new, dtype=xd.dtype, device=DEV).contiguous(memory_format=fmt)
                    abi.forward(xd, wd, pad, active, b, out=out_cl)
                    assert abi.last_path() == abi.PATH_CL, (shape, pad, active)
                    assert abi.last_kernel() == (("cl_tiled_active_forward" if tiled else "cl_active_forward") if active else
                                                 ("cl_tiled_forward" if tiled else "cl_gather_forward"))
                    assert np.array_equal(out_cl.cpu().numpy(), ref), (shape, pad, active)
                    for god in (_dev(go), _dev(go).contiguous(memory_format=fmt)):
                        for gxd in (torch.empty_like(xd), torch.empty(shape, dtype=xd.dtype, device=DEV)):
                            gx, gw = abi.backward(god, wd, xd, pad, active, b, grad_x=gxd)
                            all_cl = god.stride(1) == 1 and gxd.stride(1) == 1
                            # (NCHW gradient + channels-last grad_x: the mixed form of the LDS-tiled backward)
                            mixed = tiled and policy == 0 and god.is_contiguous() and gxd.stride(1) == 1
                            assert abi.last_path() == (abi.PATH_CL if (policy == 4 or all_cl or mixed) else abi.PATH_STRIDED)
                            if mixed and shape[1] > 1:
                                assert abi.last_kernel() == "cl_tiled_backward_nchw_grad"
                            assert np.array_equal(gx.cpu().numpy(), gx_o), (shape, pad, active)
                            assert rel_err(gw.cpu().numpy(), gw64) < (1e-12 if dt == "f64" else 1e-5), (shape, pad, active)
                abi.set_path_policy(0)


def test_channels_last_16bit_and_large(abi):
    """bf16 channels-last (one RNE rounding) and a large channels-last problem against the NCHW kernels"""
    torch.manual_seed(4)
    x = torch.rand(8, 192, 56, 56, device=DEV)
    w = torch.rand(192, 2, device=DEV) * 6 - 3
    go = torch.rand_like(x)
    xc = x.contiguous(memory_format=torch.channels_last)
    for pad, active in ((0, 0), (3, 1), (2, 0)):
        ref = abi.forward(x, w, pad, active)
        gx_r, gw_r = abi.backward(go, w, x, pad, active)
        out = abi.forward(xc, w, pad, active, out=torch.empty_like(xc))
        assert abi.last_path() == abi.PATH_CL and torch.equal(out, ref)
        gx, gw = abi.backward(go.contiguous(memory_format=torch.channels_last), w, xc, pad, active, grad_x=torch.empty_like(xc))
        assert abi.last_path() == abi.PATH_CL and torch.equal(gx, gx_r)
        assert rel_err(gw.cpu().numpy(), gw_r.cpu().numpy()) < 1e-5
    xb, wb, gb = x.bfloat16(), w.bfloat16(), go.bfloat16()
    xbc = xb.contiguous(memory_format=torch.channels_last)
    for pad, active in ((0, 0), (4, 1)):
        ref = abi.forward(xb, wb, pad, active)
        out = abi.forward(xbc, wb, pad, active, out=torch.empty_like(xbc))
        assert abi.last_path() == abi.PATH_CL and torch.equal(out, ref)
        gx_r, gw_r = abi.backward(gb, wb, xb, pad, active)
        gx, gw = abi.backward(gb, wb, xbc, pad, active, grad_x=torch.empty_like(xbc))
        assert torch.equal(gx, gx_r)
        assert rel_err(gw.float().cpu().numpy(), gw_r.float().cpu().numpy()) < 2 * gw16_tol(torch.finfo(torch.bfloat16).eps)   # (two roundings)
