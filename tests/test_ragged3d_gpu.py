"""3-D volumes whose rows are not whole 16-byte pieces, beyond the small-plane kernels' 16 KiB (16 x 28 x 28 bf16, 8 x 30 x 62 fp32 ...):
the direct-load plane kernels with 4- / 8-byte chunks (csrc/shiftnd_plane.hip: plane_ragged_forward / plane_ragged_backward) instead
of the one-thread-per-element fallback the route census found them on.  Every padding, both shifts, windows; fp32 / fp64 bit-exact
with the oracle, 16-bit within 1 ulp, grad_w within the parity bars (reference: kernels/shifts_kernels.h:156-327)."""
import numpy as np
import pytest
import torch

from cases import rel_err, gw16_tol
from oracle import oracle as O
from test_hip_parity import _ulp_close, _weights

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TDT = {"f32": torch.float32, "f64": torch.float64, "f16": torch.float16, "bf16": torch.bfloat16}

# (shape, cut): rows of 28 / 30 / 62 / 110 elements; volumes of 19 - 60 KB
CASES = [((2, 3, 16, 28, 28), None), ((1, 2, 8, 30, 62), None), ((2, 2, 9, 20, 30), None), ((1, 2, 3, 40, 110), None),
         ((1, 2, 10, 30, 62), [[1, 0], [0, 2], [3, 1]]), ((1, 3, 12, 28, 30), [[0, 1], [2, 0], [0, 2]])]


@pytest.fixture(scope="module")
def abi():
    from torchshifts import abi as A
    assert torch.cuda.is_available(), "the gpu tests need an MI355X"
    A.set_path_policy(0)
    return A


def _serves(shape, new, es, backward):
    """plane_ragged_*_eligible's geometry part: 3-D, ragged rows of whole 4-byte groups, beyond the small-plane kernels"""
    rows = shape[-1] if backward else new[-1]
    if (rows * es) % 16 == 0 or (es == 2 and rows % 2):
        return False
    cropped = list(shape[2:]) != list(new[2:])
    return cropped or int(np.prod(shape[2:])) * es > 16 * 1024


@pytest.mark.parametrize("dt", ["f32", "f64", "f16", "bf16"])
@pytest.mark.parametrize("shape,crop", CASES)
def test_ragged_volumes_vs_oracle(abi, shape, crop, dt):
    tdt = TDT[dt]
    b, new = abi.check_borders(list(shape), crop, 3)
    rs = np.random.RandomState(sum(shape) * 5 + 1)
    xt = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt)
    gt = torch.from_numpy(rs.uniform(-1, 1, size=new)).to(tdt)
    wt = torch.from_numpy(_weights(rs, shape[1], 3, shape[2:])).to(tdt)
    wide = np.float64 if tdt == torch.float64 else np.float32
    x, go, w = (t.to(torch.float64).numpy().astype(wide) for t in (xt, gt, wt))
    xd, god, wd = xt.to(DEV), gt.to(DEV), wt.to(DEV)
    es = xt.element_size()
    for pad in range(5):
        out = abi.forward(xd, wd, pad, 1, b)
        if _serves(shape, new, es, False):
            assert abi.last_kernel() == "plane_active_forward_ragged", (shape, crop, dt, abi.last_kernel())
        ref = torch.from_numpy(O.forward(x, w, pad, 1, b)).to(tdt)
        if es >= 4:
            assert torch.equal(out.cpu(), ref), ("fwd", shape, crop, dt, pad)
        else:
            assert _ulp_close(out.cpu(), ref, tdt), ("fwd", shape, crop, dt, pad)
        for active in (0, 1):
            gx, gw = abi.backward(god, wd, xd, pad, active, b)
            if _serves(shape, new, es, True):
                assert abi.last_kernel() == "plane_backward_ragged", (shape, crop, dt, abi.last_kernel())
            gx_ref = torch.from_numpy(O.backward(go, w, x, pad, active, b)[0]).to(tdt)
            if es >= 4 or not active:
                assert torch.equal(gx.cpu(), gx_ref), ("gx", shape, crop, dt, pad, active)
            else:
                assert _ulp_close(gx.cpu(), gx_ref, tdt), ("gx", shape, crop, dt, pad, active)
            _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active, b)
            tol = {"f32": 1e-5, "f64": 1e-12}.get(dt, gw16_tol(torch.finfo(tdt).eps))
            assert rel_err(gw.to(torch.float64).cpu().numpy(), gw64) < tol, ("gw", shape, crop, dt, pad, active)
            gx2, gw2 = abi.backward(god, wd, xd, pad, active, b)
            assert torch.equal(gx, gx2) and torch.equal(gw, gw2)   # deterministic


# round 6 -- what the route census still found on the strided fallback: (a) 3-D rows of an odd number of 16-bit elements, (b) 1-D rows
# of 5 - 28 elements cut by a window (more planes per 4 KiB step than the flat stream's table holds), (c) -- forward only -- 2-D
# windows on rows too long for the row-span kernels.  (shape, cut, dtypes, forward kernel or None = whatever serves, backward ditto;
# the names are asserted for the 16-bit types -- some 4- / 8-byte rows of these shapes are whole pieces and have other kernels)
TAIL = [((2, 2, 40, 16, 31), None, ("f16", "bf16"), "plane_active_forward_ragged", "plane_backward_ragged"),
        ((1, 3, 300, 6, 7), None, ("f16", "bf16"), "plane_active_forward_ragged", "plane_backward_ragged"),
        ((2, 2, 9, 18, 13), [[1, 0], [2, 1], [1, 2]], ("f16", "bf16"), "plane_active_forward_ragged", "plane_backward_ragged"),
        ((1, 2, 300, 40, 1), None, ("bf16",), "plane_active_forward_ragged", "plane_backward_ragged"),
        ((1, 2, 12, 14, 32), [[2, 1], [2, 2], [2, 1]], ("f16",), "plane_active_forward_ragged", None),
        ((64, 16, 12), [[0, 2]], ("f32", "f64", "f16", "bf16"), None, "plane_backward_ragged"),
        ((8, 3, 7), [[2, 0]], ("f32", "f64", "f16", "bf16"), None, "plane_backward_ragged"),
        ((3, 256, 18), [[2, 2]], ("f32", "f16"), None, "plane_backward_ragged"),
        ((2, 1, 5), [[0, 2]], ("f64", "bf16"), None, "plane_backward_ragged"),
        ((2, 2, 6, 5000), [[1, 0], [0, 2]], ("f32", "f64", "bf16"), "plane_active_forward_ragged", None),
        ((2, 3, 6, 5), [[2, 1], [1, 2]], ("f32", "f16"), None, None)]


@pytest.mark.parametrize("shape,crop,dts,kf,kb", TAIL)
def test_fallback_tail_vs_oracle(abi, shape, crop, dts, kf, kb):
    nd = len(shape) - 2
    b, new = abi.check_borders(list(shape), crop, nd)
    for dt in dts:
        tdt = TDT[dt]
        rs = np.random.RandomState(sum(shape) * 3 + len(dt))
        xt = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt)
        gt = torch.from_numpy(rs.uniform(-1, 1, size=new)).to(tdt)
        wt = torch.from_numpy(_weights(rs, shape[1], nd, shape[2:])).to(tdt)
        wide = np.float64 if tdt == torch.float64 else np.float32
        x, go, w = (t.to(torch.float64).numpy().astype(wide) for t in (xt, gt, wt))
        xd, god, wd = xt.to(DEV), gt.to(DEV), wt.to(DEV)
        es = xt.element_size()
        for pad in range(5):
            out = abi.forward(xd, wd, pad, 1, b)
            assert not abi.last_kernel().startswith("strided"), ("fwd", shape, crop, dt, abi.last_kernel())
            assert kf is None or es != 2 or abi.last_kernel() == kf, ("fwd", shape, crop, dt, abi.last_kernel())
            ref = torch.from_numpy(O.forward(x, w, pad, 1, b)).to(tdt)
            assert (torch.equal(out.cpu(), ref) if es >= 4 else _ulp_close(out.cpu(), ref, tdt)), ("fwd", shape, crop, dt, pad)
            for active in (0, 1):
                gx, gw = abi.backward(god, wd, xd, pad, active, b)
                assert not abi.last_kernel().startswith("strided"), ("bwd", shape, crop, dt, abi.last_kernel())
                assert kb is None or es != 2 or abi.last_kernel() == kb, ("bwd", shape, crop, dt, abi.last_kernel())
                gx_ref = torch.from_numpy(O.backward(go, w, x, pad, active, b)[0]).to(tdt)
                if es >= 4 or not active:
                    assert torch.equal(gx.cpu(), gx_ref), ("gx", shape, crop, dt, pad, active)
                else:
                    assert _ulp_close(gx.cpu(), gx_ref, tdt), ("gx", shape, crop, dt, pad, active)
                _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active, b)
                tol = {"f32": 1e-5, "f64": 1e-12}.get(dt, gw16_tol(torch.finfo(tdt).eps))
                assert rel_err(gw.to(torch.float64).cpu().numpy(), gw64) < tol, ("gw", shape, crop, dt, pad, active)
                gx2, gw2 = abi.backward(god, wd, xd, pad, active, b)
                assert torch.equal(gx, gx2) and torch.equal(gw, gw2)


def _offset_copy(t, elems):
    """the same values at a storage offset of `elems` elements (a view into a larger buffer): element-aligned, not 16-byte aligned"""
    big = torch.empty(t.numel() + 16, dtype=t.dtype, device=t.device)
    v = big[elems:elems + t.numel()].view(t.shape)
    v.copy_(t)
    assert v.data_ptr() % 16 != 0 and v.is_contiguous()
    return v


@pytest.mark.parametrize("shape,crop", [((8, 16, 14, 14), None), ((4, 8, 7, 7), None), ((2, 4, 6, 8, 8), None), ((4, 8, 64), None),
                                        ((2, 3, 16, 24), [[1, 0], [2, 1]]), ((2, 2, 5, 6, 16), [[1, 1], [0, 1], [2, 2]])])
def test_element_aligned_storage_offsets(abi, shape, crop):
    """tensors at an element-aligned storage offset (every 16-byte-piece family declines them) ran the one-thread-per-element
    strided kernels; they take the direct-load plane kernels with element-wide chunks now.  Against the oracle, every padding, both shifts."""
    nd = len(shape) - 2
    b, new = abi.check_borders(list(shape), crop, nd)
    for dt in ("f32", "f64", "f16", "bf16"):
        tdt = TDT[dt]
        rs = np.random.RandomState(sum(shape) * 7 + len(dt))
        xt = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt)
        gt = torch.from_numpy(rs.uniform(-1, 1, size=new)).to(tdt)
        wt = torch.from_numpy(_weights(rs, shape[1], nd, shape[2:])).to(tdt)
        wide = np.float64 if tdt == torch.float64 else np.float32
        x, go, w = (t.to(torch.float64).numpy().astype(wide) for t in (xt, gt, wt))
        es = xt.element_size()
        xd, god, wd = _offset_copy(xt.to(DEV), 1), _offset_copy(gt.to(DEV), 3 if es < 8 else 1), wt.to(DEV)
        for pad in range(5):
            outb = _offset_copy(torch.zeros(new, dtype=tdt, device=DEV), 1)
            out = abi.forward(xd, wd, pad, 1, b, out=outb)
            assert not abi.last_kernel().startswith("strided"), ("fwd", shape, crop, dt, abi.last_kernel())
            ref = torch.from_numpy(O.forward(x, w, pad, 1, b)).to(tdt)
            assert (torch.equal(out.cpu(), ref) if es >= 4 else _ulp_close(out.cpu(), ref, tdt)), ("fwd", shape, crop, dt, pad)
            for active in (0, 1):
                gxb = _offset_copy(torch.zeros(shape, dtype=tdt, device=DEV), 1)
                gx, gw = abi.backward(god, wd, xd, pad, active, b, grad_x=gxb)
                assert not abi.last_kernel().startswith("strided"), ("bwd", shape, crop, dt, abi.last_kernel())   # (band_plane_ / plane_backward_ragged)
                gx_ref = torch.from_numpy(O.backward(go, w, x, pad, active, b)[0]).to(tdt)
                if es >= 4 or not active:
                    assert torch.equal(gx.cpu(), gx_ref), ("gx", shape, crop, dt, pad, active)
                else:
                    assert _ulp_close(gx.cpu(), gx_ref, tdt), ("gx", shape, crop, dt, pad, active)
                _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active, b)
                tol = {"f32": 1e-5, "f64": 1e-12}.get(dt, gw16_tol(torch.finfo(tdt).eps))
                assert rel_err(gw.to(torch.float64).cpu().numpy(), gw64) < tol, ("gw", shape, crop, dt, pad, active)


def test_ragged_volume_at_size(abi):
    """N8 C128 16 x 28 x 28 bf16 (a video network's 28 x 28 stage): linearity of the backward in the incoming gradient and the
    forward's values (1 ulp) against the strided fallback (policy 1), which the oracle-sized cases above pin"""
    torch.manual_seed(5)
    shape = (8, 128, 16, 28, 28)
    x = torch.rand(shape, device=DEV).bfloat16()
    go = torch.rand(shape, device=DEV).bfloat16()
    w = ((torch.rand(shape[1], 3, device=DEV) * 2 - 1) * 1.5).bfloat16()
    for pad in (0, 2, 4):
        out = abi.forward(x, w, pad, 1)
        assert abi.last_kernel() == "plane_active_forward_ragged"
        gx, gw = abi.backward(go, w, x, pad, 1)
        assert abi.last_kernel() == "plane_backward_ragged"
        abi.set_path_policy(1)
        try:
            out_s = abi.forward(x, w, pad, 1)
            gx_s, gw_s = abi.backward(go, w, x, pad, 1)
        finally:
            abi.set_path_policy(0)
        # (two 16-bit kernel families: 1 ulp -- the compiler may fuse the last blend and the rounding in one of them)
        assert _ulp_close(out.cpu(), out_s.cpu(), torch.bfloat16) and _ulp_close(gx.cpu(), gx_s.cpu(), torch.bfloat16), pad
        assert (gw.float() - gw_s.float()).abs().max().item() <= gw16_tol(torch.finfo(torch.bfloat16).eps) * max(1.0, gw_s.float().abs().max().item())


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("shape", [(2, 2, 5, 16384), (1, 2, 3, 4, 12400)])
def test_16bit_interpolating_forward_of_very_long_rows(abi, shape, dt):
    """2-D / 3-D 16-bit tensors whose rows are beyond the LDS kernels' reach (more than 12 288 map entries): the sweep-shaped interpolating
    forward instead of the strided fallback (route census, round 5); within 1 ulp of the oracle, every padding"""
    tdt = TDT[dt]
    nd = len(shape) - 2
    rs = np.random.RandomState(sum(shape))
    xt = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt)
    wt = torch.from_numpy(_weights(rs, shape[1], nd, shape[2:])).to(tdt)
    x, w = xt.float().numpy(), wt.float().numpy()
    xd, wd = xt.to(DEV), wt.to(DEV)
    for pad in range(5):
        out = abi.forward(xd, wd, pad, 1)
        assert abi.last_kernel() == "sweep_active_forward", (shape, dt, abi.last_kernel())
        assert _ulp_close(out.cpu(), torch.from_numpy(O.forward(x, w, pad, 1)).to(tdt), tdt), (shape, dt, pad)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("shape,crop", [((2, 3, 9, 24), None), ((2, 2, 5, 6, 16), None), ((1, 2, 4, 7, 64), [[1, 0], [0, 2], [8, 8]]),
                                        ((2, 2, 33, 40), [[2, 1], [0, 8]])])
def test_16bit_sweep_forward_under_policy(abi, shape, crop, dt):
    """the same kernel on everyday shapes (policy 3 = the sweep kernels or fail): windows, small planes, every padding"""
    tdt = TDT[dt]
    nd = len(shape) - 2
    b, new = abi.check_borders(list(shape), crop, nd)
    rs = np.random.RandomState(sum(shape) + 5)
    xt = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt)
    wt = torch.from_numpy(_weights(rs, shape[1], nd, shape[2:])).to(tdt)
    x, w = xt.float().numpy(), wt.float().numpy()
    abi.set_path_policy(3)
    try:
        for pad in range(5):
            out = abi.forward(xt.to(DEV), wt.to(DEV), pad, 1, b)
            assert abi.last_kernel() == "sweep_active_forward", (shape, dt, abi.last_kernel())
            assert _ulp_close(out.cpu(), torch.from_numpy(O.forward(x, w, pad, 1, b)).to(tdt), tdt), (shape, crop, dt, pad)
    finally:
        abi.set_path_policy(0)


@pytest.mark.parametrize("shape,crop,dt", [((2, 2, 40, 16, 31), None, "bf16"), ((2, 2, 9, 18, 13), [[1, 0], [2, 1], [1, 2]], "f16"),
                                           ((8, 3, 7), [[2, 0]], "f32"), ((64, 16, 12), [[0, 2]], "bf16"), ((2, 2, 6, 5000), [[1, 0], [0, 2]], "f32")])
def test_element_wide_kernels_stay_inside_their_tensors(abi, shape, crop, dt):
    """the element-wide plane kernels behind the fallback tail: sentinel bytes around `out` and `grad_x` survive"""
    tdt = TDT[dt]
    nd = len(shape) - 2
    b, new = abi.check_borders(list(shape), crop, nd)
    rs = np.random.RandomState(sum(shape) + 9)
    xd = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt).to(DEV)
    gd = torch.from_numpy(rs.uniform(-1, 1, size=new)).to(tdt).to(DEV)
    wd = torch.from_numpy(_weights(rs, shape[1], nd, shape[2:])).to(tdt).to(DEV)
    es = xd.element_size()

    def guarded(shp):
        n, padn = int(np.prod(shp)), 512 // es
        big = torch.full((n + 2 * padn,), 7.0, dtype=tdt, device=DEV)
        v = big[padn:padn + n].view(shp)
        v.zero_()
        return big, v, padn
    for pad in (0, 2, 3):
        for active in (0, 1):
            big_o, out, po = guarded(new)
            abi.forward(xd, wd, pad, active, b, out=out)
            torch.cuda.synchronize()
            assert bool((big_o[:po] == 7).all()) and bool((big_o[po + out.numel():] == 7).all()), ("fwd", shape, crop, dt, pad, active, abi.last_kernel())
            big_g, gx, pg = guarded(shape)
            abi.backward(gd, wd, xd, pad, active, b, grad_x=gx)
            torch.cuda.synchronize()
            assert bool((big_g[:pg] == 7).all()) and bool((big_g[pg + gx.numel():] == 7).all()), ("bwd", shape, crop, dt, pad, active, abi.last_kernel())
