"""Seeded random sweep over geometries, layouts, paddings, crops, shift magnitudes, kernel-family policies and the
fused-pool entry points: every result against the CPU oracle (forward / grad_x bit-exact in fp32 / fp64, grad_w
<= 1e-5 / 1e-12 relative to the fp64 oracle).  Sizes are small (the oracle runs in milliseconds); the point is the
number of distinct code paths: row bands, several planes per workgroup, LDS-staged and direct kernels, one / two LDS
tiles, ragged rows (strided fallback), channels-last kernels, empty crops."""
import numpy as np
import pytest
import torch

from cases import rel_err
from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def abi():
    from torchshifts import abi as A
    assert torch.cuda.is_available(), "the gpu tests need an MI355X"
    yield A
    A.set_path_policy(0)
    for k, v in ((0, 0), (1, 128 * 1024), (3, 2), (4, 1), (7, 0)):
        A.set_tuning(k, v)


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _case(rs):
    nd = int(rs.randint(1, 4))
    N, C = int(rs.randint(1, 4)), int(rs.randint(1, 7))
    inner = int(rs.choice([4, 8, 12, 16, 20, 24, 28, 36, 5, 7, 10]))
    outer = [int(rs.randint(1, 12)) for _ in range(nd - 1)]
    shape = (N, C) + tuple(outer) + (inner,)
    crop = None
    if rs.rand() < 0.3 and min(shape[2:]) >= 5:
        crop = [[int(rs.randint(0, 3)), int(rs.randint(0, 3))] for _ in range(nd)]
    scale = float(rs.choice([1.5, 4.0, 12.0, 60.0]))
    w = rs.uniform(-scale, scale, size=(C, nd))
    if rs.rand() < 0.5:
        w[rs.randint(0, C)] = np.round(w[rs.randint(0, C)]) + rs.choice([0.0, 0.5, -0.5])
    return nd, shape, crop, w


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_float(abi, seed):
    rs = np.random.RandomState(1000 + seed)
    checked = 0
    for it in range(150):
        nd, shape, crop, w64 = _case(rs)
        dt = np.float32 if rs.rand() < 0.7 else np.float64
        x = rs.uniform(-1, 1, size=shape).astype(dt)
        w = w64.astype(dt)
        b, new = abi.check_borders(list(shape), crop, nd)
        go = rs.uniform(-1, 1, size=new).astype(dt)
        pad, active = int(rs.randint(0, 5)), int(rs.randint(0, 2))
        key = (seed, it, shape, crop, pad, active, dt.__name__)
        ref = O.forward(x, w, pad, active, b)
        gx_o, _ = O.backward(go, w, x, pad, active, b)
        _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active, b)
        tol = 1e-12 if dt == np.float64 else 1e-5
        xd, wd, god = _dev(x), _dev(w), _dev(go)
        # launch-planning knobs: force row bands / many planes per workgroup / one or two LDS tiles / direct kernels
        abi.set_tuning(0, int(rs.choice([1, 64, 2048, 100000])))
        abi.set_tuning(7, int(rs.choice([1, 2048, 65536, 1000000])))
        abi.set_tuning(1, int(rs.choice([1, 4096, 128 * 1024])))
        abi.set_tuning(3, int(rs.choice([1, 2])))
        abi.set_tuning(4, int(rs.choice([1, 2, 3])))
        for policy in (0, 1, 2, 3):
            abi.set_path_policy(policy)
            try:
                out = abi.forward(xd, wd, pad, active, b)
            except RuntimeError:
                assert policy in (2, 3)  # that family does not serve this geometry
                out = None
            if out is not None:
                assert np.array_equal(out.cpu().numpy(), ref), ("fwd", policy) + key
            try:
                gx, gw = abi.backward(god, wd, xd, pad, active, b)
            except RuntimeError:
                assert policy in (2, 3)
                continue
            assert np.array_equal(gx.cpu().numpy(), gx_o), ("gx", policy) + key
            assert rel_err(gw.cpu().numpy(), gw64) < tol, ("gw", policy) + key
            checked += 1
        abi.set_path_policy(0)
        if nd >= 2 and shape[1] > 1:  # channels-last input, output, gradients
            fmt = torch.channels_last if nd == 2 else torch.channels_last_3d
            xc = xd.contiguous(memory_format=fmt)
            oc = torch.empty(new, dtype=xd.dtype, device=DEV).contiguous(memory_format=fmt)
            abi.forward(xc, wd, pad, active, b, out=oc)
            assert abi.last_path() == abi.PATH_CL and np.array_equal(oc.cpu().numpy(), ref), ("cl fwd",) + key
            gx, gw = abi.backward(god.contiguous(memory_format=fmt), wd, xc, pad, active, b, grad_x=torch.empty_like(xc))
            assert abi.last_path() == abi.PATH_CL and np.array_equal(gx.cpu().numpy(), gx_o), ("cl gx",) + key
            assert rel_err(gw.cpu().numpy(), gw64) < tol, ("cl gw",) + key
        # fused pool tail
        pool = [int(rs.randint(1, 4)) for _ in range(nd)]
        refp = O.forward_pooled(x, w, pad, active, pool, b)
        outp = abi.forward_pooled(xd, wd, pad, active, pool, b)
        assert np.array_equal(outp.cpu().numpy(), refp), ("pool fwd", pool) + key
        gp = rs.uniform(-1, 1, size=refp.shape).astype(dt)
        gxp_o, _ = O.backward_pooled(gp, w, x, pad, active, pool, b)
        _, gwp64 = O.backward_pooled(gp.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active, pool, b)
        if (shape[-1] * x.itemsize) % 16 == 0:  # the fused backward serves whole 16-byte rows
            abi.set_path_policy(2)
            gxp, gwp = abi.backward_pooled(_dev(gp), wd, xd, pad, active, pool, b)
            abi.set_path_policy(0)
            assert np.array_equal(gxp.cpu().numpy(), gxp_o), ("pool gx", pool) + key
            assert rel_err(gwp.cpu().numpy(), gwp64) < tol, ("pool gw", pool) + key
    for k, v in ((0, 0), (1, 128 * 1024), (3, 2), (4, 1), (7, 0)):
        abi.set_tuning(k, v)
    assert checked >= 300  # 150 cases x (automatic + strided at least)


@pytest.mark.parametrize("seed", range(2))
def test_fuzz_quantized(abi, seed):
    rs = np.random.RandomState(2000 + seed)
    tdt = {np.uint8: torch.uint8, np.int8: torch.int8, np.int32: torch.int32}
    for it in range(200):
        nd, shape, crop, w64 = _case(rs)
        npdt = [np.uint8, np.int8, np.int32][int(rs.randint(0, 3))]
        info = np.iinfo(npdt)
        lo, hi = max(info.min, -1000), min(info.max, 1000)
        xq = rs.randint(lo, hi + 1, size=shape).astype(npdt)
        xzp = int(rs.randint(lo, hi + 1))
        wzp = 128
        wq = np.clip(np.rint(w64) + wzp, 0, 255).astype(np.uint8)
        b, new = abi.check_borders(list(shape), crop, nd)
        pad = int(rs.randint(0, 5))
        ref = O.forward_q(xq, wq, wzp, xzp, pad, b)
        x = torch.from_numpy(xq).to(DEV)
        w = torch.from_numpy(wq).to(DEV)
        abi.set_tuning(0, int(rs.choice([1, 2048, 100000])))
        abi.set_tuning(1, int(rs.choice([1, 4096, 128 * 1024])))
        for policy in (0, 1, 2, 3):
            abi.set_path_policy(policy)
            try:
                out = abi.forward_quantized(x, w, wzp, xzp, pad, b)
            except RuntimeError:
                assert policy in (2, 3)  # that family does not serve this geometry
                continue
            assert np.array_equal(out.cpu().numpy(), ref), (seed, it, policy, shape, crop, pad, npdt.__name__)
        abi.set_path_policy(0)
        if nd >= 2 and shape[1] > 1:
            fmt = torch.channels_last if nd == 2 else torch.channels_last_3d
            oc = torch.empty(new, dtype=tdt[npdt], device=DEV).contiguous(memory_format=fmt)
            abi.forward_quantized(x.contiguous(memory_format=fmt), w, wzp, xzp, pad, b, out=oc)
            assert abi.last_path() == abi.PATH_CL and np.array_equal(oc.cpu().numpy(), ref)
    abi.set_tuning(0, 0)
    abi.set_tuning(1, 128 * 1024)
