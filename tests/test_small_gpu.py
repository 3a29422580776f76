"""Small-plane kernels (csrc/shiftnd_small.hip): contiguous problems whose rows are not whole 16-byte pieces.  Since round 5 the
1-D / 2-D shapes of this file run the flat-stream kernels (csrc/shiftnd_flat.hip; their own file: test_flat_gpu.py) by default: the
small-plane kernels keep the 3-D volumes, the row-band kernels what the flat-stream kernels do not take; the band tests turn the
flat-stream kernels off (knob 27 = 1) to measure them."""
import numpy as np
import pytest
import torch

from cases import rel_err, gw16_tol
from oracle import oracle as O
from test_hip_parity import _ulp_close, _weights
from test_flat_gpu import flat_serves

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [(3, 5, 7, 7), (4, 6, 14, 14), (2, 3, 13, 5), (5, 4, 1, 9), (3, 4, 13), (2, 3, 4, 5, 6), (2, 2, 3, 7, 7), (9, 3, 1, 1, 3)]


@pytest.fixture(scope="module")
def abi():
    from torchshifts import abi as A
    yield A
    for k in (24, 25, 26):
        A.set_tuning(k, 1 if k == 24 else 0)
    A.set_tuning(27, 0)
    A.set_path_policy(0)


def _fwd_name(nd, shape=None, es=4):   # who serves an interpolating forward with ragged rows
    if nd == 3:
        return "small_plane_forward"
    return "flat_active_forward" if shape is None or flat_serves(shape, shape, es, False) else "band_plane_forward"


def _bwd_name(nd, shape=None, es=4):
    if nd == 3:
        return "small_plane_backward"
    return "flat_backward" if shape is None or flat_serves(shape, shape, es, True) else "band_plane_backward"


@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("shape", SHAPES)
def test_small_planes_vs_oracle(abi, shape, dt):
    """interpolating forward, sparse / active backward, every padding, shifts beyond the dims; several rounds per
    workgroup and ragged last rounds (knobs 25 / 26); forward and grad_x bit-exact, grad_w vs the fp64 oracle"""
    npdt = np.float32 if dt == "f32" else np.float64
    rs = np.random.RandomState(sum(shape) * 7 + len(shape))
    nd = len(shape) - 2
    x = rs.uniform(-1, 1, size=shape).astype(npdt)
    go = rs.uniform(-1, 1, size=shape).astype(npdt)
    w = _weights(rs, shape[1], nd, shape[2:]).astype(npdt)
    xd, god, wd = torch.from_numpy(x).to(DEV), torch.from_numpy(go).to(DEV), torch.from_numpy(w).to(DEV)
    ragged = (shape[-1] * x.itemsize) % 16 != 0   # (whole 16-byte rows go to the row-chunk kernels)
    for ppr, rpw in ((0, 0), (1, 2), (2, 3)):
        abi.set_tuning(25, ppr)
        abi.set_tuning(26, rpw)
        for pad in range(5):
            out = abi.forward(xd, wd, pad, 1)
            assert (abi.last_kernel() == _fwd_name(nd, shape, x.itemsize)) == ragged, (shape, pad, abi.last_kernel())
            assert np.array_equal(out.cpu().numpy(), O.forward(x, w, pad, 1)), (shape, pad, ppr, rpw)
            for active in (0, 1):
                gx, gw = abi.backward(god, wd, xd, pad, active)
                assert (abi.last_kernel() == _bwd_name(nd, shape, x.itemsize)) == ragged, (shape, pad, active, abi.last_kernel())
                gx_o, _ = O.backward(go, w, x, pad, active)
                _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active)
                assert np.array_equal(gx.cpu().numpy(), gx_o), (shape, pad, active, ppr, rpw)
                assert rel_err(gw.cpu().numpy(), gw64) < (1e-12 if dt == "f64" else 1e-5), (shape, pad, active)
    abi.set_tuning(25, 0)
    abi.set_tuning(26, 0)


@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16])
def test_small_planes_16bit_agree_with_the_strided_kernels(abi, tdt):
    """fp16 / bf16: within 1 ulp of the oracle (bit-exact for the sparse shift's raw copies), and the same bits as the
    one-thread-per-element kernels (one definition of the arithmetic)"""
    torch.manual_seed(3)
    for shape in [(6, 8, 14, 14), (5, 7, 7, 7), (3, 4, 3, 5, 6)]:
        nd = len(shape) - 2
        x = torch.rand(shape, device=DEV).to(tdt)
        go = torch.rand(shape, device=DEV).to(tdt)
        w = ((torch.rand(shape[1], nd, device=DEV) - 0.5) * 7).to(tdt)
        for pad in (0, 2, 3):
            for active in (0, 1):
                abi.set_path_policy(1)
                ref = abi.forward(x, w, pad, 1)
                gx_r, gw_r = abi.backward(go, w, x, pad, active)
                abi.set_path_policy(0)
                # (16-bit interpolation: the kernel families agree to 1 ulp of the 16-bit type -- the compiler may fuse
                # a widening into one family's multiply-add -- the sparse shift's grad_x is a raw copy: equal)
                out = abi.forward(x, w, pad, 1)
                assert abi.last_kernel() == _fwd_name(nd) and _ulp_close(out.cpu(), ref.cpu(), tdt)
                gx, gw = abi.backward(go, w, x, pad, active)
                assert abi.last_kernel() == _bwd_name(nd)
                assert _ulp_close(gx.cpu(), gx_r.cpu(), tdt) if active else torch.equal(gx, gx_r)
                assert rel_err(gw.float().cpu().numpy(), gw_r.float().cpu().numpy()) < 2 * gw16_tol(torch.finfo(tdt).eps)   # (two roundings)
                # ... and against the oracle itself (widened inputs, one rounding), not only against a sibling kernel
                xn, gn, wn = x.float().cpu().numpy(), go.float().cpu().numpy(), w.float().cpu().numpy()
                assert _ulp_close(out.cpu(), torch.from_numpy(O.forward(xn, wn, pad, 1)).to(tdt), tdt), (shape, pad)
                gx_o = torch.from_numpy(O.backward(gn, wn, xn, pad, active)[0]).to(tdt)
                assert _ulp_close(gx.cpu(), gx_o, tdt) if active else torch.equal(gx.cpu(), gx_o), (shape, pad, active)
                _, gw64 = O.backward(gn.astype(np.float64), wn.astype(np.float64), xn.astype(np.float64), pad, active)
                assert rel_err(gw.float().cpu().numpy(), gw64) < gw16_tol(torch.finfo(tdt).eps), (shape, pad, active)


def test_small_planes_full_batch(abi):
    """ResNet-stage shapes at full batch against the one-thread-per-element kernels"""
    torch.manual_seed(9)
    for shape in [(128, 512, 14, 14), (128, 1024, 7, 7)]:
        x = torch.rand(shape, device=DEV)
        go = torch.rand(shape, device=DEV)
        w = (torch.rand(shape[1], 2, device=DEV) - 0.5) * 6
        for active in (0, 1):
            abi.set_path_policy(1)
            ref = abi.forward(x, w, 0, 1)
            gx_r, gw_r = abi.backward(go, w, x, 0, active)
            abi.set_path_policy(0)
            out = abi.forward(x, w, 0, 1)
            gx, gw = abi.backward(go, w, x, 0, active)
            assert abi.last_kernel() == "flat_backward"
            assert torch.equal(out, ref) and torch.equal(gx, gx_r)
            assert rel_err(gw.cpu().numpy(), gw_r.cpu().numpy()) < 1e-5


BAND_SHAPES = [(2, 3, 40, 113), (2, 2, 300, 25), (2, 3, 2301), (1, 2, 37, 131)]


@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("shape", BAND_SHAPES)
def test_row_bands_vs_oracle(abi, shape, dt):
    """band_plane_forward / band_plane_backward: 1-D / 2-D planes that do not fit LDS, rows that are not whole 16-byte
    pieces; every padding, shifts beyond the dims, ragged last bands and one-row bands (knob 25); forward and grad_x
    bit-exact, grad_w vs the fp64 oracle"""
    npdt = np.float32 if dt == "f32" else np.float64
    if len(shape) == 3:  # 1-D: one row is the whole plane -- longer than the small-plane limit, short enough for LDS
        shape = (shape[0], shape[1], 4501 if dt == "f32" else 2201)
    rs = np.random.RandomState(sum(shape) * 5 + len(shape))
    nd = len(shape) - 2
    x = rs.uniform(-1, 1, size=shape).astype(npdt)
    go = rs.uniform(-1, 1, size=shape).astype(npdt)
    w = _weights(rs, shape[1], nd, shape[2:]).astype(npdt)
    xd, god, wd = torch.from_numpy(x).to(DEV), torch.from_numpy(go).to(DEV), torch.from_numpy(w).to(DEV)
    ragged = (shape[-1] * x.itemsize) % 16 != 0
    assert ragged
    for k in (27, 32, 34):   # (the flat-stream and the row-relative kernels would take most of these shapes: off, the band kernels are measured)
        abi.set_tuning(k, 1)
    for br in (0, 1, 3):
        abi.set_tuning(25, br)
        for pad in range(5):
            out = abi.forward(xd, wd, pad, 1)
            assert abi.last_kernel() == "band_plane_forward", (shape, pad)
            assert np.array_equal(out.cpu().numpy(), O.forward(x, w, pad, 1)), (shape, pad, br)
            for active in (0, 1):
                gx, gw = abi.backward(god, wd, xd, pad, active)
                # (a 1-D fp32 row longer than the small-plane limit does not fit LDS twice with its maps: one element per thread, read
                #  straight from memory -- round 6: the direct-load plane kernel with element-wide chunks, no longer the strided fallback)
                assert abi.last_kernel() == ("plane_backward_ragged" if (nd == 1 and dt == "f32") else "band_plane_backward"), (shape, pad, active)
                gx_o, _ = O.backward(go, w, x, pad, active)
                _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active)
                assert np.array_equal(gx.cpu().numpy(), gx_o), (shape, pad, active, br)
                assert rel_err(gw.cpu().numpy(), gw64) < (1e-12 if dt == "f64" else 1e-5), (shape, pad, active)
    abi.set_tuning(25, 0)
    for k in (27, 32, 34):
        abi.set_tuning(k, 0)


def test_row_bands_16bit_and_full_size(abi):
    """bf16 against the oracle and the one-thread-per-element kernels, and an odd-sized image batch at full size (oracle on
    its first samples)"""
    torch.manual_seed(11)
    for k in (27, 32, 34):
        abi.set_tuning(k, 1)
    for shape, tdt in [((3, 8, 60, 151), torch.bfloat16), ((8, 64, 225, 225), torch.float32)]:
        x = torch.rand(shape, device=DEV).to(tdt)
        go = torch.rand(shape, device=DEV).to(tdt)
        w = ((torch.rand(shape[1], 2, device=DEV) - 0.5) * 7).to(tdt)
        for pad, active in ((0, 0), (3, 1), (2, 1)):
            abi.set_path_policy(1)
            ref = abi.forward(x, w, pad, 1)
            gx_r, gw_r = abi.backward(go, w, x, pad, active)
            abi.set_path_policy(0)
            out = abi.forward(x, w, pad, 1)
            assert abi.last_kernel() == "band_plane_forward" and torch.equal(out, ref)
            gx, gw = abi.backward(go, w, x, pad, active)
            assert abi.last_kernel() == "band_plane_backward" and torch.equal(gx, gx_r)
            tol = 1e-5 if tdt == torch.float32 else 4 * float(torch.finfo(tdt).eps)
            assert rel_err(gw.float().cpu().numpy(), gw_r.float().cpu().numpy()) < tol
            # the oracle itself on the first samples (all of the small bf16 case): 1 ulp of the 16-bit type / bit-exact
            ns = 2 if shape[0] > 3 else shape[0]
            xn, gn, wn = x[:ns].float().cpu().numpy(), go[:ns].float().cpu().numpy(), w.float().cpu().numpy()
            ref_o = torch.from_numpy(O.forward(xn, wn, pad, 1)).to(tdt)
            gx_o = torch.from_numpy(O.backward(gn, wn, xn, pad, active)[0]).to(tdt)
            if tdt == torch.float32:
                assert torch.equal(out[:ns].cpu(), ref_o) and torch.equal(gx[:ns].cpu(), gx_o), (shape, pad, active)
            else:
                assert _ulp_close(out[:ns].cpu(), ref_o, tdt), (shape, pad)
                assert _ulp_close(gx[:ns].cpu(), gx_o, tdt) if active else torch.equal(gx[:ns].cpu(), gx_o), (shape, pad)
                _, gw64 = O.backward(gn.astype(np.float64), wn.astype(np.float64), xn.astype(np.float64), pad, active)
                assert rel_err(gw.float().cpu().numpy(), gw64) < gw16_tol(torch.finfo(tdt).eps), (shape, pad, active)
    for k in (27, 32, 34):
        abi.set_tuning(k, 0)


def test_row_band_gather_forward_vs_oracle(abi):
    """band_gather_forward: sparse-shift / quantized forward of planes above 16 KiB whose rows are not whole 16-byte
    pieces, every element size, every padding; bit-exact"""
    for k in (27, 34):   # (the flat-stream / row-relative kernels take the float cases by default: off here)
        abi.set_tuning(k, 1)
    rs = np.random.RandomState(17)
    for shape in [(2, 3, 70, 113), (1, 2, 300, 25), (2, 2, 4501)]:
        nd = len(shape) - 2
        for npdt in (np.float32, np.float64):
            if shape[-1] * shape[-2 if nd == 2 else -1] * np.dtype(npdt).itemsize <= 16 * 1024 and nd == 2:
                continue
            x = rs.uniform(-1, 1, size=shape).astype(npdt)
            w = _weights(rs, shape[1], nd, shape[2:]).astype(npdt)
            xd, wd = torch.from_numpy(x).to(DEV), torch.from_numpy(w).to(DEV)
            for br in (0, 1, 5):
                abi.set_tuning(25, br)
                for pad in range(5):
                    out = abi.forward(xd, wd, pad, 0)
                    assert abi.last_kernel() == "band_gather_forward", (shape, npdt, pad)
                    assert np.array_equal(out.cpu().numpy(), O.forward(x, w, pad, 0)), (shape, npdt, pad, br)
            abi.set_tuning(25, 0)
        # quantized uint8 / int32, fp16 (raw copies): uint8 planes above 16 KiB need more elements
    for shape, npdt in [((2, 3, 150, 131), np.uint8), ((1, 2, 90, 57), np.int32)]:
        xq = rs.randint(0, 200, size=shape).astype(npdt)
        wq = rs.randint(100, 160, size=(shape[1], 2)).astype(np.uint8)
        xqd, wqd = torch.from_numpy(xq).to(DEV), torch.from_numpy(wq).to(DEV)
        for pad in range(5):
            out = abi.forward_quantized(xqd, wqd, 128, 5, pad)
            assert abi.last_kernel() == "band_gather_forward", (shape, pad)
            assert np.array_equal(out.cpu().numpy(), O.forward_q(xq, wq, 128, 5, pad)), (shape, pad)
    xh = torch.rand(2, 3, 100, 131, device=DEV).half()
    wh = ((torch.rand(3, 2, device=DEV) - 0.5) * 9).half()
    abi.set_path_policy(1)
    ref = abi.forward(xh, wh, 3, 0)
    abi.set_path_policy(0)
    out = abi.forward(xh, wh, 3, 0)
    assert abi.last_kernel() == "band_gather_forward" and torch.equal(out, ref)

    for k in (27, 32, 34):
        abi.set_tuning(k, 0)

@pytest.mark.parametrize("shape,npdt", [((2, 3, 40, 64), np.uint8), ((1, 2, 3, 33, 48), np.int8), ((3, 2, 150, 224), np.uint8),
                                        ((2, 2, 1, 4096), np.uint8), ((2, 3, 130), np.uint8)])
def test_rows_gather_forward_vs_oracle(abi, shape, npdt):
    """rows_gather_forward (csrc/shiftnd_rows.hip): quantized forward of 1-byte rows of whole 16-byte pieces beyond the
    byte kernel's small planes; every padding, shifts beyond the dims, ragged last row groups / bands (knob 29); bit-exact"""
    rs = np.random.RandomState(sum(shape) + 1)
    nd = len(shape) - 2
    info = np.iinfo(npdt)
    xq = rs.randint(info.min, info.max + 1, size=shape).astype(npdt)
    wq = rs.randint(120, 137, size=(shape[1], nd)).astype(np.uint8)
    wq[0, :] = min(255, 128 + shape[-1] + 3)
    x, w = torch.from_numpy(xq).to(DEV), torch.from_numpy(wq).to(DEV)
    rows = (shape[-1] % 16 == 0) and int(np.prod(shape[2:])) > 16384
    try:
        abi.set_tuning(16, 0 if not rows else 1)   # (small planes: the byte kernel off, so that this kernel is measured)
        abi.set_tuning(33, 1)                      # (the one-step kernel takes large zero-padded planes first: test_step_gpu.py)
        for br in (0, 7):
            abi.set_tuning(29, br)
            for pad in range(5):
                out = abi.forward_quantized(x, w, 128, -3 if npdt == np.int8 else 9, pad)
                if shape[-1] % 16 == 0:
                    assert abi.last_kernel() == "rows_gather_forward", (shape, pad, abi.last_kernel())
                assert np.array_equal(out.cpu().numpy(), O.forward_q(xq, wq, 128, -3 if npdt == np.int8 else 9, pad)), (shape, pad, br)
    finally:
        abi.set_tuning(16, 1)
        abi.set_tuning(29, 0)
        abi.set_tuning(33, 0)


def test_rows_gather_forward_16bit(abi):
    """the same kernel for fp16 / bf16 sparse shifts (knob 28 bit 1) against the default kernels"""
    torch.manual_seed(2)
    for shape, tdt in [((3, 4, 50, 72), torch.float16), ((2, 3, 4, 20, 40), torch.bfloat16)]:
        nd = len(shape) - 2
        x = torch.rand(shape, device=DEV).to(tdt)
        w = ((torch.rand(shape[1], nd, device=DEV) - 0.5) * 9).to(tdt)
        for pad in range(5):
            ref = abi.forward(x, w, pad, 0)
            abi.set_tuning(28, 3)
            try:
                out = abi.forward(x, w, pad, 0)
                assert abi.last_kernel() == "rows_gather_forward"
            finally:
                abi.set_tuning(28, 1)
            assert torch.equal(out, ref), (shape, pad)


@pytest.mark.parametrize("shape,npdt", [((6, 8, 14, 14), np.uint8), ((5, 32, 7, 7), np.int8), ((37, 16, 7, 7), np.uint8),
                                        ((2, 16, 3, 5, 7), np.uint8), ((4, 16, 13), np.int8), ((3, 4, 14, 14), np.uint8),
                                        ((130, 64, 14, 14), np.uint8), ((2, 8, 30, 30), np.uint8), ((3, 2, 11, 8), np.int8)])
def test_one_byte_ragged_planes_vs_oracle(abi, shape, npdt):
    """bytes_block_forward (csrc/shiftnd_bytes.hip): quantized forward of one-byte planes that are not whole 16-byte pieces
    (14 x 14, 7 x 7: blocks of 16 / gcd(plane bytes, 16) consecutive channels through LDS); every padding, shifts beyond
    the dims, ragged last rounds and batch groups; bit-exact"""
    rs = np.random.RandomState(sum(shape) + 3)
    nd = len(shape) - 2
    info = np.iinfo(npdt)
    xq = rs.randint(info.min, info.max + 1, size=shape).astype(npdt)
    wq = rs.randint(122, 135, size=(shape[1], nd)).astype(np.uint8)
    wq[0, :] = min(255, 128 + shape[-1] + 2)
    wq[1, :] = 128 - 5
    x, w = torch.from_numpy(xq).to(DEV), torch.from_numpy(wq).to(DEV)
    zp = -3 if npdt == np.int8 else 9
    for pad in range(5):
        out = abi.forward_quantized(x, w, 128, zp, pad)
        assert abi.last_kernel() == "bytes_block_forward", (shape, pad, abi.last_kernel())
        assert np.array_equal(out.cpu().numpy(), O.forward_q(xq, wq, 128, zp, pad)), (shape, pad)
