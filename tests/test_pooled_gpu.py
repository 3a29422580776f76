"""GPU parity of the fused shift + average pool entry points (shiftnd_forward_pooled / shiftnd_backward_pooled)
against the oracle's restatement of the module-level sequence the reference runs
(modules/shifts.py:150-153: shift, then avg_pool{N}d(kernel = stride, ceil_mode=True)).

Bars: fp32 / fp64 forward and grad_x bit-exact (same summation order and one IEEE division);
grad_w <= 1e-5 (fp32) / 1e-12 (fp64) relative to the oracle; bf16 / fp16 within 1 ulp of the 16-bit type
against the fp32 oracle on widened inputs.
"""
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
    return A


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


CASES = [  # nd, shape, pool, crop
    (2, (3, 4, 37, 64), (2, 2), None),      # several one-step workgroups per plane, ragged last step, odd rows
    (2, (2, 3, 9, 40), (3, 5), None),       # windows that straddle 16-byte pieces, ragged last window
    (1, (2, 5, 32), (2,), None),
    (1, (3, 4, 36), (3,), [[1, 2]]),
    (2, (2, 6, 12, 16), (2, 2), None),
    (2, (2, 5, 13, 20), (2, 2), None),      # odd rows: ceil-mode partial windows
    (2, (2, 4, 12, 24), (3, 2), [[1, 0], [2, 3]]),
    (2, (1, 3, 9, 8), (4, 4), None),
    (2, (2, 3, 10, 12), (1, 2), None),
    (3, (2, 3, 6, 7, 8), (2, 2, 2), None),
    (3, (1, 4, 5, 6, 12), (2, 3, 2), [[0, 1], [1, 0], [0, 0]]),
    # round 6 -- the module's emulate_dw {kernel 3, stride 2, padding 0}: cut 1 / 1, pool 2 (crop_backward<.., POOL>): even and odd windows
    # (ragged last pooled row / column), several steps per plane, a 3 x 3 pool, a one-sided cut, rows of a single piece
    (2, (2, 4, 20, 32), (2, 2), [[1, 1], [1, 1]]),
    (2, (2, 3, 17, 24), (2, 2), [[1, 1], [1, 1]]),
    (2, (1, 3, 70, 64), (2, 2), [[1, 1], [1, 1]]),
    (2, (1, 3, 16, 40), (3, 3), [[1, 1], [1, 1]]),
    (2, (2, 2, 11, 28), (2, 2), [[0, 2], [3, 0]]),
    (2, (2, 3, 9, 4), (2, 2), [[1, 1], [1, 1]]),
    # cropped 3-D volumes with the 2 x 2 x 2 pool: crop_backward3<.., POOL> (fp32; fp64 composes the pool's backward and crop_backward3)
    (3, (2, 3, 6, 9, 16), (2, 2, 2), [[1, 1], [1, 1], [1, 1]]),
    (3, (1, 4, 5, 8, 12), (2, 2, 2), [[0, 1], [1, 0], [0, 2]]),
]


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_pooled_vs_oracle(abi, dt):
    rs = np.random.RandomState(11)
    for nd, shape, pool, crop in CASES:
        x = rs.uniform(-1, 1, size=shape).astype(dt)
        w = rs.uniform(-3.2, 3.2, size=(shape[1], nd)).astype(dt)
        w[0] = 0.0
        w[1, 0] = 2.5
        b, new = abi.check_borders(list(shape), crop, nd)
        for pad in range(5):
            for active in (0, 1):
                key = (nd, shape, pool, crop, pad, active)
                ref = O.forward_pooled(x, w, pad, active, pool, b)
                xd, wd = _dev(x), _dev(w)
                out = abi.forward_pooled(xd, wd, pad, active, pool, b)
                # (3-D interpolating, windows (K0, K1 <= 2, 2), no crop, rows of whole 16-byte pieces: the walk through the planes)
                fwalk = (nd == 3 and active and crop is None and pool[-1] == 2 and shape[2] >= 2 and (shape[-1] * x.itemsize) % 16 == 0
                         and (pool[-2] == 1 or (pool[-2] == 2 and shape[3] >= 2)))
                # (2-D sparse shift, 2 x 2 windows -- round 6: of any width: the one-step sweep with the pool as its epilogue)
                fstep = nd == 2 and tuple(pool) == (2, 2) and (not active or min(shape[2:]) >= 2)   # (both shifts since round 6: fp32 / fp64)
                # (round 6) 3-D, 2 x 2 x 2 windows, 4-byte elements here, rows of whole pieces, dims of at least 2: pooled rows through LDS
                f3 = (nd == 3 and not fwalk and tuple(pool) == (2, 2, 2) and x.itemsize == 4 and (shape[-1] * x.itemsize) % 16 == 0
                      and min(shape[2:]) >= 2 and min(new[2:]) >= 2)
                assert abi.last_kernel() == ("walk_forward_pool" if fwalk else ("step_gather_forward_pool" if fstep else
                                                                                  ("crop_forward3_pool" if f3 else "plane_pool_forward"))), key + (abi.last_kernel(),)
                assert list(out.shape) == list(ref.shape), key
                assert np.array_equal(out.cpu().numpy(), ref), key
                gp = rs.uniform(-1, 1, size=ref.shape).astype(dt)
                gx_r, gw_r = O.backward_pooled(gp, w, x, pad, active, pool, b)
                # 3-D interpolating: the walk through the planes (2- / 4-byte elements, windows (K0, K1, 2), no crop); what it does
                # not take is not fused by default (the band-walk kernels are slower than the two-step sequence there): forced
                walk = (nd == 3 and dt == np.float32 and crop is None and pool[-1] == 2 and shape[2] >= 2
                        and (shape[-1] * x.itemsize) % 16 == 0)   # (both shifts)
                # (round 6) a cropped 3-D volume crop_backward3 serves is not fused either: avg_pool backward + crop_backward3 is faster
                crop3 = (nd == 3 and crop is not None and (shape[-1] * x.itemsize) % 16 == 0 and min(shape[2:]) >= 2 and min(new[2:]) >= 2)
                # ... unless the windows are 2 x 2 x 2 (4-byte elements here): crop_backward3<.., POOL> fuses the pool
                fused3 = crop3 and tuple(pool) == (2, 2, 2) and x.itemsize == 4 and -(-new[-1] // 2) >= 2
                # (zeros padding, 4-byte elements here, windows (K0, K1, 2) over a crop that begins at most two columns into the rows and
                #  has an even width: the walk through the planes with the window inside and the pool riding on it, walk_backward<.., POOL, .., CROP>)
                cwalk = crop3 and pad == 0 and x.itemsize == 4 and pool[-1] == 2 and crop[2][0] <= 2 and new[-1] % 2 == 0
                if fused3 or cwalk:
                    gx, gw = abi.backward_pooled(_dev(gp), wd, xd, pad, active, pool, b)
                    assert abi.last_kernel() == ("walk_backward_crop_pool" if cwalk else "crop_backward3_pool"), key + (abi.last_kernel(),)
                    assert np.array_equal(gx.cpu().numpy(), gx_r), key
                    assert rel_err(gw.cpu().numpy(), gw_r) < 1e-5, key
                    continue
                if nd == 3 and ((active and not walk) or crop3):
                    with pytest.raises(RuntimeError, match="not served"):
                        abi.backward_pooled(_dev(gp), wd, xd, pad, active, pool, b)
                    abi.set_path_policy(2)
                gx, gw = abi.backward_pooled(_dev(gp), wd, xd, pad, active, pool, b)
                abi.set_path_policy(0)
                step = nd == 2 and crop is None and (shape[-1] * x.itemsize) % 16 == 0 and not active  # the one-step kernel's cases
                # (round 6) 2-D, x rows of whole pieces: what the one-step kernel leaves -- windows, the interpolating shift -- goes to
                # crop_backward<.., POOL> (a window one column wide under zeros padding excepted: the affine column state)
                # 2 x 2 windows, pooled rows of at least half a 16-byte piece; other windows keep the band-walk kernels
                span = (nd == 2 and not step and (shape[-1] * x.itemsize) % 16 == 0 and not (pad == 0 and new[-1] == 1)
                        and tuple(pool) == (2, 2) and -(-new[-1] // 2) >= max(1, 8 // x.itemsize))
                want = ("walk_backward_pool",) if walk else (("step_backward_pool",) if step else (("crop_backward_pool",) if span else
                                                                                                     ("plane_backward_pool", "plane_backward_lds_pool")))
                assert abi.last_kernel() in want, key
                assert np.array_equal(gx.cpu().numpy(), gx_r), key
                assert rel_err(gw.cpu().numpy(), gw_r) < (1e-5 if dt == np.float32 else 1e-12), key


@pytest.mark.parametrize("tdt", [torch.bfloat16, torch.float16])
def test_pooled_16bit(abi, tdt):
    rs = np.random.RandomState(5)
    eps = 2.0 ** -8 if tdt == torch.bfloat16 else 2.0 ** -11
    for nd, shape, pool, crop in [(2, (2, 4, 12, 16), (2, 2), None), (3, (1, 3, 6, 6, 8), (2, 2, 2), None),
                                  (2, (2, 3, 13, 24), (3, 2), [[1, 0], [0, 3]]), (2, (2, 3, 18, 32), (2, 2), [[1, 1], [1, 1]]),
                                  (2, (1, 2, 21, 40), (2, 2), [[1, 1], [1, 1]]), (3, (1, 3, 6, 9, 16), (2, 2, 2), [[1, 1], [1, 1], [1, 1]]),
                                  (3, (2, 2, 5, 7, 24), (2, 2, 2), None), (3, (1, 2, 5, 8, 24), (2, 2, 2), [[1, 0], [0, 1], [2, 2]]),
                                  (3, (2, 2, 4, 20, 32), (2, 2, 2), [[0, 0], [1, 1], [1, 1]])]:
        xt = torch.from_numpy(rs.uniform(-1, 1, size=shape).astype(np.float32)).to(tdt)
        wt = torch.from_numpy(rs.uniform(-2.5, 2.5, size=(shape[1], nd)).astype(np.float32)).to(tdt)
        x, w = xt.float().numpy(), wt.float().numpy()
        b, _ = abi.check_borders(list(shape), crop, nd)
        for pad in (0, 3):
            for active in (0, 1):
                # unfused sequence on widened inputs with the shift output rounded to the storage type
                y = torch.from_numpy(O.forward(x, w, pad, active, b)).to(tdt).float().numpy()
                ref = O.avg_pool(y, pool)
                out = abi.forward_pooled(xt.to(DEV), wt.to(DEV), pad, active, pool, b)
                got = out.float().cpu().numpy()
                assert np.max(np.abs(got - ref)) <= eps * max(1.0, np.max(np.abs(ref))), (shape, pad, active)
                gpt = torch.from_numpy(rs.uniform(-1, 1, size=ref.shape).astype(np.float32)).to(tdt)
                g = torch.from_numpy(O.avg_pool_backward(gpt.float().numpy(), pool, y.shape[2:])).to(tdt).float().numpy()
                gx_r, gw_r = O.backward(g, w, x, pad, active, b)
                for policy in ((0, 2) if (nd == 3 and active) else (0,)):  # 3-D interpolating: the walk (0) and the band-walk kernels (2)
                    abi.set_path_policy(policy)
                    gx, gw = abi.backward_pooled(gpt.to(DEV), wt.to(DEV), xt.to(DEV), pad, active, pool, b)
                    abi.set_path_policy(0)
                    if nd == 3:   # (uncropped: the walk with the pool riding on it; cropped, round 6: crop_backward3<.., POOL>)
                        if crop is not None:   # (zeros padding, policy 0: the cropped walk; else crop_backward3, which the plane family hands cropped volumes to)
                            _, new16 = abi.check_borders(list(shape), crop, nd)
                            cw = pad == 0 and policy == 0 and new16[-1] % 2 == 0 and crop[2][0] <= 2   # (pooled rows at 2-byte boundaries: three dwords per 8 pooled bytes)
                            assert abi.last_kernel() == ("walk_backward_crop_pool" if cw else "crop_backward3_pool"), (shape, policy, abi.last_kernel())
                        else:
                            assert (abi.last_kernel() == "walk_backward_pool") == (policy == 0), (shape, policy, abi.last_kernel())
                    assert np.max(np.abs(gx.float().cpu().numpy() - gx_r)) <= eps * max(1.0, np.max(np.abs(gx_r)))
                    assert rel_err(gw.float().cpu().numpy(), gw_r) < 1.02 * eps + 1e-5   # one rounding (eps = half a unit) + the fp32 oracle's own error


def test_pooled_matches_unfused_fullsize(abi):
    """size-independent property at a large size: fused == shift kernel + torch avg_pool (fp32, bit-exact)"""
    torch.manual_seed(0)
    x = torch.rand(8, 64, 224, 224, device=DEV)
    w = (torch.rand(64, 2, device=DEV) * 6 - 3)
    for pad, active in ((0, 0), (4, 1)):
        y = abi.forward(x, w, pad, active)
        ref = torch.nn.functional.avg_pool2d(y, 2, 2, ceil_mode=True)
        out = abi.forward_pooled(x, w, pad, active, 2)
        assert torch.equal(out, ref)
        gp = torch.rand_like(ref)
        yy = y.clone().requires_grad_(True)
        torch.nn.functional.avg_pool2d(yy, 2, 2, ceil_mode=True).backward(gp)
        gx_r, gw_r = abi.backward(yy.grad, w, x, pad, active)
        gx, gw = abi.backward_pooled(gp, w, x, pad, active, 2)
        assert torch.equal(gx, gx_r)
        assert rel_err(gw.cpu().numpy(), gw_r.cpu().numpy()) < 1e-5


def test_pooled_backward_workspace_small_batch_large_plane():
    """N = 1 with large planes: the pooled backward cuts planes into more row bands than the plain backward plans for,
    so its workspace comes from its own query (shiftnd_backward_pooled_workspace_bytes); through the op and the C ABI"""
    import ctypes
    from torchshifts import abi
    torch.manual_seed(2)
    for shape, tdt in (((1, 8, 224, 224), torch.float32), ((1, 3, 300, 64), torch.float16), ((2, 2, 6, 40, 32), torch.float32)):
        nd = len(shape) - 2
        x = torch.rand(shape, device=DEV).to(tdt)
        w = ((torch.rand(shape[1], nd, device=DEV) - 0.5) * 5).to(tdt)
        for active in (0, 1):
            gp = torch.rand(abi.pooled_shape(x, 2), device=DEV).to(tdt)
            p = abi.problem(x, 0, active, None)
            need = int(abi.lib().shiftnd_backward_pooled_workspace_bytes(ctypes.byref(p), abi._pool_arg(2, nd)))
            plain = int(abi.lib().shiftnd_backward_workspace_bytes(ctypes.byref(p)))
            assert need > 0
            try:
                gx, gw = abi.backward_pooled(gp, w, x, 0, active, 2)
            except RuntimeError as e:  # (what the 3-D walk does not take is not fused: SHIFTND_ERR_NOT_FUSED)
                assert "not served" in str(e) and nd == 3 and active
                continue
            # the two-step reference: avg_pool backward (ATen) then the plain shift backward
            xo = torch.zeros(shape, device=DEV, dtype=tdt, requires_grad=True)
            pool = {2: torch.nn.functional.avg_pool2d, 3: torch.nn.functional.avg_pool3d}[nd]
            pool(xo, 2, 2, 0, True).backward(gp)
            gx_ref, gw_ref = abi.backward(xo.grad, w, x, 0, active)
            assert torch.equal(gx, gx_ref) or (tdt != torch.float32 and (gx.float() - gx_ref.float()).abs().max() < 1e-2)
            assert (gw.float() - gw_ref.float()).abs().max() <= 1e-3 * max(1.0, float(gw_ref.float().abs().max())), (shape, active, need, plain)

QCASES = [  # the planes the per-channel quantized kernel (qpool_plane_forward) stages: nd, shape, pool, crop
    (2, (5, 6, 56, 56), (2, 2), None),      # one plane per round, 16-byte pieces, batch not a multiple of anything
    (2, (11, 3, 28, 28), (2, 2), None),     # several planes per round, a ragged last round
    (2, (9, 4, 14, 14), (2, 2), None),      # dword pieces
    (2, (21, 5, 7, 7), (2, 2), None),       # byte pieces, ragged windows, pooled rows that are not whole dwords
    (2, (3, 3, 56, 56), (3, 3), None),      # nine-byte windows, ragged
    (2, (2, 2, 112, 112), (2, 2), None),    # four items per thread
    (2, (2, 3, 40, 48), (2, 2), [[3, 2], [4, 8]]),
    (3, (3, 2, 8, 28, 28), (2, 2, 2), None),
    (3, (2, 3, 4, 14, 16), (1, 2, 2), None),
    (1, (7, 5, 64), (2,), None),
    (1, (4, 3, 99), (3,), [[2, 1]]),
    # round 4, qpool_band_forward: planes beyond the plane kernel (more than 48 KiB or more than 1024 items), bands of pooled rows
    (2, (3, 6, 224, 224), (2, 2), None),              # 50 KB planes, 16-byte pieces, four bands (six channels: the shifts of BAND_SHIFTS below)
    (2, (2, 6, 150, 226), (2, 2), [[1, 0], [3, 3]]),  # byte / dword pieces, ragged last band and windows, a crop
    (2, (5, 2, 130, 120), (3, 3), None),              # nine-byte windows
    (2, (2, 2, 300, 36), (1, 2), None),               # a row window
    (1, (3, 2, 9000), (2,), None),                    # Shift1d: one long row
    (2, (2, 6, 60, 230), (2, 2), [[0, 0], [3, 2]]),   # a window whose left border is not a multiple of 4 and whose width is odd: the non-fast band kernel
    # round 6, qpool_band_fast<.., 1, PM>: small planes of whole 16-byte pieces whose ROWS are not (56 x 56, 48 x 52), one item per thread
    (2, (3, 6, 48, 52), (2, 2), None),                # six channels: the shifts of BAND_SHIFTS (around the pads' edges, half the plane)
    (2, (3, 2, 56, 56), (2, 2), [[1, 1], [1, 1]]),    # the module's cut 1 / 1
    (2, (2, 3, 24, 56), (1, 2), None),                # a row window
]

# column shifts at and beyond the edge of qpool_band_fast's 16 / 32-byte zero-point pads (+-7, +-8, +-9), and shifts
# that are partially in range (about half the row): (row, column) shifts of channels 1 .. 5 of the six-channel band cases
BAND_SHIFTS = [(-7, 8), (9, -9), (None, None), (8, 7), (-8, -7)]   # (None: half the plane, set per case)


def aten_inv(cnt):
    """1 / scale of ATen's requantization in its average pool: scale = 1 / float(1 / count), all in fp32"""
    mult = np.float32(1.0 / cnt)
    return np.float32(1.0) / (np.float32(1.0) / mult)


@pytest.mark.parametrize("npdt", [np.uint8, np.int8])
def test_quantized_pooled_forward_vs_oracle(abi, npdt):
    """shiftnd_forward_quantized_pooled (csrc/shiftnd_qpool.hip) through the C ABI: the oracle's quantized shift followed by
    ATen's QuantizedCPU average-pool arithmetic restated in numpy, BOTH of ATen's roundings (zero point inside the rounding:
    nearbyint(zp + sum(x - zp) * inv), inv = 1 / (1 / float(1 / count)); outside: nearbyint(sum * float(1 / count)) + zp --
    each pinned against torch's own CPU kernels by tests/test_quant_convert.py::
    test_quantized_avg_pool_restatement_matches_aten); odd zero points so that the two differ; every padding, crops, ragged
    last windows, 1-D / 2-D / 3-D"""
    rs = np.random.RandomState(21)
    info = np.iinfo(npdt)
    differ = False
    served = set()
    for nd, shape, pool, crop in CASES + QCASES:
        xq = rs.randint(info.min, info.max + 1, size=shape).astype(npdt)
        wq = rs.randint(123, 134, size=(shape[1], nd)).astype(np.uint8)
        wq[0] = 128 + shape[-1] + 2 if shape[-1] < 120 else 130
        if nd == 2 and shape[1] == 6 and (shape[-1] >= 200 or shape[-1] == 52):   # the band cases: shifts around the pads' edges and partially in range
            for ch, (sr, sc) in enumerate(BAND_SHIFTS, start=1):
                wq[ch] = [128 + (shape[2] // 2 if sr is None else sr), 128 - (shape[3] // 2 - 14 if sc is None else -sc)]
        zp = 7 if npdt == np.uint8 else -9
        b, new = abi.check_borders(list(shape), crop, nd)
        for pad in range(5):
            y = O.forward_q(xq, wq, 128, zp, pad, b).astype(np.int64) - zp
            psz = [-(-new[2 + r] // pool[r]) for r in range(nd)]
            refs = [np.zeros(list(new[:2]) + psz, dtype=npdt) for _ in range(2)]
            for idx in np.ndindex(*psz):
                sl = tuple(slice(idx[r] * pool[r], min((idx[r] + 1) * pool[r], new[2 + r])) for r in range(nd))
                win = y[(slice(None), slice(None)) + sl]
                cnt = int(np.prod(win.shape[2:]))
                s32 = win.reshape(win.shape[0], win.shape[1], -1).sum(axis=2).astype(np.float32)
                q_in = np.rint(np.float32(zp) + s32 * aten_inv(cnt)).astype(np.int64)
                q_out = np.rint(s32 * np.float32(1.0 / cnt)).astype(np.int64) + zp
                for ref, q in zip(refs, (q_in, q_out)):
                    ref[(slice(None), slice(None)) + idx] = np.clip(q, info.min, info.max).astype(npdt)
            differ = differ or not np.array_equal(refs[0], refs[1])
            for requant, ref in zip((abi.REQUANT_ZP_INSIDE, abi.REQUANT_ZP_OUTSIDE), refs):
                for knob in (0, 1):  # 0: the per-channel plane kernel where it serves, 1: one thread per pooled element
                    abi.set_tuning(36, knob)
                    out = abi.forward_quantized_pooled(torch.from_numpy(xq).to(DEV), torch.from_numpy(wq).to(DEV), 128, zp, pad, pool,
                                                       b, requant=requant)
                    abi.set_tuning(36, 0)
                    assert abi.last_kernel() in (("qpool_forward",) if knob else ("qpool_forward", "qpool_plane_forward", "qpool_band_forward", "qpool_band_fast"))
                    served.add(abi.last_kernel())
                    assert np.array_equal(out.cpu().numpy(), ref), (nd, shape, pool, crop, pad, requant, abi.last_kernel())
    assert differ  # the inputs do separate the two roundings
    assert served == {"qpool_forward", "qpool_plane_forward", "qpool_band_forward", "qpool_band_fast"}


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("shape,pool", [((2, 3, 5, 7, 16), (2, 2, 2)), ((1, 2, 6, 9, 32), (3, 2, 2)), ((1, 2, 4, 40, 112), (1, 2, 2)),
                                        ((2, 2, 9, 33, 64), (2, 3, 2)), ((1, 3, 2, 1, 8), (2, 2, 2))])
def test_pooled_3d_walk_backward(abi, shape, pool, dt):
    """walk_forward<..., POOL> / walk_backward<..., POOL>: the 3-D interpolating shift + average pool in one pass each way (the
    pool as the forward walk's epilogue; the pooled gradient expanded on its way into LDS).  fp32: grad_x bit-exact with the oracle's fused backward, grad_w within 1e-5; 16-bit: against the two-step
    sequence on widened values (the unpooled gradient rounded to the storage type, as ATen's avg_pool backward returns it);
    ragged windows along planes and rows, every padding"""
    tdt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[dt]
    rs = np.random.RandomState(sum(shape) + 41)
    xt = torch.from_numpy(rs.uniform(-1, 1, size=shape).astype(np.float32)).to(tdt)
    if (shape[-1] * xt.element_size()) % 16:
        pytest.skip("rows are not whole 16-byte pieces")
    wt = torch.from_numpy(rs.uniform(-2.6, 2.6, size=(shape[1], 3)).astype(np.float32)).to(tdt)
    wt[0] = torch.tensor([shape[2] + 1.5, -0.25, 9.75]).to(tdt)   # beyond the planes / beyond a piece of columns
    x, w = xt.float().numpy(), wt.float().numpy()
    b, _ = abi.check_borders(list(shape), None, 3)
    eps = {"f32": 0.0, "f16": 2.0 ** -11, "bf16": 2.0 ** -8}[dt]
    for pad in range(5):
        y = O.forward(x, w, pad, 1, b)
        pshape = O.avg_pool(y, pool).shape
        # the forward: walk_forward<..., POOL> for windows (K0, K1 <= 2, 2) (the pool as the walk's epilogue), else the plane kernel
        out = abi.forward_pooled(xt.to(DEV), wt.to(DEV), pad, 1, pool, b)
        fwalk = pool[1] == 1 or (pool[1] == 2 and shape[3] >= 2)   # (a window's two rows live in one workgroup)
        assert abi.last_kernel() == ("walk_forward_pool" if fwalk else "plane_pool_forward"), (shape, pool, abi.last_kernel())
        if dt == "f32":
            assert np.array_equal(out.cpu().numpy(), O.forward_pooled(x, w, pad, 1, pool, b)), ("fwd", shape, pool, pad)
        else:
            ref = O.avg_pool(torch.from_numpy(y).to(tdt).float().numpy(), pool)
            assert np.max(np.abs(out.float().cpu().numpy() - ref)) <= eps * max(1.0, np.max(np.abs(ref))), ("fwd", shape, pool, pad)
        gpt = torch.from_numpy(rs.uniform(-1, 1, size=pshape).astype(np.float32)).to(tdt)
        for active in (1, 0):   # the sparse shift's pooled backward rides the same walk (one gradient tap, copied)
            gx, gw = abi.backward_pooled(gpt.to(DEV), wt.to(DEV), xt.to(DEV), pad, active, pool, b)
            assert abi.last_kernel() == "walk_backward_pool", (shape, pool, abi.last_kernel())
            if dt == "f32":
                gx_r, gw_r = O.backward_pooled(gpt.numpy(), w, x, pad, active, pool, b)
                assert np.array_equal(gx.cpu().numpy(), gx_r), (shape, pool, pad, active)
                _, gw64 = O.backward_pooled(gpt.numpy().astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, active, pool, b)
                assert rel_err(gw.cpu().numpy(), gw64) < 1e-5, (shape, pool, pad, active)
            else:
                g = torch.from_numpy(O.avg_pool_backward(gpt.float().numpy(), pool, y.shape[2:])).to(tdt).float().numpy()
                gx_r, gw_r = O.backward(g, w, x, pad, active, b)
                assert np.max(np.abs(gx.float().cpu().numpy() - gx_r)) <= (eps if active else 0.0) * max(1.0, np.max(np.abs(gx_r))), (shape, pool, pad, active)
                assert rel_err(gw.float().cpu().numpy(), gw_r) < 1.02 * eps + 1e-5, (shape, pool, pad, active)


# round 6 -- Shift1d behind a stride-2 depthwise emulation: row_forward<.., POOL> / row_backward<.., POOL> (rows of at least 128 chunks by
# default; knobs 32 / 34 = 2 send the short rows of these cases there too): whole and ragged last windows, cuts, rows longer than one
# workgroup pass, every padding, both shifts
ROWS_1D = [((2, 3, 2048), None), ((2, 2, 1032), [[1, 1]]), ((1, 3, 640), [[0, 3]]), ((2, 2, 528), None), ((1, 2, 4104), [[2, 1]])]


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("shape,crop", ROWS_1D)
def test_pooled_1d_rows(abi, shape, crop, dt):
    tdt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[dt]
    eps = {"f32": 0.0, "bf16": 2.0 ** -8, "f16": 2.0 ** -11}[dt]
    rs = np.random.RandomState(sum(shape) + 3)
    xt = torch.from_numpy(rs.uniform(-1, 1, size=shape).astype(np.float32)).to(tdt)
    wt = torch.from_numpy(rs.uniform(-6, 6, size=(shape[1], 1)).astype(np.float32)).to(tdt)
    wt[0, 0] = 2.5
    x, w = xt.float().numpy(), wt.float().numpy()
    b, new = abi.check_borders(list(shape), crop, 1)
    abi.set_tuning(32, 2)
    abi.set_tuning(34, 2)
    try:
        for pad in range(5):
            for active in (0, 1):
                key = (shape, crop, dt, pad, active)
                out = abi.forward_pooled(xt.to(DEV), wt.to(DEV), pad, active, (2,), b)
                assert abi.last_kernel() == "row_forward_pool", key + (abi.last_kernel(),)
                # the unfused sequence on widened inputs, the shift's output rounded to the storage type
                y = torch.from_numpy(O.forward(x, w, pad, active, b)).to(tdt).float().numpy()
                ref = O.avg_pool(y, (2,))
                got = out.float().cpu().numpy()
                if dt == "f32":
                    assert np.array_equal(got, O.forward_pooled(x, w, pad, active, (2,), b)), key
                else:
                    assert np.max(np.abs(got - ref)) <= eps * max(1.0, np.max(np.abs(ref))), key
                gpt = torch.from_numpy(rs.uniform(-1, 1, size=ref.shape).astype(np.float32)).to(tdt)
                gx, gw = abi.backward_pooled(gpt.to(DEV), wt.to(DEV), xt.to(DEV), pad, active, (2,), b)
                assert abi.last_kernel() == "row_backward_pool", key + (abi.last_kernel(),)
                if dt == "f32":
                    gx_r, gw_r = O.backward_pooled(gpt.float().numpy(), w, x, pad, active, (2,), b)
                    assert np.array_equal(gx.cpu().numpy(), gx_r), key
                    assert rel_err(gw.cpu().numpy(), gw_r) < 1e-5, key
                else:
                    g = torch.from_numpy(O.avg_pool_backward(gpt.float().numpy(), (2,), y.shape[2:])).to(tdt).float().numpy()
                    gx_r, gw_r = O.backward(g, w, x, pad, active, b)
                    assert np.max(np.abs(gx.float().cpu().numpy() - gx_r)) <= eps * max(1.0, np.max(np.abs(gx_r))), key
                    assert rel_err(gw.float().cpu().numpy(), gw_r) < 1.02 * eps + 1e-5, key
    finally:
        abi.set_tuning(32, 0)
        abi.set_tuning(34, 0)
