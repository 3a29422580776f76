"""CPU / QuantizedCPU dispatch keys of the drop-in `torchshifts` library vs the golden fixtures
produced by the real reference (tests/golden/make_golden.py), through the reference's own call
path: torch.ops.torchshifts.shift{N}d + autograd.

This is BASELINE config 1 ("CPU reference path, plumbing") and the host-logic coverage of the
operator boundary: schemas, check_borders, autograd glue, error behaviour.
"""
import numpy as np
import pytest
import torch

import torchshifts  # noqa: F401  (loads _C.so)
from cases import float_cases, quant_cases, rel_err, golden

OPS = torch.ops.torchshifts


def _op(nd):
    return getattr(OPS, "shift%dd" % nd)


def _run(x, w, pad, active, crop, go_full):
    xt = torch.from_numpy(x.copy()).requires_grad_(True)
    wt = torch.from_numpy(w.copy()).requires_grad_(True)
    b = torch.Tensor() if crop is None else torch.tensor(crop, dtype=torch.long)
    out = _op(x.ndim - 2)(xt, wt, b, pad, bool(active))
    go = torch.from_numpy(np.ascontiguousarray(go_full[tuple(slice(0, s) for s in out.shape)]))
    out.backward(go)
    return out.detach().numpy(), xt.grad.numpy(), wt.grad.numpy()


def test_exact_fixture_bit_exact():
    """inputs are dyadic rationals: every output incl. grad_w is exactly representable -> bit-exact"""
    n = 0
    for key, nd, dt, pad, active, crop, x, w, go, out_r, gx_r, gw_r in float_cases("g1_float.npz"):
        out, gx, gw = _run(x, w, pad, active, crop, go)
        assert np.array_equal(out, out_r), "forward " + key
        assert np.array_equal(gx, gx_r), "grad_x " + key
        assert np.array_equal(gw, gw_r), "grad_w " + key
        n += 1
    assert n == 120


def test_random_fixture_tolerances():
    d = golden("g1_random.npz")
    for key, nd, dt, pad, active, crop, x, w, go, out_r, gx_r, gw_r in float_cases("g1_random.npz"):
        out, gx, gw = _run(x, w, pad, active, crop, go)
        # forward / grad_x: same expression order as the reference, no FMA -> bit-exact
        assert np.array_equal(out, out_r), "forward " + key
        assert np.array_equal(gx, gx_r), "grad_x " + key
        # grad_w: fp64-accumulated here, sequential scalar_t sum in the reference; compare with the
        # reference's fp64 run on the same (fp32-representable) data at 1e-5 relative
        if dt == "f64":
            assert rel_err(gw, gw_r) < 1e-12, "grad_w " + key
        else:
            assert rel_err(gw, gw_r) < 1e-5, "grad_w " + key


def _qtensor(xi, name, zp, layout):
    dt = {"quint8": torch.uint8, "qint8": torch.int8, "qint32": torch.int32}[name]
    q = torch._make_per_tensor_quantized_tensor(torch.from_numpy(xi).to(dt), 0.05, zp)
    if layout == "cl":
        q = q.contiguous(memory_format=torch.channels_last)
    elif layout == "cl3d":
        q = q.contiguous(memory_format=torch.channels_last_3d)
    return q


def test_quantized_bit_exact():
    n = 0
    for key, nd, xname, layout, wname, pad, crop, xq, xzp, wq, wzp, out_r in quant_cases():
        x = _qtensor(xq, xname, xzp, layout)
        w = torch._make_per_tensor_quantized_tensor(
            torch.from_numpy(wq).to(torch.uint8 if wname == "wu8" else torch.int8), 1.0, wzp)
        b = torch.Tensor() if crop is None else torch.tensor(crop, dtype=torch.long)
        out = _op(nd)(x, w, b, pad, False)
        assert out.q_zero_point() == xzp and abs(out.q_scale() - 0.05) < 1e-12
        assert np.array_equal(out.int_repr().contiguous().numpy(), out_r), key
        if layout != "nchw":  # channels-last in -> channels-last out (shifts_quantized.cpp:119-121)
            mf = torch.channels_last if nd == 2 else torch.channels_last_3d
            assert out.is_contiguous(memory_format=mf), key
        n += 1
    assert n == 300


def test_channels_last_float_matches_nchw():
    d = golden("g1_random.npz")
    x, w = torch.from_numpy(d["x_2d_f32"]), torch.from_numpy(d["w_2d_f32"])
    for pad in range(5):
        for active in (False, True):
            a = OPS.shift2d(x, w, torch.Tensor(), pad, active)
            b = OPS.shift2d(x.contiguous(memory_format=torch.channels_last), w, torch.Tensor(), pad, active)
            assert torch.equal(a, b) and b.is_contiguous()  # float path: output is NCHW-contiguous


def test_borders_windows():
    d = golden("borders.npz")
    x = torch.from_numpy(d["x"])
    w = torch.zeros(3, 2, dtype=torch.float64)
    for i, c in enumerate(d["cases"]):
        out = OPS.shift2d(x, w, torch.tensor(c, dtype=torch.long), 0, False)
        assert np.array_equal(out.numpy(), d["out_%d" % i]), c
    out3 = OPS.shift3d(torch.from_numpy(d["x3"]), torch.zeros(2, 3, dtype=torch.float64),
                       torch.tensor(d["case3"], dtype=torch.long), 0, False)
    assert np.array_equal(out3.numpy(), d["out3"])
    with pytest.raises(RuntimeError):  # left cut beyond the size -> negative dimension
        OPS.shift2d(x, w, torch.tensor([[7, 0], [0, 0]]), 0, False)


def test_schemas_match_reference():
    """ops/shifts.cpp:168-181 and torchshifts.cpp:35-40"""
    for nd in (1, 2, 3):
        f = str(getattr(OPS, "_shift%dd_forward" % nd).default._schema)
        b = str(getattr(OPS, "_shift%dd_backward" % nd).default._schema)
        assert f == ("torchshifts::_shift%dd_forward(Tensor input, Tensor weights, Tensor borders, int[] new_size, "
                     "int padding_mode, bool active_flag) -> Tensor" % nd)
        assert b == ("torchshifts::_shift%dd_backward(Tensor grad, Tensor weights, Tensor input, Tensor borders, "
                     "int padding_mode, bool active_flag) -> (Tensor, Tensor)" % nd)
        p = str(getattr(OPS, "shift%dd" % nd).default._schema)
        assert p == "torchshifts::shift%dd(Tensor _0, Tensor _1, Tensor _2, int _3, bool _4) -> Tensor _0" % nd
    assert isinstance(OPS._cuda_version(), int)


def test_error_behaviour():
    x = torch.rand(2, 3, 5, 5)
    w = torch.rand(3, 2)
    with pytest.raises(RuntimeError):  # Q9: weights dtype must equal input dtype
        OPS.shift2d(x, w.double(), torch.Tensor(), 0, False)
    with pytest.raises(RuntimeError):  # fp16 has no CPU kernel in the reference either
        OPS.shift2d(x.half(), w.half(), torch.Tensor(), 0, False)
    xq = torch.quantize_per_tensor(x, 0.1, 0, torch.quint8)
    wq = torch.quantize_per_tensor(w, 1.0, 128, torch.quint8)
    with pytest.raises(RuntimeError, match="backwards on quantized tensor are not supported"):
        OPS._shift2d_backward(xq, wq, xq, torch.tensor([0, 5, 0, 5, 0, 1], dtype=torch.int32), 0, False)
    # double backward guard (shifts_autograd.cpp:68-71)
    xt = x.clone().requires_grad_(True)
    wt = w.clone().requires_grad_(True)
    out = OPS.shift2d(xt, wt, torch.Tensor(), 0, True)
    (gx,) = torch.autograd.grad(out.sum(), xt, create_graph=True)
    with pytest.raises(RuntimeError, match="double backwards on shift2d not supported"):
        gx.sum().backward()


def test_weight_grad_is_deterministic_multithreaded():
    """the reference's CPU backward races on grad_w (global_scope.h:22); ours must not"""
    torch.manual_seed(0)
    x = torch.rand(8, 4, 32, 32)
    w = torch.tensor([[0.3, -1.2], [1.5, 0.5], [-2.5, 2.2], [0.0, 0.7]])
    go = torch.rand(8, 4, 32, 32)
    b = torch.tensor([0, 32, 0, 32, 0, 1], dtype=torch.int32)
    old = torch.get_num_threads()
    try:
        torch.set_num_threads(1)
        ref = OPS._shift2d_backward(go, w, x, b, 3, True)
        torch.set_num_threads(8)
        for _ in range(3):
            got = OPS._shift2d_backward(go, w, x, b, 3, True)
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    finally:
        torch.set_num_threads(old)


def test_pool_op_is_the_two_step_sequence_on_cpu():
    """torchshifts::shift{N}d_pool on CPU tensors == shift{N}d followed by avg_pool{N}d(ceil_mode=True), values and
    gradients (the reference modules' sequence, modules/shifts.py:150-153)"""
    import torchshifts.functional as TF
    torch.manual_seed(3)
    F = {1: torch.nn.functional.avg_pool1d, 2: torch.nn.functional.avg_pool2d, 3: torch.nn.functional.avg_pool3d}
    for nd, shape, pool, crop in [(1, (2, 3, 17), 2, None), (2, (2, 4, 9, 12), (2, 3), [[1, 0], [0, 2]]),
                                  (3, (1, 2, 5, 6, 8), 2, None)]:
        for pad, active in ((0, False), (3, True)):
            x = torch.rand(shape, dtype=torch.float64)
            w = torch.rand(shape[1], nd, dtype=torch.float64) * 4 - 2
            b = None if crop is None else torch.tensor(crop)
            xa, wa = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            out = getattr(TF, "shift%dd_pool_func" % nd)(xa, wa, pad, active, b, pool)
            xb, wb = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            y = getattr(TF, "shift%dd_func" % nd)(xb, wb, pad, active, b)
            k = pool if isinstance(pool, int) else list(pool)
            ref = F[nd](y, kernel_size=k, stride=k, ceil_mode=True)
            assert torch.equal(out, ref)
            g = torch.rand_like(ref)
            out.backward(g)
            ref.backward(g)
            assert torch.equal(xa.grad, xb.grad) and torch.equal(wa.grad, wb.grad)
    with pytest.raises(AssertionError):
        TF.shift2d_pool_func(torch.rand(1, 2, 4, 4), torch.zeros(2, 2), 0, False, None, [2])
    with pytest.raises(RuntimeError, match="double backwards"):
        x = torch.rand(1, 2, 6, 6, requires_grad=True)
        w = torch.zeros(2, 2, requires_grad=True)
        out = TF.shift2d_pool_func(x, w, 0, False, None, 2)
        (gx,) = torch.autograd.grad(out.sum(), x, create_graph=True)
        gx.sum().backward()
