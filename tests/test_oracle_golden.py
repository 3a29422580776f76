"""The CPU oracle (oracle/shift_oracle.c) must reproduce the REAL reference bit for bit.

Fixtures in tests/golden/ were produced by tests/golden/make_golden.py from oracle/_ref (the
reference's own C++ built in place from /root/reference).  This pins the oracle; the GPU parity
tests then compare the HIP kernels with the oracle.
"""
import os

import numpy as np
import pytest

from oracle import oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(G, name))


def _borders(shape, crop):
    nd = len(shape) - 2
    b, _ = O.check_borders(list(shape), None if crop is None else crop.tolist(), nd)
    return b


def _eq(a, b):
    """bit-exact, treating -0.0 == +0.0 as different only through the bit pattern of non-zeros"""
    return a.shape == b.shape and np.array_equal(a, b)


@pytest.mark.parametrize("fixture", ["g1_float.npz", "g1_random.npz"])
@pytest.mark.parametrize("nd", [1, 2, 3])
@pytest.mark.parametrize("dt", ["f32", "f64"])
def test_float_grid_bit_exact(fixture, nd, dt):
    d = _load(fixture)
    x, go_full, w = d["x_%dd_%s" % (nd, dt)], d["go_%dd_%s" % (nd, dt)], d["w_%dd_%s" % (nd, dt)]
    crops = d["crops_%dd" % nd]
    for pad in range(5):
        for active in (0, 1):
            for ci, crop in enumerate((None, crops)):
                key = "%dd_%s_p%d_a%d_c%d" % (nd, dt, pad, active, ci)
                b = _borders(x.shape, crop)
                out = O.forward(x, w, pad, active, b)
                assert _eq(out, d["out_" + key]), "forward " + key
                go = np.ascontiguousarray(go_full[tuple(slice(0, s) for s in out.shape)])
                gx, gw = O.backward(go, w, x, pad, active, b)
                assert _eq(gx, d["gx_" + key]), "grad_x " + key
                # single-threaded reference sums in the same (n, c, i, j, k) order -> bit-exact
                assert _eq(gw, d["gw_" + key]), "grad_w " + key


def test_known_answer_1d():
    d = _load("ka_1d.npz")
    x, g = d["x"], d["g"]
    for wi, wv in enumerate(d["ws"]):
        w = np.array([[wv]], np.float32)
        for pad in range(5):
            for active in (0, 1):
                key = "w%d_p%d_a%d" % (wi, pad, active)
                assert _eq(O.forward(x, w, pad, active), d["out_" + key]), key
                gx, gw = O.backward(g, w, x, pad, active)
                assert _eq(gx, d["gx_" + key]) and _eq(gw, d["gw_" + key]), key
    # the survey's hand-checked vectors
    w = np.array([[0.25]], np.float32)
    assert O.forward(x, w, 0, True).ravel().tolist() == [1.25, 2.5, 5, 10, 20, 24]
    assert O.forward(x, w, 2, True).ravel().tolist() == [1.25, 2.5, 5, 10, 20, 24.25]
    gx, gw = O.backward(g, np.array([[1.25]], np.float32), x, 0, False)
    assert gx.ravel().tolist() == [10, 100, 1e3, 1e4, 1e5, 0] and gw.item() == 1684211


def test_index_maps():
    d = _load("ka_1d.npz")
    for length in (1, 2, 3, 5):
        for pad in range(5):
            ref = d["map_len%d_p%d" % (length, pad)]
            for si, s in enumerate(range(-12, 13)):
                for i in range(length):
                    m = 0 if length == 1 else O.infer_index(i - s, length, pad)
                    m = m if m >= 0 else -1
                    assert m == ref[si, i], (length, pad, s, i)


@pytest.mark.parametrize("nd", [1, 2, 3])
def test_quantized_bit_exact(nd):
    d = _load("g2_quant.npz")
    crops = {1: [[1, 2]], 2: [[1, 2], [0, 1]], 3: [[1, 2], [0, 1], [1, 0]]}[nd]
    layouts = {1: ["nchw"], 2: ["nchw", "cl"], 3: ["nchw", "cl3d"]}[nd]
    for xname in ("quint8", "qint8", "qint32"):
        xq = d["xq_%dd_%s" % (nd, xname)]
        xzp = int(d["xzp_%s" % xname])
        for layout in layouts:
            xin = xq
            if layout != "nchw":  # channels-last memory, same logical NC... view
                perm = [0] + list(range(2, xq.ndim)) + [1]
                inv = np.argsort(perm)
                xin = np.ascontiguousarray(xq.transpose(perm)).transpose(inv)
            for wname in ("wu8", "wi8"):
                wq, wzp = d["wq_%dd_%s" % (nd, wname)], int(d["wzp_%s" % wname])
                for pad in range(5):
                    for ci, crop in enumerate((None, crops)):
                        key = "%dd_%s_%s_%s_p%d_c%d" % (nd, xname, layout, wname, pad, ci)
                        b, _ = O.check_borders(list(xq.shape), crop, nd)
                        out = O.forward_q(xin, wq, wzp, xzp, pad, b)
                        assert _eq(out, d["out_" + key]), key


def test_check_borders_windows():
    d = _load("borders.npz")
    x = d["x"]
    for i, c in enumerate(d["cases"]):
        b, new = O.check_borders(list(x.shape), c.tolist(), 2)
        ref = d["out_%d" % i]
        assert list(ref.shape) == new, (c, new, ref.shape)
        assert _eq(x[:, :, b[0]:b[1], b[2]:b[3]], ref)
        w = np.zeros((3, 2))
        assert _eq(O.forward(x, w, 0, False, b), ref)
    b, new = O.check_borders(list(d["x3"].shape), d["case3"].tolist(), 3)
    assert list(d["out3"].shape) == new
    assert _eq(d["x3"][:, :, b[0]:b[1], b[2]:b[3], b[4]:b[5]], d["out3"])


def test_channels_last_loop_order_matches_values():
    """NHWC loop order (cpu/shifts_cpu.cpp:57-75) changes only the grad_w summation order."""
    d = _load("g1_float.npz")
    x, go, w = d["x_2d_f64"], d["go_2d_f64"], d["w_2d_f64"]
    a = O.backward(go, w, x, 3, 1, None, nhwc_order=False)
    b = O.backward(go, w, x, 3, 1, None, nhwc_order=True)
    assert _eq(a[0], b[0]) and _eq(a[1], b[1])  # exact data -> order independent


def test_avg_pool_restatement():
    """the oracle's numpy restatement of ATen's avg_pool (ceil mode, kernel = stride) and of its backward is
    bit-exact against torch's CPU kernels: this pins the pooled-sequence oracle used by the fused-op tests"""
    import torch
    torch.set_num_threads(1)
    F = {1: torch.nn.functional.avg_pool1d, 2: torch.nn.functional.avg_pool2d, 3: torch.nn.functional.avg_pool3d}
    rs = np.random.RandomState(0)
    for dt in (np.float32, np.float64):
        for shape, k in [((2, 3, 17), (2,)), ((2, 3, 16), (3,)), ((2, 3, 9, 11), (2, 2)), ((2, 3, 10, 12), (3, 2)),
                         ((1, 2, 7, 7), (4, 4)), ((2, 2, 5, 6, 7), (2, 2, 2)), ((1, 2, 6, 7, 9), (2, 3, 4)),
                         ((1, 1, 1, 5), (2, 2))]:
            nd = len(k)
            y = rs.uniform(-1, 1, size=shape).astype(dt)
            t = torch.from_numpy(y.copy()).requires_grad_(True)
            kk = k[0] if nd == 1 else list(k)
            o = F[nd](t, kernel_size=kk, stride=kk, ceil_mode=True)
            assert np.array_equal(O.avg_pool(y, k), o.detach().numpy()), (shape, k)
            g = rs.uniform(-1, 1, size=tuple(o.shape)).astype(dt)
            o.backward(torch.from_numpy(g))
            assert np.array_equal(O.avg_pool_backward(g, k, shape[2:]), t.grad.numpy()), (shape, k)
