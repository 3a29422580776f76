"""ctypes/numpy front end of the CPU ORACLE (oracle/shift_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product path (activesparseshifts-pytorch_amd/) never does.  Parity status: pinned against the
real reference build (oracle/_ref) through tests/golden/*.npz -- see shift_oracle.c's header.

All functions take numpy arrays (any strides) laid out as the reference's tensors:
x[N, C, H(, W(, D))], w[C, nD] (same dtype as x), borders = 6 ints [l_i, r_i, l_j, r_j, l_k, r_k]
as produced by check_borders (ops/shifts.cpp:93-135).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_SRC_PATH = os.path.join(_HERE, "shift_oracle.c")

PADDINGS = {"zeros": 0, "border": 1, "periodic": 2, "reflect": 3, "symmetric": 4}


def build(force=False):
    """Compile shift_oracle.c -> liboracle.so (gcc, no FMA contraction)."""
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(_SRC_PATH)):
        return _LIB_PATH
    tmp = _LIB_PATH + ".tmp.%d" % os.getpid()
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-Wall",
                           _SRC_PATH, "-o", tmp, "-lm"])
    os.replace(tmp, _LIB_PATH)
    return _LIB_PATH


class _Geom(ctypes.Structure):
    _fields_ = [("nd", ctypes.c_int32), ("pad", ctypes.c_int32), ("active", ctypes.c_int32),
                ("nhwc_order", ctypes.c_int32),
                ("N", ctypes.c_int64), ("C", ctypes.c_int64), ("H", ctypes.c_int64),
                ("W", ctypes.c_int64), ("D", ctypes.c_int64),
                ("x_s", ctypes.c_int64 * 5), ("o_s", ctypes.c_int64 * 5), ("gx_s", ctypes.c_int64 * 5),
                ("b", ctypes.c_int64 * 6)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_infer_index.restype = ctypes.c_int64
        _lib.oracle_infer_index.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int]
        _lib.oracle_abi_version.restype = ctypes.c_int
    return _lib


def infer_index(index, length, pad):
    return int(lib().oracle_infer_index(int(index), int(length), int(pad)))


def _estrides(a, nd):
    s = [st // a.itemsize for st in a.strides]
    assert all(st % a.itemsize == 0 for st in a.strides)
    return s + [0] * (5 - len(s))


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def _geom(x, nd, pad, active, borders, nhwc_order):
    g = _Geom()
    g.nd, g.pad, g.active, g.nhwc_order = nd, int(pad), int(bool(active)), int(bool(nhwc_order))
    shp = list(x.shape) + [1] * (5 - x.ndim)
    g.N, g.C, g.H, g.W, g.D = shp
    for i, v in enumerate(_estrides(x, nd)):
        g.x_s[i] = v
    for i, v in enumerate(borders):
        g.b[i] = int(v)
    return g


def default_borders(x):
    nd = x.ndim - 2
    b = []
    for d in range(3):
        b += [0, x.shape[2 + d] if d < nd else 1]
    return b


def out_shape(x, borders):
    nd = x.ndim - 2
    return tuple(list(x.shape[:2]) + [borders[2 * d + 1] - borders[2 * d] for d in range(nd)])


def _suffix(dt):
    if dt == np.float32:
        return "f32", ctypes.c_float
    if dt == np.float64:
        return "f64", ctypes.c_double
    raise TypeError("oracle float paths exist for float32/float64 only (cpu/shifts_cpu.cpp:228), got %s" % dt)


def weights_forward(w, active):
    """cpu/shifts_cpu.cpp:223-224 -> (iw int64 [C,nD], dw)."""
    suf, _ = _suffix(w.dtype)
    w = np.ascontiguousarray(w)
    iw = np.empty(w.shape, np.int64)
    dw = np.empty_like(w)
    getattr(lib(), "oracle_weights_forward_" + suf)(_ptr(w), ctypes.c_int64(w.size), int(bool(active)),
                                                    _ptr(iw), _ptr(dw))
    return iw, dw


def weights_backward(w, active):
    """cpu/shifts_cpu.cpp:242-244 -> (iw int64 [C,nD], dw)."""
    suf, _ = _suffix(w.dtype)
    w = np.ascontiguousarray(w)
    iw = np.empty(w.shape, np.int64)
    dw = np.empty_like(w)
    getattr(lib(), "oracle_weights_backward_" + suf)(_ptr(w), ctypes.c_int64(w.size), int(bool(active)),
                                                     _ptr(iw), _ptr(dw))
    return iw, dw


def forward(x, w, pad, active, borders=None, nhwc_order=False):
    """shiftnd_forward (cpu/shifts_cpu.cpp:216-232): returns a new C-contiguous array."""
    nd = x.ndim - 2
    assert 1 <= nd <= 3 and w.shape == (x.shape[1], nd) and w.dtype == x.dtype
    suf, _ = _suffix(x.dtype)
    borders = default_borders(x) if borders is None else list(borders)
    iw, dw = weights_forward(w, active)
    out = np.zeros(out_shape(x, borders), x.dtype)
    g = _geom(x, nd, pad, active, borders, nhwc_order)
    for i, v in enumerate(_estrides(out, nd)):
        g.o_s[i] = v
    getattr(lib(), "oracle_forward_" + suf)(ctypes.byref(g), _ptr(x), _ptr(iw), _ptr(dw), _ptr(out))
    return out


def backward(grad_out, w, x, pad, active, borders=None, nhwc_order=False):
    """shiftnd_backward (cpu/shifts_cpu.cpp:237-255): returns (grad_x, grad_w)."""
    nd = x.ndim - 2
    assert w.shape == (x.shape[1], nd) and w.dtype == x.dtype == grad_out.dtype
    suf, _ = _suffix(x.dtype)
    borders = default_borders(x) if borders is None else list(borders)
    assert grad_out.shape == out_shape(x, borders), (grad_out.shape, out_shape(x, borders))
    iw, dw = weights_backward(w, active)
    gx = np.zeros(x.shape, x.dtype)
    gw = np.zeros(w.shape, w.dtype)
    g = _geom(x, nd, pad, active, borders, nhwc_order)
    for i, v in enumerate(_estrides(grad_out, nd)):
        g.o_s[i] = v
    for i, v in enumerate(_estrides(gx, nd)):
        g.gx_s[i] = v
    getattr(lib(), "oracle_backward_" + suf)(ctypes.byref(g), _ptr(grad_out), _ptr(x), _ptr(iw), _ptr(dw),
                                             _ptr(gx), _ptr(gw))
    return gx, gw


def forward_q(xq, wq, w_zero_point, x_zero_point, pad, borders=None):
    """qshiftnd (quantized/shifts_quantized.cpp:107-130) on integer representations.

    xq: int8/uint8/int32 array (int_repr of the input), wq: integer array [C,nD] (int_repr of the
    weights), shift = wq - w_zero_point, fill = x_zero_point."""
    nd = xq.ndim - 2
    assert xq.dtype in (np.int8, np.uint8, np.int32)
    borders = default_borders(xq) if borders is None else list(borders)
    wq = np.ascontiguousarray(wq, dtype=np.int64)
    assert wq.shape == (xq.shape[1], nd)
    out = np.zeros(out_shape(xq, borders), xq.dtype)
    g = _geom(xq, nd, pad, False, borders, False)
    for i, v in enumerate(_estrides(out, nd)):
        g.o_s[i] = v
    fill = np.array([x_zero_point]).astype(xq.dtype)
    lib().oracle_forward_q(ctypes.byref(g), ctypes.c_int(xq.itemsize), _ptr(xq), _ptr(wq),
                           ctypes.c_int64(int(w_zero_point)), _ptr(fill), _ptr(out))
    return out


def check_borders(sizes, user, dim):
    """ops/shifts.cpp:93-135 -> (6 ints, new sizes)."""
    sizes_a = (ctypes.c_int64 * len(sizes))(*[int(s) for s in sizes])
    std = (ctypes.c_int32 * 6)()
    shift = 1 if (dim + 1) == len(sizes) else 2
    new = (ctypes.c_int64 * (shift + min(3, dim)))()
    if user is None:
        up = None
    else:
        flat = [int(v) for row in user for v in row]
        up = (ctypes.c_int32 * len(flat))(*flat)
    lib().oracle_check_borders(sizes_a, len(sizes), up, int(dim), std, new)
    return list(std), list(new)


# ---------------------------------------------------------------------------------------------
# The pooling tail of the reference's modules (torchshifts/modules/shifts.py:81-89, 150-153):
# `_reduction_fn(out)` = torch.nn.functional.avg_pool{N}d(out, kernel_size=k, stride=k, ceil_mode=True).
# The algorithm lives in ATen (torch 2.10, aten/src/ATen/native/cpu/AvgPoolKernel.cpp: the window is summed
# sequentially in row-major order starting from 0 and divided by the number of window elements inside the
# input when padding = 0; backward: every window element receives grad / that count).  Restated in numpy and
# pinned bit-exact against torch's CPU avg_pool by tests/test_oracle_golden.py::test_avg_pool_restatement and
# against the reference modules' pooled outputs in tests/golden/modules.npz.
# ---------------------------------------------------------------------------------------------
def _pool_counts(spatial, pooled, k):
    cnt = np.ones([1, 1] + list(pooled), np.int64)
    for d in range(len(k)):
        c = np.minimum(k[d], spatial[d] - np.arange(pooled[d]) * k[d])
        sh = [1] * (2 + len(k))
        sh[2 + d] = pooled[d]
        cnt = cnt * c.reshape(sh)
    return cnt


def avg_pool(y, k):
    """avg_pool{N}d(y, kernel_size=k, stride=k, ceil_mode=True) for y[N, C, spatial...] (float32/float64)."""
    import itertools
    k = [int(k)] * (y.ndim - 2) if np.isscalar(k) else [int(v) for v in k]
    sp = y.shape[2:]
    pooled = [-(-s // kk) for s, kk in zip(sp, k)]
    acc = np.zeros(y.shape[:2] + tuple(pooled), y.dtype)
    for u in itertools.product(*[range(kk) for kk in k]):
        part = y[(slice(None), slice(None)) + tuple(slice(uu, None, kk) for uu, kk in zip(u, k))]
        idx = (slice(None), slice(None)) + tuple(slice(0, n) for n in part.shape[2:])
        acc[idx] = acc[idx] + part
    return (acc / _pool_counts(sp, pooled, k).astype(y.dtype)).astype(y.dtype)


def avg_pool_backward(grad_pooled, k, spatial):
    """gradient of avg_pool() with respect to its input of spatial sizes `spatial`."""
    k = [int(k)] * (grad_pooled.ndim - 2) if np.isscalar(k) else [int(v) for v in k]
    q = (grad_pooled / _pool_counts(spatial, grad_pooled.shape[2:], k).astype(grad_pooled.dtype)).astype(grad_pooled.dtype)
    for d in range(len(k)):
        q = np.repeat(q, k[d], axis=2 + d)
        q = q[(slice(None),) * (2 + d) + (slice(0, spatial[d]),)]
    return np.ascontiguousarray(q)


def forward_pooled(x, w, pad, active, k, borders=None):
    """the module-level sequence shift -> avg_pool (modules/shifts.py:150-153)"""
    return avg_pool(forward(x, w, pad, active, borders), k)


def backward_pooled(grad_pooled, w, x, pad, active, k, borders=None):
    sp = out_shape(x, default_borders(x) if borders is None else borders)[2:]
    return backward(avg_pool_backward(grad_pooled, k, sp), w, x, pad, active, borders)
