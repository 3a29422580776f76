"""ctypes binding of the C ABI in include/shiftnd_hip.h (libshiftnd_hip.so).

This is the thinnest possible host side: it hands raw device pointers, sizes and strides of torch
tensors to the library.  The dispatcher ops in `_C.so` call exactly the same entry points; tests and
bench.py use this module to reach the kernels without dispatcher/allocator overhead.
"""
import ctypes
import os

import torch

# SHIFTND_HIP_LIB: another build of the library (kernel A/B runs with tools/kbench.py); the dispatcher ops in _C.so
# always use the in-tree library
_LIB_PATH = os.environ.get("SHIFTND_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libshiftnd_hip.so")

F32, F64, F16, BF16, I8, U8, I32 = range(7)
PATH_NONE, PATH_EMPTY, PATH_PLANE, PATH_STRIDED, PATH_SWEEP, PATH_CL = range(6)

DTYPES = {torch.float32: F32, torch.float64: F64, torch.float16: F16, torch.bfloat16: BF16,
          torch.int8: I8, torch.uint8: U8, torch.int32: I32}

EXPORTS = ["shiftnd_abi_version", "shiftnd_status_string", "shiftnd_last_path", "shiftnd_set_path_policy",
           "shiftnd_set_tuning", "shiftnd_debug_map", "shiftnd_last_kernel",
           "shiftnd_check_borders", "shiftnd_forward", "shiftnd_forward_serves_channels_last", "shiftnd_backward_serves_channels_last",
           "shiftnd_backward_workspace_bytes", "shiftnd_backward",
           "shiftnd_forward_quantized", "shiftnd_pooled_sizes", "shiftnd_backward_pooled_workspace_bytes", "shiftnd_forward_pooled", "shiftnd_backward_pooled", "shiftnd_forward_quantized_pooled", "shiftnd_transpose"]


class Problem(ctypes.Structure):
    _fields_ = [("ndim", ctypes.c_int32), ("dtype", ctypes.c_int32), ("padding_mode", ctypes.c_int32),
                ("active", ctypes.c_int32), ("sizes", ctypes.c_int64 * 5), ("borders", ctypes.c_int32 * 6)]


_lib = None

# Test aid (tests/conftest.py sets it): outputs this module allocates are filled with a byte pattern before the call, so that an
# output element no kernel wrote fails a comparison every time instead of only when the allocator hands back dirty memory.
POISON = False


def _new(shape, like):
    t = torch.empty(list(shape), dtype=like.dtype, device=like.device)
    if POISON and t.numel():
        t.view(-1).view(torch.uint8).fill_(0xA5)
    return t


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(_LIB_PATH)
        i64p = ctypes.POINTER(ctypes.c_int64)
        vp = ctypes.c_void_p
        L.shiftnd_abi_version.restype = ctypes.c_int
        L.shiftnd_status_string.restype = ctypes.c_char_p
        L.shiftnd_status_string.argtypes = [ctypes.c_int]
        L.shiftnd_last_path.restype = ctypes.c_int
        L.shiftnd_last_kernel.restype = ctypes.c_char_p
        L.shiftnd_set_path_policy.argtypes = [ctypes.c_int]
        L.shiftnd_set_tuning.argtypes = [ctypes.c_int, ctypes.c_int]
        L.shiftnd_debug_map.restype = ctypes.c_int
        L.shiftnd_debug_map.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int]
        L.shiftnd_check_borders.restype = ctypes.c_int
        L.shiftnd_check_borders.argtypes = [i64p, ctypes.c_int, ctypes.POINTER(ctypes.c_int32), ctypes.c_int,
                                            ctypes.POINTER(ctypes.c_int32), i64p]
        L.shiftnd_forward.restype = ctypes.c_int
        L.shiftnd_forward.argtypes = [ctypes.POINTER(Problem), vp, i64p, vp, vp, i64p, vp]
        L.shiftnd_forward_serves_channels_last.restype = ctypes.c_int
        L.shiftnd_forward_serves_channels_last.argtypes = [ctypes.POINTER(Problem), vp, i64p, vp, i64p]
        L.shiftnd_backward_serves_channels_last.restype = ctypes.c_int
        L.shiftnd_backward_serves_channels_last.argtypes = [ctypes.POINTER(Problem), vp, i64p, vp, i64p, vp, i64p]
        L.shiftnd_backward_workspace_bytes.restype = ctypes.c_size_t
        L.shiftnd_backward_workspace_bytes.argtypes = [ctypes.POINTER(Problem)]
        L.shiftnd_backward.restype = ctypes.c_int
        L.shiftnd_backward.argtypes = [ctypes.POINTER(Problem), vp, i64p, vp, i64p, vp, vp, i64p, vp, vp,
                                       ctypes.c_size_t, vp]
        L.shiftnd_forward_quantized.restype = ctypes.c_int
        L.shiftnd_forward_quantized.argtypes = [ctypes.POINTER(Problem), vp, i64p, vp, ctypes.c_int32, ctypes.c_int64,
                                                ctypes.c_int64, vp, i64p, vp]
        i32p = ctypes.POINTER(ctypes.c_int32)
        L.shiftnd_pooled_sizes.restype = ctypes.c_int
        L.shiftnd_pooled_sizes.argtypes = [ctypes.POINTER(Problem), i32p, i64p]
        L.shiftnd_backward_pooled_workspace_bytes.restype = ctypes.c_size_t
        L.shiftnd_backward_pooled_workspace_bytes.argtypes = [ctypes.POINTER(Problem), i32p]
        L.shiftnd_forward_pooled.restype = ctypes.c_int
        L.shiftnd_forward_pooled.argtypes = [ctypes.POINTER(Problem), i32p, vp, vp, vp, vp]
        L.shiftnd_backward_pooled.restype = ctypes.c_int
        L.shiftnd_backward_pooled.argtypes = [ctypes.POINTER(Problem), i32p, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]
        L.shiftnd_forward_quantized_pooled.restype = ctypes.c_int
        L.shiftnd_forward_quantized_pooled.argtypes = [ctypes.POINTER(Problem), i32p, vp, vp, ctypes.c_int32, ctypes.c_int64,
                                                       ctypes.c_int64, ctypes.c_int32, vp, vp]
        L.shiftnd_transpose.restype = ctypes.c_int
        L.shiftnd_transpose.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, vp]
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s (%d)" % (what, lib().shiftnd_status_string(rc).decode(), rc))


def strides5(t):
    s = list(t.stride()) + [0] * (5 - t.dim())
    return (ctypes.c_int64 * 5)(*s)


def default_borders(x):
    nd = x.dim() - 2
    b = []
    for d in range(3):
        b += [0, x.shape[2 + d] if d < nd else 1]
    return b


def check_borders(sizes, user, ndim):
    sizes_a = (ctypes.c_int64 * len(sizes))(*sizes)
    out = (ctypes.c_int32 * 6)()
    new = (ctypes.c_int64 * len(sizes))()
    up = None
    if user is not None:
        flat = [int(v) for row in user for v in row]
        up = (ctypes.c_int32 * len(flat))(*flat)
    check(lib().shiftnd_check_borders(sizes_a, len(sizes), up, ndim, out, new), "shiftnd_check_borders")
    shift = 1 if ndim + 1 == len(sizes) else 2
    return list(out), list(new)[:shift + min(ndim, 3)]


def problem(x, pad, active, borders, dtype=None):
    p = Problem()
    p.ndim = x.dim() - 2
    p.dtype = DTYPES[x.dtype] if dtype is None else dtype
    p.padding_mode = int(pad)
    p.active = int(bool(active))
    for i in range(5):
        p.sizes[i] = x.shape[i] if i < x.dim() else 1
    for i, v in enumerate(default_borders(x) if borders is None else borders):
        p.borders[i] = int(v)
    return p


def out_shape(x, borders):
    nd = x.dim() - 2
    b = default_borders(x) if borders is None else borders
    return list(x.shape[:2]) + [b[2 * d + 1] - b[2 * d] for d in range(nd)]


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def forward(x, w, pad, active, borders=None, out=None):
    """x, w: device tensors (float dtypes).  Returns a new contiguous output (or fills `out`)."""
    p = problem(x, pad, active, borders)
    if out is None:
        out = _new(out_shape(x, borders), x)
    w = w.contiguous()
    check(lib().shiftnd_forward(ctypes.byref(p), x.data_ptr(), strides5(x), w.data_ptr(), out.data_ptr(), strides5(out),
                                _stream()), "shiftnd_forward")
    return out


def backward_workspace(x, pad, active, borders=None):
    p = problem(x, pad, active, borders)
    return torch.empty(int(lib().shiftnd_backward_workspace_bytes(ctypes.byref(p))), dtype=torch.uint8, device=x.device)


def backward(grad_out, w, x, pad, active, borders=None, grad_x=None, grad_w=None, workspace=None):
    p = problem(x, pad, active, borders)
    w = w.contiguous()
    if grad_x is None:
        grad_x = _new(x.shape, x)
    if grad_w is None:
        grad_w = _new(w.shape, w)
    if workspace is None:
        workspace = backward_workspace(x, pad, active, borders)
    check(lib().shiftnd_backward(ctypes.byref(p), grad_out.data_ptr(), strides5(grad_out), x.data_ptr(), strides5(x),
                                 w.data_ptr(), grad_x.data_ptr(), strides5(grad_x), grad_w.data_ptr(),
                                 workspace.data_ptr(), workspace.numel(), _stream()), "shiftnd_backward")
    return grad_x, grad_w


def forward_quantized(xq, wq, w_zero_point, x_zero_point, pad, borders=None, out=None):
    """xq: int8/uint8/int32 device tensor (int_repr); wq: int8/uint8/int32 device tensor [C, nd]."""
    p = problem(xq, pad, False, borders)
    if out is None:
        out = _new(out_shape(xq, borders), xq)
    wq = wq.contiguous()
    check(lib().shiftnd_forward_quantized(ctypes.byref(p), xq.data_ptr(), strides5(xq), wq.data_ptr(), DTYPES[wq.dtype],
                                          int(w_zero_point), int(x_zero_point), out.data_ptr(), strides5(out), _stream()),
          "shiftnd_forward_quantized")
    return out


def _pool_arg(pool, nd):
    pool = [int(pool)] * nd if isinstance(pool, int) else [int(k) for k in pool]
    assert len(pool) == nd
    return (ctypes.c_int32 * nd)(*pool)


def pooled_shape(x, pool, borders=None):
    p = problem(x, 0, False, borders)
    sp = (ctypes.c_int64 * 3)()
    check(lib().shiftnd_pooled_sizes(ctypes.byref(p), _pool_arg(pool, p.ndim), sp), "shiftnd_pooled_sizes")
    return list(x.shape[:2]) + list(sp)[:p.ndim]


def forward_pooled(x, w, pad, active, pool, borders=None, out=None):
    """Fused shift + avg_pool(kernel = stride = pool, ceil_mode=True); x, w contiguous device tensors."""
    assert x.is_contiguous()
    p = problem(x, pad, active, borders)
    if out is None:
        out = _new(pooled_shape(x, pool, borders), x)
    w = w.contiguous()
    check(lib().shiftnd_forward_pooled(ctypes.byref(p), _pool_arg(pool, p.ndim), x.data_ptr(), w.data_ptr(),
                                       out.data_ptr(), _stream()), "shiftnd_forward_pooled")
    return out


REQUANT_ZP_INSIDE, REQUANT_ZP_OUTSIDE = 0, 1


def forward_quantized_pooled(xq, wq, w_zero_point, x_zero_point, pad, pool, borders=None, out=None, requant=REQUANT_ZP_INSIDE):
    """Quantized shift + avg_pool(kernel = stride = pool, ceil_mode=True) in one pass; xq: int8 / uint8 contiguous device
    tensor (int_repr), wq: integer tensor [C, nd]; requant: which of ATen's two roundings (include/shiftnd_hip.h)."""
    assert xq.is_contiguous()
    p = problem(xq, pad, False, borders)
    if out is None:
        out = _new(pooled_shape(xq, pool, borders), xq)
    wq = wq.contiguous()
    check(lib().shiftnd_forward_quantized_pooled(ctypes.byref(p), _pool_arg(pool, p.ndim), xq.data_ptr(), wq.data_ptr(),
                                                 DTYPES[wq.dtype], int(w_zero_point), int(x_zero_point), int(requant), out.data_ptr(),
                                                 _stream()), "shiftnd_forward_quantized_pooled")
    return out


def backward_pooled_workspace_bytes(x, pad, active, pool, borders=None):
    p = problem(x, pad, active, borders)
    return int(lib().shiftnd_backward_pooled_workspace_bytes(ctypes.byref(p), _pool_arg(pool, p.ndim)))


def backward_pooled(grad_pooled, w, x, pad, active, pool, borders=None, grad_x=None, grad_w=None, workspace=None):
    assert x.is_contiguous() and grad_pooled.is_contiguous()
    p = problem(x, pad, active, borders)
    w = w.contiguous()
    if grad_x is None:
        grad_x = _new(x.shape, x)
    if grad_w is None:
        grad_w = _new(w.shape, w)
    if workspace is None:
        nbytes = int(lib().shiftnd_backward_pooled_workspace_bytes(ctypes.byref(p), _pool_arg(pool, p.ndim)))
        workspace = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    check(lib().shiftnd_backward_pooled(ctypes.byref(p), _pool_arg(pool, p.ndim), grad_pooled.data_ptr(), x.data_ptr(),
                                        w.data_ptr(), grad_x.data_ptr(), grad_w.data_ptr(), workspace.data_ptr(),
                                        workspace.numel(), _stream()), "shiftnd_backward_pooled")
    return grad_x, grad_w


def to_contiguous(x):
    """channels-last dense [N, C, spatial...] device tensor -> new contiguous tensor (shiftnd_transpose)"""
    out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    P = x[0, 0].numel()
    check(lib().shiftnd_transpose(x.data_ptr(), out.data_ptr(), x.shape[0], P, x.shape[1], x.element_size(), _stream()),
          "shiftnd_transpose")
    return out


def to_channels_last(x):
    """contiguous [N, C, spatial...] device tensor -> new channels-last dense tensor (shiftnd_transpose)"""
    fmt = torch.channels_last if x.dim() == 4 else torch.channels_last_3d
    out = torch.empty(x.shape, dtype=x.dtype, device=x.device, memory_format=fmt)
    P = x[0, 0].numel()
    check(lib().shiftnd_transpose(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], P, x.element_size(), _stream()),
          "shiftnd_transpose")
    return out


def last_path():
    return lib().shiftnd_last_path()


def last_kernel():
    return lib().shiftnd_last_kernel().decode()


def set_path_policy(policy):
    lib().shiftnd_set_path_policy(int(policy))


def set_tuning(knob, value):
    """diagnostics: launch-planning knobs (see include/shiftnd_hip.h)"""
    lib().shiftnd_set_tuning(int(knob), int(value))
