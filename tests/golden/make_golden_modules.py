#!/usr/bin/env python3
"""Module-level golden fixtures (SURVEY.md section 8c "G3") from the REAL reference Python package.

Runs only in the build container.  The reference's `torchshifts` package is imported from
/root/reference (read-only) on top of oracle/_ref/_C.so; two shims make it importable on torch 2.x
without touching the reference: (1) the attribute `DEFAULT_OP_LIST_TO_FUSER_METHOD` its
quantized/__init__.py copies (and never uses) is provided on the torch module for the duration of the
import; (2) its extension loader looks for `_C.so` inside /root/reference (absent), so the ops are
registered by loading oracle/_ref/_C.so first and `_has_ops` is pointed at True.

Records, for a grid of constructor arguments: cut_borders, init_shift, the weight post-scale, the
padding id, whether a pooling tail is attached, and -- for fixed weights and inputs -- the module's
output and the quantized module's int_repr output.  Writes tests/golden/modules.npz.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
torch.set_num_threads(1)
torch.ops.load_library(os.path.join(ROOT, "oracle", "_ref", "_C.so"))

import torch.quantization.fuser_method_mappings as _fmm  # noqa: E402

if not hasattr(_fmm, "DEFAULT_OP_LIST_TO_FUSER_METHOD"):
    _fmm.DEFAULT_OP_LIST_TO_FUSER_METHOD = {}
sys.path.insert(0, "/root/reference")
import torchshifts  # noqa: E402  (the reference's Python)
import torchshifts.extension as _ext  # noqa: E402

_ext._has_ops = lambda: True
from torchshifts import Shift1d, Shift2d, Shift3d  # noqa: E402
from torchshifts.quantized.modules import Shift2d as QShift2d  # noqa: E402

assert torchshifts.__file__.startswith("/root/reference"), torchshifts.__file__

sys.path.insert(0, HERE)
from make_golden_modules_cases import CASES  # noqa: E402

CLS = {1: Shift1d, 2: Shift2d, 3: Shift3d}
SHAPES = {1: (2, 4, 16), 2: (2, 4, 10, 12), 3: (2, 4, 6, 8, 8)}


def main():
    d = {}
    rs = np.random.RandomState(7)
    for i, (dim, kw) in enumerate(CASES):
        import copy
        m = CLS[dim](4, **copy.deepcopy(kw))
        key = "case%d_" % i
        d[key + "cut_borders"] = np.array([]) if m.cut_borders is None else m.cut_borders.numpy()
        d[key + "has_cut"] = np.array(m.cut_borders is not None)
        d[key + "init_shift"] = m.init_shift.numpy()
        d[key + "scale"] = m._w_post_init_scale.numpy()
        d[key + "padding"] = np.array(m.padding)
        d[key + "pooled"] = np.array(m._reduction_fn is not m._identity)
        w = rs.uniform(-2.5, 2.5, size=(4, dim)).astype(np.float32)
        x = rs.uniform(-1, 1, size=SHAPES[dim]).astype(np.float32)
        m.weight.data = torch.from_numpy(w.copy())
        out, loss = m(torch.from_numpy(x))
        d[key + "w"], d[key + "x"], d[key + "out"] = w, x, out.detach().numpy()
        d[key + "loss"] = np.array(np.nan if loss is None else float(loss))
    # quantized module: from_float + forward on a quantized input
    m = Shift2d(4, padding='border', sparsity_term=0.)
    w = np.array([[1.4, -2.6], [0.5, 1.5], [-0.5, 2.5], [3.0, -3.49]], np.float32)
    m.weight.data = torch.from_numpy(w.copy())
    q = QShift2d.from_float(m)
    x = rs.uniform(0, 1, size=(2, 4, 9, 11)).astype(np.float32)
    xq = torch.quantize_per_tensor(torch.from_numpy(x), 1 / 255., 0, torch.quint8)
    d["q_w"], d["q_x"] = w, x
    d["q_qweight_repr"] = q.qweight.int_repr().numpy()
    d["q_qweight_scale"] = np.array(q.qweight.q_scale())
    d["q_qweight_zp"] = np.array(q.qweight.q_zero_point())
    d["q_out_repr"] = q(xq).int_repr().numpy()
    path = os.path.join(HERE, "modules.npz")
    np.savez_compressed(path, **d)
    print("modules.npz %d arrays %.1f KB" % (len(d), os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()
