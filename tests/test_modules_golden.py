"""Module surface (Shift1d/2d/3d, emulate_dw heuristics, quantized from_float) vs fixtures recorded from the
reference's own Python package (tests/golden/make_golden_modules.py): same cut_borders, init_shift, weight
post-scale, padding id, pooling tail, output and loss for identical weights and inputs -- on the CPU key."""
import copy

import numpy as np
import torch

from cases import golden
from torchshifts import Shift1d, Shift2d, Shift3d
from torchshifts.quantized.modules import Shift2d as QShift2d
import torchshifts.quantized as tq

import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_golden_modules_cases import CASES  # noqa: E402  (the constructor grid: data only)

CLS = {1: Shift1d, 2: Shift2d, 3: Shift3d}


def test_module_grid_matches_reference():
    d = golden("modules.npz")
    for i, (dim, kw) in enumerate(CASES):
        m = CLS[dim](4, **copy.deepcopy(kw))
        key = "case%d_" % i
        if bool(d[key + "has_cut"]):
            assert m.cut_borders is not None and np.array_equal(m.cut_borders.numpy(), d[key + "cut_borders"]), i
        else:
            assert m.cut_borders is None, i
        assert np.array_equal(m.init_shift.numpy(), d[key + "init_shift"]), i
        assert np.array_equal(m._w_post_init_scale.numpy(), d[key + "scale"]), i
        assert m.padding == int(d[key + "padding"]), i
        assert (m._reduction_fn is not m._identity) == bool(d[key + "pooled"]), i
        assert m.weight.shape == (4, dim) and list(m.state_dict().keys()) == ["weight"]
        m.weight.data = torch.from_numpy(d[key + "w"].copy())
        out, loss = m(torch.from_numpy(d[key + "x"]))
        assert np.array_equal(out.detach().numpy(), d[key + "out"]), i
        ref_loss = float(d[key + "loss"])
        assert (loss is None) == np.isnan(ref_loss)
        if loss is not None:
            assert abs(float(loss) - ref_loss) <= 1e-6 * abs(ref_loss)


def test_quantized_from_float_matches_reference():
    d = golden("modules.npz")
    m = Shift2d(4, padding='border', sparsity_term=0.)
    m.weight.data = torch.from_numpy(d["q_w"].copy())
    q = QShift2d.from_float(m)
    assert q._get_name() == 'QuantizedShift2D'
    assert np.array_equal(q.qweight.int_repr().numpy(), d["q_qweight_repr"])
    assert q.qweight.q_scale() == float(d["q_qweight_scale"]) and q.qweight.q_zero_point() == int(d["q_qweight_zp"])
    xq = torch.quantize_per_tensor(torch.from_numpy(d["q_x"]), 1 / 255., 0, torch.quint8)
    out = q(xq)  # quantized modules return the tensor only
    assert isinstance(out, torch.Tensor) and np.array_equal(out.int_repr().numpy(), d["q_out_repr"])
    assert tq.quant_mapping[Shift2d] is QShift2d


def test_emulate_dw_dict_is_annotated_like_the_reference():
    args = {'kernel_size': 3, 'stride': 2, 'padding': 0}
    Shift2d(4, emulate_dw=args, init_thumb_rule=2)
    assert args['init_thumb_rule_type'] == 2  # the reference writes into the caller's dict (modules/shifts.py:125)
