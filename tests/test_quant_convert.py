"""torch.ao.quantization.convert(..., mapping=torchshifts.quant_mapping) swaps float shift
modules for the quantized ones (the reference's README:87-92 flow; its own mapping table no longer imports on
torch 2.x); the quantized weights checkpoint and move with the module; quantize_shift_weights never changes a shift.
CPU only: exercises the QuantizedCPU key."""
import math
import warnings

import pytest
import torch
from torch import nn

import torchshifts
from torchshifts import Shift1d, Shift2d, Shift3d, quant_mapping
from torchshifts.quantized.modules.shifts import quantize_shift_weights

QMODS = torchshifts.quantized.modules


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.quant = torch.ao.quantization.QuantStub()
        self.shift = Shift2d(4, padding='border', sparsity_term=0.)
        self.dequant = torch.ao.quantization.DeQuantStub()

    def forward(self, x):
        x = self.quant(x)
        out = self.shift(x)
        out = out[0] if isinstance(out, tuple) else out  # float module returns (out, loss), quantized the tensor
        return self.dequant(out)


def test_convert_with_quant_mapping():
    torch.manual_seed(0)
    net = Net().eval()
    net.shift.weight.data = torch.tensor([[1.2, -0.7], [0.4, 2.6], [-1.5, 0.5], [3.0, -2.2]])
    x = torch.rand(2, 4, 12, 12)
    ref = net(x)
    net.qconfig = torch.ao.quantization.default_qconfig
    torch.ao.quantization.prepare(net, inplace=True)
    net(x)  # calibrate the stubs
    torch.ao.quantization.convert(net, inplace=True, mapping=quant_mapping)
    assert type(net.shift) is QMODS.Shift2d
    out = net(x)
    # a pure gather: the quantized result is the float result up to the input's quantisation step
    assert out.shape == ref.shape and (out - ref).abs().max() < 2.0 / 127


def test_quantized_weights_checkpoint_and_move():
    torch.manual_seed(1)
    q = QMODS.Shift2d.from_float(Shift2d(3, init_shift=3, sparsity_term=0.))
    sd = q.state_dict()
    assert "weight" in sd and "qweight_repr" in sd and "_extra_state" in sd
    fresh = QMODS.Shift2d(3)
    assert not torch.equal(fresh.qweight.int_repr(), q.qweight.int_repr())
    fresh.load_state_dict(sd)
    assert torch.equal(fresh.qweight.int_repr(), q.qweight.int_repr())
    assert fresh.qweight.q_zero_point() == q.qweight.q_zero_point() == 128 and fresh.qweight.dtype == torch.quint8
    # .to() moves the buffer the property is built from (meta device stands in for the GPU here)
    assert q.to("meta").qweight_repr.device.type == "meta"
    # assigning a quantized tensor goes through the property and stays registered
    q2 = QMODS.Shift2d(3)
    q2.qweight = torch.quantize_per_tensor(torch.tensor([[1., -2.], [0., 3.], [5., 5.]]), 1.0, 128, torch.quint8)
    assert "qweight_repr" in dict(q2.named_buffers()) and q2.qweight.int_repr().tolist() == [[129, 126], [128, 131], [133, 133]]


def test_qint32_weights_round_trip_through_state_dict():
    m = Shift1d(2, sparsity_term=0.)
    m.weight.data = torch.tensor([[300.0], [-2.0]])
    with pytest.warns(UserWarning, match="qint32"):
        q = QMODS.Shift1d.from_float(m)
    fresh = QMODS.Shift1d(2)
    fresh.load_state_dict(q.state_dict())
    assert fresh.qweight.dtype == torch.qint32 and fresh.qweight.int_repr().flatten().tolist() == [300, -2]


def test_quantize_shift_weights_edge_cases():
    # all weights equal: the reference's scale is ceil(0 / 255) = 0 and torch raises; here the shift is kept
    qw = quantize_shift_weights(torch.full((4, 2), 2.0))
    assert qw.dtype == torch.quint8 and (qw.int_repr().long() - qw.q_zero_point()).unique().tolist() == [2]
    # in range: identical to the reference's tensor (scale ceil(range / 255) = 1, zero point 128, quint8)
    w = torch.tensor([[-3.4, 2.5], [0.5, -1.5], [7.0, 0.0]])
    ref = torch.quantize_per_tensor(w, math.ceil((w.max().item() - w.min().item()) / 255.), 128, torch.quint8)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        got = quantize_shift_weights(w)
    assert torch.equal(got.int_repr(), ref.int_repr()) and got.q_zero_point() == 128
    assert (got.int_repr().long() - 128).tolist() == [[-3, 2], [0, -2], [7, 0]]  # round half to even, like the float path
    # range > 255: the reference's scale 2 would halve every shift (the kernel ignores the scale); qint32 keeps them
    w = torch.tensor([[-200.0, 3.0], [200.0, -1.0]])
    with pytest.warns(UserWarning):
        got = quantize_shift_weights(w)
    assert got.dtype == torch.qint32 and (got.int_repr().long() - got.q_zero_point()).tolist() == [[-200, 3], [200, -1]]
    with pytest.raises(ValueError):
        quantize_shift_weights(torch.tensor([[float("nan"), 0.0]]))


@pytest.mark.parametrize("cls,shape", [(Shift1d, (2, 3, 20)), (Shift2d, (2, 3, 9, 11)), (Shift3d, (1, 2, 5, 6, 7))])
def test_quantized_module_equals_float_module_on_the_grid(cls, shape):
    """integer shifts on integer-valued data: the quantized module (QuantizedCPU) equals the float module, with and
    without an average-pool tail (stride 2), qint32 weights included"""
    torch.manual_seed(3)
    for emulate, big in ((None, False), ({'kernel_size': 3, 'stride': 2, 'padding': 1}, False), (None, True)):
        m = cls(shape[1], padding='zeros', sparsity_term=0., emulate_dw=dict(emulate) if emulate else None)
        m.weight.data = torch.randint(-3, 4, m.weight.shape).float()
        if big:
            m.weight.data[0, 0] = 150.0
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            q = getattr(QMODS, cls.__name__).from_float(m)
        x = torch.randint(0, 64, shape).float()
        xq = torch.quantize_per_tensor(x, 1.0, 0, torch.quint8)
        ref = m(x)[0]
        out = q(xq)
        assert out.is_quantized and out.shape == ref.shape
        # pooled values: ATen rounds the window mean to the grid (half to even)
        assert torch.equal(out.dequantize(), torch.round(ref)), (cls.__name__, emulate, big)


def test_checkpoint_in_the_reference_layout_loads():
    """A checkpoint that holds only the float `weight` (the reference's quantized modules keep `qweight` as a plain
    attribute that state_dict() drops; round-1 checkpoints of this repo look the same): strict loading succeeds and the
    quantized shifts follow the LOADED weights, not the fresh module's initial ones."""
    torch.manual_seed(3)
    for cls, qcls, nd in ((Shift1d, QMODS.Shift1d, 1), (Shift2d, QMODS.Shift2d, 2), (Shift3d, QMODS.Shift3d, 3)):
        q = qcls(5, 'zeros')
        w = torch.tensor([[2.0, -3.0, 1.0], [0.5, 1.5, -2.5], [-1.2, 4.4, 0.0], [7.0, -7.0, 3.3], [0.0, 0.0, 0.0]])[:, :nd]
        legacy = {"weight": w.clone()}
        missing = q.load_state_dict(legacy, strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
        want = torch.round(w).to(torch.int64)
        got = q.qweight.int_repr().to(torch.int64) - q.qweight.q_zero_point()
        assert torch.equal(got, want)
        # inside a parent module too (prefix handling)
        parent = nn.Sequential(qcls(5, 'zeros'))
        parent.load_state_dict({"0.weight": w.clone()}, strict=True)
        got = parent[0].qweight.int_repr().to(torch.int64) - parent[0].qweight.q_zero_point()
        assert torch.equal(got, want)
    # a current-format checkpoint still wins over re-derivation
    q = QMODS.Shift2d(2, 'zeros')
    q.qweight = torch.quantize_per_tensor(torch.tensor([[3.0, -2.0], [1.0, 0.0]]), 1.0, 128, torch.quint8)
    sd = q.state_dict()
    q2 = QMODS.Shift2d(2, 'zeros')
    q2.load_state_dict(sd)
    assert torch.equal(q2.qweight.int_repr(), q.qweight.int_repr())


def test_quantized_avg_pool_restatement_matches_aten():
    """The arithmetic csrc/shiftnd_qpool.hip restates, against torch's own QuantizedCPU average pools over random windows,
    element types, zero points and ragged (ceil-mode) windows.  ATen rounds in two ways: its contiguous 1-D / 2-D kernel
    computes nearbyint(zp + sum(x - zp) * inv) in fp32, inv = 1 / (1 / float(1 / count)) -- the zero point INSIDE the
    rounding; its channels-last kernel (every 3-D tensor, and 4-D tensors that are channels-last-contiguous, which a
    contiguous tensor with one channel also is) computes nearbyint(sum * float(1 / count)) + zp.  The two differ when the
    zero point is odd and the mean is a tie; torch_binding.cpp: qpool_zp_outside picks per tensor."""
    import numpy as np
    torch.manual_seed(2)

    def both(acc, cnt, zp, lo, hi):
        acc = acc.astype(np.float32)
        mult = np.float32(1.0 / cnt)
        inv = np.float32(1.0) / (np.float32(1.0) / mult)
        inside = np.clip(np.rint(np.float32(zp) + acc * inv).astype(np.int64), lo, hi)
        outside = np.clip(np.rint(acc * mult).astype(np.int64) + zp, lo, hi)
        return inside, outside

    # (shape, channels_last, form ATen uses: 0 = inside, 1 = outside)
    cases = [((2, 3, 17, 19), False, 0), ((2, 3, 17, 19), True, 1), ((2, 1, 9, 11), False, 1), ((2, 3, 1, 13), False, 0),
             ((2, 3, 23), False, 0), ((3, 1, 23), False, 1), ((2, 3, 5, 7, 9), False, 1), ((2, 3, 5, 7, 9), True, 1)]
    separated = 0
    for shape, cl, form in cases:
        nd = len(shape) - 2
        for trial in range(8):
            qdt = torch.quint8 if trial % 2 == 0 else torch.qint8
            lo, hi = (0, 255) if qdt == torch.quint8 else (-128, 127)
            zp = int(torch.randint(lo, hi + 1, (1,)))
            k = tuple(int(torch.randint(1, 5, (1,))) for _ in range(nd))
            xi = torch.randint(lo, hi + 1, shape)
            xq = torch._make_per_tensor_quantized_tensor(xi.to(torch.uint8 if qdt == torch.quint8 else torch.int8), 0.0123, zp)
            if cl:
                xq = xq.contiguous(memory_format=torch.channels_last if nd == 2 else torch.channels_last_3d)
            pool = {1: torch.nn.functional.avg_pool1d, 2: torch.nn.functional.avg_pool2d, 3: torch.nn.functional.avg_pool3d}[nd]
            ref = pool(xq, k, k, 0, True).int_repr().numpy().astype(np.int64)
            x = xi.numpy().astype(np.int64)
            for idx in np.ndindex(*ref.shape[2:]):
                sl = tuple(slice(idx[r] * k[r], min((idx[r] + 1) * k[r], shape[2 + r])) for r in range(nd))
                win = x[(slice(None), slice(None)) + sl]
                cnt = int(np.prod(win.shape[2:]))
                acc = win.reshape(shape[0], shape[1], -1).sum(2) - cnt * zp
                forms = both(acc, cnt, zp, lo, hi)
                assert np.array_equal(forms[form], ref[(slice(None), slice(None)) + idx]), (shape, cl, qdt, zp, k, idx)
                separated += int((forms[0] != forms[1]).sum())
    assert separated > 0  # the two forms really differ on these inputs
