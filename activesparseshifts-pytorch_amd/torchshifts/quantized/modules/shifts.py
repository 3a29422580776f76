"""Quantized shift modules (reference torchshifts/quantized/modules/shifts.py).

`forward` returns the tensor only (no loss).  `from_float` converts a float Shift{N}d; the integer
shifts are `int_repr(qweight) - zero_point`, the scale is not used by the kernel
(kernels/shifts_kernels.h:553-555).
"""
import math

import torch

import torchshifts.modules.shifts as shifts
from torchshifts.quantized.functional import shift1d_quantized, shift2d_quantized, shift3d_quantized

rp_dict = {v: k for k, v in shifts.paddings_dict.items()}


def quantize_shift_weights(weight):
    """quint8, zero point 128, integer scale ceil(range / 255) (reference :10-12).

    QUIRK kept: ranges above 255 coarsen the shifts, and max == min gives scale 0 (torch raises)."""
    scale = math.ceil((weight.max().item() - weight.min().item()) / 255.)
    return torch.quantize_per_tensor(weight, scale, 128, torch.quint8)


class _QuantizedShiftMixin:
    _qfunc = None
    _qname = None

    def _init_quantized(self):
        self.qweight = quantize_shift_weights(self.weight.float())

    def forward(self, input):
        return self._reduction_fn(type(self)._qfunc(input, self.qweight, self.padding, self.cut_borders))

    def _get_name(self):
        return self._qname

    @classmethod
    def from_float(cls, mod):
        qshift = cls(mod.in_channels, rp_dict[mod.padding])
        qshift.cut_borders = mod.cut_borders
        qshift._reduction_fn = mod._reduction_fn
        qshift.weight = mod.weight
        qshift.qweight = quantize_shift_weights(mod.weight.float())
        return qshift


class Shift1d(_QuantizedShiftMixin, shifts.Shift1d):
    _qfunc = staticmethod(shift1d_quantized)
    _qname = 'QuantizedShift1D'

    def __init__(self, in_channels, padding='zeros'):
        super().__init__(in_channels, padding, 1, 0, False)
        self._init_quantized()


class Shift2d(_QuantizedShiftMixin, shifts.Shift2d):
    _qfunc = staticmethod(shift2d_quantized)
    _qname = 'QuantizedShift2D'

    def __init__(self, in_channels, padding='zeros'):
        super().__init__(in_channels, padding, 1, 0, False)
        self._init_quantized()


class Shift3d(_QuantizedShiftMixin, shifts.Shift3d):
    _qfunc = staticmethod(shift3d_quantized)
    _qname = 'QuantizedShift3D'

    def __init__(self, in_channels, padding='zeros'):
        super().__init__(in_channels, padding, 1, 0, False)
        self._init_quantized()
