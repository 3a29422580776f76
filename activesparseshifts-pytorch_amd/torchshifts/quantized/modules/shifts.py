"""Quantized shift modules (reference torchshifts/quantized/modules/shifts.py).

`forward` returns the tensor only (no loss).  `from_float` converts a float Shift{N}d; the integer
shifts are `int_repr(qweight) - zero_point`, the scale is not used by the kernel
(kernels/shifts_kernels.h:553-555).

Differences from the reference:
  * the quantized weights checkpoint and move with the module: their integer representation is a buffer
    (`qweight_repr`), scale / zero point / dtype travel as the module's extra state; `qweight` is a property that
    rebuilds the quantized tensor on the buffer's device (the reference keeps a plain attribute, :17, that
    `state_dict()` drops and `.to(device)` leaves behind);
  * `quantize_shift_weights` never changes a shift: the reference's `scale = ceil(range / 255)` (:10-12) is 0 when all
    weights are equal (torch raises) and > 1 when the range exceeds 255, which silently HALVES every shift because
    the kernel ignores the scale.  Here the scale is always 1; weights that do not fit quint8 around zero point 128
    (|round(w)| > 127) become qint32 with zero point 0, which both backends accept.
"""
import warnings

import torch

import torchshifts.modules.shifts as shifts
from torchshifts.functional import shift1d_pool_func, shift2d_pool_func, shift3d_pool_func
from torchshifts.quantized.functional import shift1d_quantized, shift2d_quantized, shift3d_quantized

_POOL_FUNCS = {1: shift1d_pool_func, 2: shift2d_pool_func, 3: shift3d_pool_func}

rp_dict = {v: k for k, v in shifts.paddings_dict.items()}


def quantize_shift_weights(weight):
    """Float shift weights -> quantized tensor whose `int_repr - zero_point` is round-half-even(weight).

    quint8, scale 1, zero point 128 (the reference's layout whenever the reference is right); qint32, scale 1, zero
    point 0 when a rounded shift leaves [-128, 127]."""
    w = weight.detach().float()
    if w.numel() and not bool(torch.isfinite(w).all()):
        raise ValueError("quantize_shift_weights: weights must be finite")
    amax = float(w.abs().max()) if w.numel() else 0.0
    if amax <= 127.0:
        return torch.quantize_per_tensor(w, 1.0, 128, torch.quint8)
    warnings.warn("shift weights beyond +-127 do not fit quint8 around zero point 128: using qint32 weights "
                  "(the reference would have coarsened every shift by ceil(range / 255))", stacklevel=2)
    return torch.quantize_per_tensor(w, 1.0, 0, torch.qint32)


class _QuantizedShiftMixin:
    _qfunc = None
    _qname = None

    def _init_quantized(self):
        self.register_buffer("qweight_repr", torch.zeros(self.in_channels, self.dim, dtype=torch.uint8))
        self._q_scale, self._q_zero_point, self._q_dtype = 1.0, 128, "quint8"
        self.qweight = quantize_shift_weights(self.weight.float())

    # ---- the quantized weights: buffer + extra state ---------------------------------------------------------
    @property
    def qweight(self):
        return torch._make_per_tensor_quantized_tensor(self.qweight_repr, self._q_scale, self._q_zero_point)

    @qweight.setter
    def qweight(self, q):
        if not q.is_quantized:
            raise ValueError("qweight must be a per-tensor quantized tensor")
        self._q_scale, self._q_zero_point = float(q.q_scale()), int(q.q_zero_point())
        self._q_dtype = str(q.dtype).replace("torch.", "")
        self.qweight_repr = q.int_repr()  # (an existing buffer name: nn.Module keeps it registered)

    def get_extra_state(self):
        return {"qweight_scale": self._q_scale, "qweight_zero_point": self._q_zero_point, "qweight_dtype": self._q_dtype}

    def set_extra_state(self, state):
        self._q_scale = float(state["qweight_scale"])
        self._q_zero_point = int(state["qweight_zero_point"])
        self._q_dtype = state["qweight_dtype"]
        want = {"quint8": torch.uint8, "qint8": torch.int8, "qint32": torch.int32}[self._q_dtype]
        if self.qweight_repr.dtype != want:  # load_state_dict copies INTO the existing buffer: give it the saved type
            self.qweight_repr = self.qweight_repr.to(want)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        key, extra = prefix + "qweight_repr", prefix + "_extra_state"
        legacy = key not in state_dict and (prefix + "weight") in state_dict
        if key in state_dict and state_dict[key].dtype != self.qweight_repr.dtype:
            self.qweight_repr = torch.zeros_like(self.qweight_repr, dtype=state_dict[key].dtype)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)
        if legacy:
            # a checkpoint in the reference's layout (only the float `weight`: its `qweight` is a plain attribute that
            # state_dict() drops, quantized/modules/shifts.py:17): the quantized weights are what from_float would make
            # of the loaded float weights -- never the fresh module's initial shifts
            self.qweight = quantize_shift_weights(self.weight.detach().float())
            for k in (key, extra):
                while k in missing_keys:
                    missing_keys.remove(k)

    def forward(self, input):
        if self._pool_size is not None and input.is_cuda:
            # ATen has no QuantizedCUDA average pool: torchshifts::shift{N}d_pool on the QuantizedCUDA key runs the shift and
            # the pool as one pass (int8 / uint8) with the QuantizedCPU pool's arithmetic -- the values _reduction_fn gives
            # on CPU tensors, ragged last windows included
            if not input.is_quantized:
                raise ValueError(f"Input to '{self._qname}' must be quantized!")
            return _POOL_FUNCS[self.dim](input, self.qweight, self.padding, False, self.cut_borders, self._pool_size)
        out = type(self)._qfunc(input, self.qweight, self.padding, self.cut_borders)
        return self._reduction_fn(out)

    def _get_name(self):
        return self._qname

    @classmethod
    def from_float(cls, mod):
        qshift = cls(mod.in_channels, rp_dict[mod.padding])
        qshift.cut_borders = mod.cut_borders
        qshift._reduction_fn = mod._reduction_fn
        qshift._pool_size = getattr(mod, "_pool_size", None)
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
