"""Quantized functional wrappers (reference torchshifts/quantized/functional.py:3-15)."""
from torchshifts.functional import shift1d_func, shift2d_func, shift3d_func


def _quantized(name, func, input, weight, padding_mode, cut_borders):
    if not input.is_quantized:
        raise ValueError(f"Input to '{name}' must be quantized!")
    return func(input, weight, padding_mode, False, cut_borders)


def shift1d_quantized(input, weight, padding_mode, cut_borders=None):
    return _quantized('shift1d_quantized', shift1d_func, input, weight, padding_mode, cut_borders)


def shift2d_quantized(input, weight, padding_mode, cut_borders=None):
    return _quantized('shift2d_quantized', shift2d_func, input, weight, padding_mode, cut_borders)


def shift3d_quantized(input, weight, padding_mode, cut_borders=None):
    return _quantized('shift3d_quantized', shift3d_func, input, weight, padding_mode, cut_borders)
