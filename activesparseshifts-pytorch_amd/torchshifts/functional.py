"""Functional interface: shift{1,2,3}d_func (same names, arguments and checks as the reference's
torchshifts/functional.py:7-99).  Each call validates its arguments with `assert` (AssertionError,
like the reference) and forwards to the dispatcher op `torch.ops.torchshifts.shift{N}d`, which on a
HIP tensor runs the gfx950 kernels of libshiftnd_hip.so.
"""
from typing import Optional

import torch

from .extension import _assert_has_ops

Tensor = torch.Tensor

_PADDING_DOC = "0 - zeros, 1 - border, 2 - periodic, 3 - reflect, 4 - symmetric"


def _shift_func(dim: int, input: Tensor, weights: Tensor, padding_mode: int, active_flag: bool,
                borders: Optional[Tensor], pool=None) -> Tensor:
    name = f"shift{dim}d_func()" if pool is None else f"shift{dim}d_pool_func()"
    _assert_has_ops()
    assert padding_mode in [0, 1, 2, 3, 4], f"{name} expected padding_mode can be {_PADDING_DOC}"
    assert len(input.shape) == dim + 2, f"{name}: expected {dim + 2}D tensor as input, but it is shape is {input.shape}"
    assert weights.shape[-1] == dim, \
        f"{name}: expected [n_channels,{dim}] tensor as weight, but it is shape is {weights.shape}"
    assert input.shape[1] == weights.shape[0], \
        (f"{name}: expected that input and weight have equal number of channels, but input have "
         f"{input.shape[1]} and weight have {weights.shape[0]} channels.")
    assert input.device == weights.device, \
        (f"{name}: expected input and weights to be on same device, but input is  on {input.device} "
         f"and weights is on {weights.device}")
    if borders is not None:
        assert (len(borders.shape) == 2) and (borders.shape[1] == 2) and (borders.shape[0] == dim), \
            f"borders must have shape [{dim}, 2]"
    else:
        borders = torch.Tensor()
    if pool is not None:
        if isinstance(pool, torch.Tensor):
            pool = pool.reshape(-1).tolist()
        pool = [int(pool)] * dim if isinstance(pool, (int, float)) else [int(k) for k in pool]
        assert len(pool) == dim and all(k >= 1 for k in pool), f"{name}: pool must be {dim} window sizes >= 1"
        return getattr(torch.ops.torchshifts, f"shift{dim}d_pool")(input, weights, borders, pool, padding_mode, active_flag)
    return getattr(torch.ops.torchshifts, f"shift{dim}d")(input, weights, borders, padding_mode, active_flag)


def shift1d_func(input: Tensor, weights: Tensor, padding_mode: int, active_flag: bool,
                 borders: Optional[Tensor] = None) -> Tensor:
    """Shift a [N, C, H] tensor; weights [C, 1]; borders [1, 2] = (cut_left, cut_right)."""
    return _shift_func(1, input, weights, padding_mode, active_flag, borders)


def shift2d_func(input: Tensor, weights: Tensor, padding_mode: int, active_flag: bool,
                 borders: Optional[Tensor] = None) -> Tensor:
    """Shift a [N, C, H, W] tensor; weights [C, 2] (H, W); borders [2, 2]."""
    return _shift_func(2, input, weights, padding_mode, active_flag, borders)


def shift3d_func(input: Tensor, weights: Tensor, padding_mode: int, active_flag: bool,
                 borders: Optional[Tensor] = None) -> Tensor:
    """Shift a [N, C, H, W, D] tensor; weights [C, 3] (H, W, D); borders [3, 2]."""
    return _shift_func(3, input, weights, padding_mode, active_flag, borders)


# ---- shift + average pool as one op (not in the reference) -----------------------
# `pool` is the window (= stride) per spatial dim, an int or a list.  The result equals
# avg_pool{N}d(shift{N}d_func(...), kernel_size=pool, stride=pool, ceil_mode=True), the tail the reference's
# modules attach when they emulate a strided depthwise conv (modules/shifts.py:81-89, 150-153).  On HIP tensors
# the full-size shift output is never written to memory; on CPU tensors the op is literally that sequence.
def shift1d_pool_func(input: Tensor, weights: Tensor, padding_mode: int, active_flag: bool,
                      borders: Optional[Tensor] = None, pool=2) -> Tensor:
    return _shift_func(1, input, weights, padding_mode, active_flag, borders, pool)


def shift2d_pool_func(input: Tensor, weights: Tensor, padding_mode: int, active_flag: bool,
                      borders: Optional[Tensor] = None, pool=2) -> Tensor:
    return _shift_func(2, input, weights, padding_mode, active_flag, borders, pool)


def shift3d_pool_func(input: Tensor, weights: Tensor, padding_mode: int, active_flag: bool,
                      borders: Optional[Tensor] = None, pool=2) -> Tensor:
    return _shift_func(3, input, weights, padding_mode, active_flag, borders, pool)
