"""Shift1d / Shift2d / Shift3d modules (surface of the reference's torchshifts/modules/shifts.py).

Constructor arguments, attribute names (`weight`, `padding`, `cut_borders`, `init_shift`,
`sparsity_term`, ...), the `(output, loss)` return convention and the depthwise-conv emulation
heuristics follow the reference so that checkpoints and calling code carry over unchanged.
Behavioural quirks of the reference that user code may depend on are kept and marked `QUIRK`.
"""
import random
from functools import partial

import torch
from torch import nn

from torchshifts.functional import (shift1d_func, shift1d_pool_func, shift2d_func, shift2d_pool_func, shift3d_func,
                                    shift3d_pool_func)

paddings_dict = {'zeros': 0, 'border': 1, 'periodic': 2, 'reflect': 3, 'symmetric': 4}

_SHIFT_FUNCS = {1: shift1d_func, 2: shift2d_func, 3: shift3d_func}
_SHIFT_POOL_FUNCS = {1: shift1d_pool_func, 2: shift2d_pool_func, 3: shift3d_pool_func}
_AVG_POOLS = {1: torch.nn.functional.avg_pool1d, 2: torch.nn.functional.avg_pool2d, 3: torch.nn.functional.avg_pool3d}


def _wrap_dim(val, dim, name):
    """scalar / tuple / list -> list of length `dim` (reference modules/shifts.py:10-18)."""
    if isinstance(val, tuple):
        val = list(val)
    if not isinstance(val, list):
        val = [val] * dim
    if len(val) != dim:
        print(f'{name} params has different kernel sizes, but length of list do not corresponds to dim: {dim}, '
              'and was reduced')
        val = val[:dim]
    return val


def _create_dw_emulation(args, dim):
    """Heuristics that make a shift layer mimic a depthwise conv (reference modules/shifts.py:21-57).

    Returns (init_shift, scales, borders, padding):
      borders  [dim, 2] long tensor of (cut_left, cut_right) when 2*padding - kernel_size + 1 < 0
               (the conv would shrink the output), else None
      init_shift = kernel_size // (2 if init_thumb_rule_type == 1 else 1)
      scales   = stride as a [1, dim] tensor (used both as a weight scale and as the avg-pool size)
      padding  = conv padding_mode translated to the shift numbering, or -1
    """
    assert isinstance(args, dict), 'args must be dict'
    assert 'kernel_size' in args, 'args must contains at least the kernel_size inside'
    if 'dilation' in args:
        print('Warning! Found the dilation param which is not supported and will be ignored')
    kernel_size = torch.tensor(_wrap_dim(args['kernel_size'], dim, 'kernel_size'), requires_grad=False)
    padding = torch.tensor(_wrap_dim(args.get('padding', 0), dim, 'padding'), requires_grad=False)
    stride = _wrap_dim(args.get('stride', 1), dim, 'stride')
    itrt_scale = 2 if args['init_thumb_rule_type'] == 1 else 1

    borders = None
    shrink = 2 * padding - kernel_size + 1
    if (shrink < 0).any():
        borders = torch.zeros(dim, 2, dtype=torch.long, requires_grad=False)
        neg = shrink < 0
        borders[neg, 0] = abs(shrink[neg]) // 2
        borders[neg, 1] = abs(shrink[neg]) - borders[neg, 0]

    init_shift = kernel_size // itrt_scale
    scales = torch.tensor(stride, requires_grad=False).unsqueeze(0)

    pad_conv = {'zeros': 0, 'replicate': 1, 'circular': 2, 'reflect': 3}
    pad_mode = args.get('padding_mode', -1)
    if isinstance(pad_mode, str):
        pad_mode = pad_conv[pad_mode]
    return init_shift, scales, borders, pad_mode


class _Shiftnd(nn.Module):
    """Base of all shift modules.

    Arguments:
        in_channels (int): channels of the input.
        padding (str): 'zeros' | 'border' | 'periodic' | 'reflect' | 'symmetric'. Default 'zeros'.
        init_shift (float / tuple): bound of the uniform weight initialisation. Default 1.
        sparsity_term (float): strength of the L1 sparsity loss returned by forward. Default 5e-4.
        active_flag (bool): interpolate on the forward pass (active shift). Default False.
        emulate_dw (dict): parameters of the depthwise conv this layer replaces (kernel_size,
            stride, padding, padding_mode); output cropping / pooling are derived from them.
        init_thumb_rule (int): 1: uniform(-init_shift, init_shift); 2: uniform(0, init_shift) * sign.
    """
    dim = None

    @staticmethod
    def _identity(x):
        return x

    @staticmethod
    def _pooling(ks, dim):
        if isinstance(ks, torch.Tensor):
            ks = ks.squeeze().cpu().numpy().tolist()
        return partial(_AVG_POOLS[min(dim, 3)], kernel_size=ks, stride=ks, ceil_mode=True)

    @staticmethod
    def _init_thumb_rule_1(size, shape):
        return 2 * size * torch.rand(shape) - size

    @staticmethod
    def _init_thumb_rule_2(size, shape):
        return size * torch.rand(shape) * (1 if random.random() < 0.5 else -1)

    def __init__(self, in_channels, padding='zeros', init_shift=1, sparsity_term=5e-4, active_flag=False,
                 emulate_dw=None, init_thumb_rule=1):
        super().__init__()
        assert padding.lower() in paddings_dict.keys(), f'incorrect padding option: {padding}'
        self.padding = paddings_dict[padding]
        self.sparsity_term = sparsity_term
        self.in_channels = in_channels
        self._active_flag = active_flag
        self._shift_func = self._init_shift_fn()
        self.cut_borders = None
        self._reduction_fn = self._identity
        self._pool_size = None  # window (= stride) of the average pool that follows the shift, when there is one
        # QUIRK (reference modules/shifts.py:117-118): rule 2 is selected with `==` instead of `=`,
        # so both rule numbers initialise with rule 1.  Kept: initial weights match the reference.
        self._w_init_func = self._init_thumb_rule_1
        self.init_shift = torch.tensor(_wrap_dim(init_shift, self.dim, 'init_shift'), requires_grad=False)
        self._w_post_init_scale = torch.ones(1, self.dim, requires_grad=False)

        if emulate_dw is not None:
            emulate_dw['init_thumb_rule_type'] = init_thumb_rule  # QUIRK: the caller's dict is modified (:125)
            self.init_shift, self._w_post_init_scale, self.cut_borders, _conv_padding = \
                _create_dw_emulation(emulate_dw, self.dim)
            # QUIRK (:128-129): the conv's padding_mode is compared, not assigned -> self.padding is unchanged.
            if not (self._w_post_init_scale == 1).all():
                self._reduction_fn = self._pooling(self._w_post_init_scale, self.dim)
                self._pool_size = [int(k) for k in self._w_post_init_scale.reshape(-1).tolist()]
        self._init_weights()

    def _init_shift_fn(self):
        if self.dim not in _SHIFT_FUNCS:
            raise NotImplementedError
        return _SHIFT_FUNCS[self.dim]

    def _init_weights(self):
        self.weight = nn.Parameter(torch.Tensor(self.in_channels, self.dim))
        self.reset_parameters()

    def reset_parameters(self):
        for i in range(self.dim):
            self.weight.data[:, i] = self._w_init_func(self.init_shift[i], self.in_channels)
        self.weight.data *= self._w_post_init_scale

    def _compute_weight_loss(self):
        return self.sparsity_term * torch.sum(torch.abs(self.weight))

    def forward(self, input):
        """Returns (output, loss); loss is None when sparsity_term == 0."""
        loss = self._compute_weight_loss() if bool(self.sparsity_term) else None
        if self._pool_size is not None and self.dim in _SHIFT_POOL_FUNCS:
            # reference: self._reduction_fn(shift(x)) (modules/shifts.py:150-153).  Same values from one op: on HIP
            # tensors the shift output is pooled on the fly instead of being written and re-read
            return _SHIFT_POOL_FUNCS[self.dim](input, self.weight, self.padding, self._active_flag, self.cut_borders,
                                               self._pool_size), loss
        out = self._shift_func(input, self.weight, self.padding, self._active_flag, self.cut_borders)
        return self._reduction_fn(out), loss

    def extra_repr(self):
        pad = {v: k for k, v in paddings_dict.items()}[self.padding]
        active = f'Active shift on forward pass: {"Yes" if self._active_flag else "No"}'
        sparse = ('Sparse shift: Yes - sparsity strength: {}'.format(self.sparsity_term)
                  if bool(self.sparsity_term) else 'Sparse shift: No')
        return f'in_channels={self.in_channels}, padding_method={pad}, {active}, {sparse}'


class Shift1d(_Shiftnd):
    """Learnable per-channel shift of a [N, C, H] tensor (zero-FLOP stand-in for a depthwise conv).
    forward(x) -> (out, loss).  Arguments: see _Shiftnd."""
    dim = 1


class Shift2d(_Shiftnd):
    """Learnable per-channel (H, W) shift of a [N, C, H, W] tensor.  forward(x) -> (out, loss)."""
    dim = 2


class Shift3d(_Shiftnd):
    """Learnable per-channel (H, W, D) shift of a [N, C, H, W, D] tensor.  forward(x) -> (out, loss)."""
    dim = 3
