"""torchshifts -- MI355X-native drop-in for the `torchshifts` package of
DeadAt0m/ActiveSparseShifts-PyTorch (same import name, op names, functional wrappers and modules;
reference: torchshifts/__init__.py:17-18).
"""
from .extension import _HAS_OPS  # noqa: F401  (loads _C.so, registering torch.ops.torchshifts.*)

__version__ = "3.1+mi355x"

from torchshifts.modules import Shift1d, Shift2d, Shift3d  # noqa: E402,F401
from torchshifts.quantized import quant_mapping  # noqa: E402,F401
