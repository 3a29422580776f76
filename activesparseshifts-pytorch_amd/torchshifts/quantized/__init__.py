"""torchshifts.quantized: quantized modules + `quant_mapping` for torch.ao.quantization.convert.

The reference builds its mapping tables from torch.quantization internals that no longer exist on
torch >= 2.x (torchshifts/quantized/__init__.py:6-15 raises AttributeError there); this version
looks them up defensively so that importing the package never fails.
"""
import copy

import torch


def _default_static_mappings():
    for modname in ("torch.ao.quantization.quantization_mappings", "torch.quantization.quantization_mappings"):
        try:
            mod = __import__(modname, fromlist=["x"])
            return copy.copy(mod.get_default_static_quant_module_mappings())
        except Exception:  # pragma: no cover - depends on the torch build
            continue
    return {}


tndm_mapping = _default_static_mappings()

from .modules import Shift1d, Shift2d, Shift3d, new_quant_mapping  # noqa: E402,F401

quant_mapping = {**tndm_mapping, **new_quant_mapping}
