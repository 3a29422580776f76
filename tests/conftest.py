"""pytest configuration: registers the `gpu` marker and puts the repo packages on sys.path.

`-m "not gpu"` runs on the CPU-only build container (oracle vs golden fixtures, host logic, C-ABI
symbol checks, gloo world_size-2 tests); `-m gpu` runs on an MI355X box and exercises the HIP
kernels through the C ABI and through the torch dispatcher.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "activesparseshifts-pytorch_amd")
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # outputs allocated by the ctypes binding start from a byte pattern: an element no kernel wrote fails deterministically
    from torchshifts import abi
    abi.POISON = True
