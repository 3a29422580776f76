"""Locate and load the native operator library of this package.

Mirrors the role of the reference's torchshifts/extension.py:39-68: the shared object `_C.so`
that sits next to this file is loaded with `torch.ops.load_library`, which registers the
`torchshifts::*` dispatcher ops.  `_C.so` links against `libshiftnd_hip.so` (same directory),
the C-ABI library that holds the gfx950 kernels.

There is no Python/eager fallback: if the library cannot be loaded every functional entry point
raises through `_assert_has_ops()`.
"""
import os

_HAS_OPS = False
error_str = ""
_LIB_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_LIB_DIR, "_C.so")
_HIP_LIB_PATH = os.path.join(_LIB_DIR, "libshiftnd_hip.so")


def _register_extensions():
    import torch  # imported first so that its bundled libamdhip64 is the HIP runtime of the process

    if not os.path.exists(_LIB_PATH):
        raise ImportError("native library not built: %s (run activesparseshifts-pytorch_amd/build.py)" % _LIB_PATH)
    torch.ops.load_library(_LIB_PATH)


try:
    _register_extensions()
    _HAS_OPS = True
except (ImportError, OSError) as e:  # same exception set as the reference (extension.py:54)
    error_str = str(e)


def _has_ops():
    return _HAS_OPS


def _assert_has_ops():
    if not _has_ops():
        raise RuntimeError(
            "Couldn't load custom C++ ops. The torchshifts HIP extension (_C.so / libshiftnd_hip.so) is "
            "missing or was built against a different PyTorch; rebuild it with "
            "activesparseshifts-pytorch_amd/build.py. There is no Python fallback."
            f"\n\nImport error details:\n\t{error_str}")


def _check_cuda_version():
    """Kept for API compatibility (reference extension.py:71-96).

    `torchshifts::_cuda_version` returns -1 (no CUDA toolkit); the HIP build version is available
    from `torchshifts::_hip_version`.  The CUDA cross-check only applies when torch itself was
    built with CUDA, which is never the case for this ROCm-only package.
    """
    if not _HAS_OPS:
        return -1
    import torch

    version = torch.ops.torchshifts._cuda_version()
    if version != -1 and torch.version.cuda is not None:
        raise RuntimeError("this torchshifts build targets ROCm/HIP (gfx950) and cannot be used with a CUDA PyTorch")
    return version


_check_cuda_version()
