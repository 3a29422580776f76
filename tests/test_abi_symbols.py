"""The C-ABI library loads (no GPU needed) and exports every symbol include/shiftnd_hip.h declares."""
import ctypes
import os
import re

import torch  # noqa: F401  (loads the HIP runtime torch ships before our library)

from torchshifts import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "shiftnd_hip.h")).read()
    return sorted(set(re.findall(r"SHIFTND_API[^;(]*?\b(shiftnd_\w+)\s*\(", text)))


def test_header_symbols_exported():
    decl = _declared_symbols()
    assert len(decl) == 20 and sorted(abi.EXPORTS) == decl
    L = ctypes.CDLL(abi._LIB_PATH)
    for name in decl:
        assert hasattr(L, name), name


def test_no_internal_symbols_leak():
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", abi._LIB_PATH], capture_output=True, text=True).stdout
    exported = [l.split()[-1] for l in out.splitlines() if " T " in l]
    assert sorted(exported) == sorted(abi.EXPORTS), exported


def test_host_only_entry_points():
    L = abi.lib()
    assert L.shiftnd_abi_version() == 5
    assert L.shiftnd_status_string(0) == b"ok" and L.shiftnd_status_string(-3) == b"workspace too small"
    assert abi.check_borders([2, 4, 6, 6], [[1, 2], [0, 1]], 2) == ([1, 4, 0, 5, 0, 1], [2, 4, 3, 5])
    assert abi.check_borders([2, 4, 6, 6], None, 2) == ([0, 6, 0, 6, 0, 1], [2, 4, 6, 6])
    assert abi.check_borders([2, 4, 6, 6], [[6, 0], [0, 6]], 2)[0] == [5, 6, 0, 1, 0, 1]
    # argument validation happens before any device work
    p = abi.Problem()
    p.ndim = 7
    st = (ctypes.c_int64 * 5)()
    assert L.shiftnd_forward(ctypes.byref(p), None, st, None, None, st, None) == -1
    p.ndim, p.dtype = 2, abi.I8
    assert L.shiftnd_forward(ctypes.byref(p), None, st, None, None, st, None) == -2  # float entry, int dtype
    # pooled sizes: ceil((r - l) / pool) per spatial dim
    x = torch.empty(2, 3, 13, 20)
    assert abi.pooled_shape(x, (2, 3)) == [2, 3, 7, 7]
    assert abi.pooled_shape(x, 4, [1, 12, 2, 19, 0, 1]) == [2, 3, 3, 5]
    p = abi.problem(x, 0, False, None)
    assert L.shiftnd_pooled_sizes(ctypes.byref(p), (ctypes.c_int32 * 2)(2, 0), (ctypes.c_int64 * 3)()) == -1


def test_check_borders_matches_oracle():
    from oracle import oracle as O
    import itertools
    for sizes in ([2, 3, 6, 6], [1, 2, 5, 7], [2, 2, 4]):
        nd = len(sizes) - 2
        for cuts in itertools.product(range(0, 8), repeat=2):
            user = [[cuts[0], cuts[1]]] * nd
            if cuts[0] > sizes[2]:
                continue  # negative size: the op raises
            assert abi.check_borders(sizes, user, nd) == tuple(O.check_borders(sizes, user, nd)), (sizes, user)
