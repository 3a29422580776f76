"""Host-side arithmetic of the HIP library, checked on the CPU: the sweep kernels' arithmetic padding
map (canon_shift + fold_index, shiftnd_common.hpp) must equal the reference's infer_index for every
coordinate, shift and length -- exhaustively over small lengths, plus huge (multi-wrap) shifts."""
import itertools

from oracle import oracle as O
from torchshifts import abi


def _ref(p, s, length, pad):
    if length == 1:
        return 0
    m = O.infer_index(p - s, length, pad)
    return m if m >= 0 else -1


def test_arithmetic_map_equals_reference_exhaustive():
    L = abi.lib()
    for pad in range(5):
        for length in range(1, 14):
            for s in range(-4 * length - 5, 4 * length + 6):
                for p in range(0, length + 1):
                    assert L.shiftnd_debug_map(p, s, length, pad) == _ref(p, s, length, pad), (pad, length, s, p)


def test_arithmetic_map_huge_shifts():
    L = abi.lib()
    for pad, length, s in itertools.product(range(5), (2, 3, 7, 224, 1000), (10 ** 6 + 3, -(10 ** 6) - 7, 2 ** 31 + 5,
                                                                               -(2 ** 33) - 11, 2 ** 40 + 1, 2 ** 30, -(2 ** 30))):
        for p in (0, 1, length // 2, length - 1, length):
            assert L.shiftnd_debug_map(p, s, length, pad) == _ref(p, s, length, pad), (pad, length, s, p)


def test_workspace_query_tiny_shapes():
    """shiftnd_backward_workspace_bytes plans every kernel family on the host: rows shorter than one 16-byte chunk
    (W = 1..3 fp32) once divided by a zero chunk count there (SIGFPE).  Every dtype, 1-3 dims, sizes 1..5."""
    import torch
    for dt in (torch.float32, torch.float64, torch.float16, torch.bfloat16):
        for shape in itertools.chain(itertools.product((1, 3), (1, 4), (1, 2, 5)),
                                     itertools.product((1, 2), (1, 32), (1, 64), (1, 2, 3)),
                                     itertools.product((2,), (1, 3), (1, 2), (1, 5), (1, 2))):
            x = torch.zeros(*shape, dtype=dt)
            for pad in range(5):
                for active in (0, 1):
                    ws = abi.backward_workspace(x, pad, active)
                    assert ws.numel() >= shape[0] * shape[1] * 3 * 8 // 8  # at least one group of partial sums


KNOB_DEFAULTS = [0, 128 * 1024, 4, 2, 1, 0, 1, 0, 4, 512, 2, 256, -1, 0, 16, 0, 1, 0, 0, 0, 1, 0, 1, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0,
                 0, 0, 0]
PLAN_KNOBS = [(0, 1), (0, 1 << 20), (1, 1), (1, 1 << 30), (10, 1), (10, 64), (11, 1), (11, 4096), (13, 1 << 20), (14, 1), (21, 1),
              (25, 1), (26, 1), (38, 1)]


def _workspace_bytes(shape, dt, pad, active):
    import ctypes
    import torch
    p = abi.problem(torch.empty(shape, dtype=dt, device="meta"), pad, active, None)
    return int(abi.lib().shiftnd_backward_workspace_bytes(ctypes.byref(p)))


def test_workspace_bytes_cover_the_default_plan_under_any_knobs():
    """shiftnd_backward_workspace_bytes = max(plan under the default knobs, plan under the calling thread's knobs) (include/
    shiftnd_hip.h): whatever a thread has set, its answer is never below the answer of an untouched thread -- the size a
    backward on ANY thread can fall back to.  Launch-shaping knobs at both extremes; at least one of them must raise the
    answer (otherwise this test pins nothing), and a fresh thread always sees the defaults."""
    import threading
    import torch
    shapes = [(4, 64, 224, 224), (2, 4, 512, 512), (64, 256, 56, 56), (8, 128, 16, 112, 112), (3, 5, 7, 33, 65), (2, 16, 4096),
              (1, 1, 8192, 512), (16, 32, 14, 14)]
    raised = 0
    for shape in shapes:
        for dt in (torch.float32, torch.bfloat16):
            for active in (0, 1):
                base = _workspace_bytes(shape, dt, 0, active)
                for knob, value in PLAN_KNOBS:
                    abi.set_tuning(knob, value)
                    try:
                        touched = _workspace_bytes(shape, dt, 0, active)
                        seen = []
                        t = threading.Thread(target=lambda: seen.append(_workspace_bytes(shape, dt, 0, active)))
                        t.start()
                        t.join()
                    finally:
                        abi.set_tuning(knob, KNOB_DEFAULTS[knob])
                    assert touched >= base, (shape, dt, active, knob, value)
                    assert seen == [base], (shape, dt, active, knob, value)   # knobs are thread-local: a new thread plans with the defaults
                    raised += touched > base
                assert _workspace_bytes(shape, dt, 0, active) == base   # (restored)
    assert raised > 0


def test_no_kernel_of_the_built_library_uses_scratch():
    """build() refuses to link when a kernel needs a private segment (tools/kernel_resources.py over the AMDGPU metadata notes of
    every code object); this test repeats the check on the objects that are there and pins the tool itself: it must find the
    library's kernels (several hundred -- and, since round 5's consolidation, no more than 1320) and report the resources of a
    known one"""
    import glob
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    objs = sorted(glob.glob(os.path.join(root, "activesparseshifts-pytorch_amd", "build", "*.hip.o")))
    if not objs:
        import pytest
        pytest.skip("no built objects (run __graft_entry__.build() first)")
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(root, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    assert kr.check_no_scratch(objs) == []
    rows = kr.collect(objs)
    assert 600 < len(rows) <= 1320, len(rows)   # (round 6: 998 + the pooled / two-row-group forms of crop_backward, crop_backward3 (+ pooled), crop_forward3 (3-D, 2-D rows, pooled), the pooled row kernels, the 16-bit and interpolating pooled gather forwards, the 30 element-wide plane kernels that emptied the strided tail of the route census, the cropped walks)
    assert any("walk_backward16" in r["demangled"] for r in rows)


def test_no_wide_store_with_register_soffset_is_followed_by_a_write_of_its_data():
    """DESIGN section 9 / ADVICE r05: `buffer_store_dwordx4 v[a:b], .., s<N>` followed in the next issue slot by a VALU write of
    v[a:b] stores corrupted data on gfx950 and LLVM does not insert the wait state for a register soffset.  The helper puts
    `s_nop 1` behind such stores; this reads the disassembly of every built code object so that a new kernel calling the builtin
    directly, or a compiler that moves the asm's operand copy, fails here and not as rare zeros in grad_x.  The scanner itself is
    pinned on a synthetic listing."""
    import glob
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("isa_store_hazard", os.path.join(root, "tools", "isa_store_hazard.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    assert sh.STORE.match("buffer_store_dwordx4 v[2:5], v73, s[24:27], s56 offen nt").group(4) == "s56"
    assert sh.dest_vgprs("v_cndmask_b32_e64 v2, 0, -1, s[16:17]") == {2}
    assert sh.dest_vgprs("v_pk_mul_f32 v[4:5], v[8:9], v[10:11]") == {4, 5}
    assert sh.dest_vgprs("v_cmp_lt_i32_e64 s[2:3], v4, v5") == set()
    objs = sorted(glob.glob(os.path.join(root, "activesparseshifts-pytorch_amd", "build", "*.hip.o")))
    if not objs:
        import pytest
        pytest.skip("no built objects (run __graft_entry__.build() first)")
    stores, bad = sh.check(objs)
    assert stores > 50, stores     # the walk / span / step kernels store rows at an SGPR plane offset
    assert bad == [], bad[:5]
