"""A seeded, COUNT-boxed slice of tools/fuzz_round2.py in the GPU suite: random shapes, paddings, shift kinds, launch-planning
knobs and dtypes through the round-2 / round-3 kernel families (LDS-tiled channels-last, small planes / row bands, the
byte kernel, the one-step kernels, the 3-D walk kernels incl. the pooled backward; round 4: crops and Shift1d through the crop / row kernels,
channels-last windows), every case against the CPU oracle (bit-exact gathers and fp32 / fp64 interpolation,
1 ulp for 16-bit interpolation, grad_w 1e-5 / 16-bit epsilon of the fp64 evaluation).  The standalone tool runs the same
cases for as long as asked."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_cases_against_the_oracle(seed):
    import fuzz_round2 as F
    assert torch.cuda.is_available()
    rs = np.random.RandomState(1000 + seed)
    F.count.clear()
    # a fixed number of cases per seed (round-5 verdict: a wall-clock box made the number of cases the driver ran depend on the
    # box): ROUNDS passes over the case list, the same cases on every box (~18 s per seed on an MI355X box: 6 s for 20 rounds measured)
    ROUNDS = 60
    for n in range(ROUNDS * len(F.CASES)):
        F.CASES[n % len(F.CASES)](rs)
    kernels = set(F.count)
    for must in ("cl_tiled_backward", "step_backward", "step_gather_forward", "walk_forward", "walk_backward", "walk_backward16", "walk_backward_pool",
                 "crop_backward", "cl_tiled_backward/crop", "cl_tiled_forward_3d", "crop_backward3", "crop_active_forward3"):
        assert must in kernels, (must, dict(F.count))
    assert any(k.startswith(("small_", "band_")) for k in kernels), dict(F.count)
