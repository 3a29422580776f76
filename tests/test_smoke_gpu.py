"""`__graft_entry__.smoke()` is what the driver runs on a fresh MI355X before the bench: it pins kernel names along the way, and a
routing change (round 5: the flat-stream kernels took the 14 x 14 stage it expected on the small-plane kernels) breaks it without
breaking any parity test.  Run it with the suite."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_graft_entry_smoke():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import __graft_entry__ as g
    from torchshifts import abi
    from test_routing_gpu import KNOB_DEFAULTS
    abi.set_path_policy(0)
    for knob, value in enumerate(KNOB_DEFAULTS):   # (other modules of the suite leave their knobs behind on this thread)
        abi.set_tuning(knob, value)
    assert g.smoke() is None
