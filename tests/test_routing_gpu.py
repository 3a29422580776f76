"""Which kernel serves each BASELINE config and each bench.py workload with ALL tuning knobs at their defaults -- the route a
user of the drop-in gets.  The parity tests force kernel families through knobs; a routing regression would otherwise only show
up as a slower bench line.  Small batch sizes (the choice depends on the plane geometry, the dtype and the alignment, not on N)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# workload -> (shape with a small batch, dtype, active, cut, forward kernel, backward kernel)
ROUTES = {
    "c2": ((2, 256, 224, 224), torch.float32, False, None, "step_gather_forward", "step_backward"),
    "c2a": ((2, 256, 224, 224), torch.float32, True, None, "step_active_forward", "step_backward"),
    "c3": ((1, 128, 16, 112, 112), torch.bfloat16, True, None, "walk_forward16", "walk_backward16"),
    "c3 sparse": ((1, 128, 16, 112, 112), torch.bfloat16, False, None, "step_gather_forward_lds", "walk_backward16_sparse"),
    "c3 fp32": ((1, 128, 16, 112, 112), torch.float32, True, None, "walk_forward", "walk_backward"),
    "c5": ((2, 512, 224, 224), torch.float16, False, None, "step_gather_forward_small", "step_backward"),
    "c2crop": ((2, 256, 224, 224), torch.float32, False, [[1, 1], [1, 1]], "crop_gather_forward", "crop_backward"),
    "c2acrop": ((2, 256, 224, 224), torch.float32, True, [[1, 1], [1, 1]], "crop_active_forward", "crop_backward"),
    "t1": ((8, 16, 64, 64), torch.float32, False, [[1, 1], [1, 1]], "crop_gather_forward", "crop_backward"),
    "t1a": ((8, 16, 64, 64), torch.float32, True, [[1, 1], [1, 1]], "crop_active_forward", "crop_backward"),
    "c1d": ((4, 512, 4096), torch.float32, False, None, "row_gather_forward", "row_backward"),
    "c1da": ((4, 512, 4096), torch.float32, True, None, "row_active_forward", "row_backward"),
    "c1dh": ((4, 512, 4096), torch.float16, False, None, "row_gather_forward", "row_backward"),
}


@pytest.mark.parametrize("name", sorted(ROUTES))
def test_default_route(name):
    from torchshifts import abi
    shape, tdt, active, cut, fwd, bwd = ROUTES[name]
    abi.set_path_policy(0)   # (no knob is touched: the library's defaults; the other test modules restore theirs)
    nd = len(shape) - 2
    b, oshape = abi.check_borders(list(shape), cut, nd) if cut else (None, list(shape))
    x = torch.rand(shape, device=DEV).to(tdt)
    go = torch.rand(oshape, device=DEV).to(tdt)
    w = (torch.rand(shape[1], nd, device=DEV) * 6 - 3).to(tdt)
    for pad in range(5):
        abi.forward(x, w, pad, active, b)
        if fwd is not None and not (name == "c5" and pad != 0) and not (name == "c3 sparse" and False):
            assert abi.last_kernel() == fwd, (name, pad, abi.last_kernel())
        abi.backward(go, w, x, pad, active, b)
        assert abi.last_kernel() == bwd, (name, pad, abi.last_kernel())


def test_default_route_c4_quantized():
    from torchshifts import abi
    abi.set_path_policy(0)
    xq = torch.randint(0, 255, (128, 512, 56, 56), dtype=torch.uint8, device=DEV)   # (the byte kernel wants enough planes per channel)
    wq = (torch.rand(512, 2, device=DEV) * 6 - 3).round().add(128).to(torch.uint8)
    for pad in range(5):
        abi.forward_quantized(xq, wq, 128, 0, pad)
        assert abi.last_kernel() == "bytes_gather_forward", (pad, abi.last_kernel())
