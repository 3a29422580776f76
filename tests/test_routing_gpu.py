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


# -- sizing on one thread, running on another ----------------------------------------------------------------------------------
KNOB_DEFAULTS = [0, 128 * 1024, 4, 2, 1, 0, 1, 0, 4, 512, 2, 256, -1, 0, 16, 0, 1, 0, 0, 0, 1, 0, 1, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0,
                 0, 0, 0]
# (shape, dtype, channels-last, knob settings of the running thread); the knobs shape the launch plan, and with it the number of
# partial-sum records a family needs -- knob 21 = 1 (one row per band of the NHWC kernel) needs 8-20 x the default plan's bytes
CROSS_THREAD = [
    ((4, 64, 56, 56), torch.float32, True, [(21, 1)]),
    ((4, 64, 56, 56), torch.bfloat16, True, [(21, 1)]),
    ((2, 4, 96, 512), torch.float32, False, [(11, 1), (10, 1)]),
    ((4, 32, 56, 56), torch.float32, False, [(0, 1 << 20), (1, 1), (32, 1), (12, 0)]),
    ((2, 8, 6, 40, 48), torch.float32, False, [(0, 1 << 20), (1, 1), (38, 1)]),
    ((8, 16, 14, 14), torch.float32, False, [(25, 1), (26, 1)]),
]


@pytest.mark.parametrize("case", range(len(CROSS_THREAD)))
def test_backward_sized_with_default_knobs_runs_under_any(case):
    """include/shiftnd_hip.h: the bytes shiftnd_backward_workspace_bytes returns on a thread with untouched knobs serve a
    shiftnd_backward on a thread whose knobs plan more partial records (the call runs the default plan instead of failing with
    WORKSPACE_TOO_SMALL).  grad_x is the oracle's bit for bit (fp32; 1 ulp for bf16) and the default thread's, grad_w within
    the parity bar."""
    import threading
    import numpy as np
    from cases import rel_err, gw16_tol
    from oracle import oracle as O
    from test_hip_parity import _ulp_close
    from torchshifts import abi
    shape, tdt, cl, knobs = CROSS_THREAD[case]
    nd = len(shape) - 2
    abi.set_path_policy(0)
    rs = np.random.RandomState(case + 11)
    xt = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt)
    gt = torch.from_numpy(rs.uniform(-1, 1, size=shape)).to(tdt)
    wt = torch.from_numpy(rs.uniform(-2.5, 2.5, size=(shape[1], nd))).to(tdt)
    x, go, w = (t.to(torch.float64).numpy().astype(np.float32) for t in (xt, gt, wt))
    xd, god, wd = xt.to(DEV), gt.to(DEV), wt.to(DEV)
    if cl:
        xd, god = abi.to_channels_last(xd), abi.to_channels_last(god)
    for pad in (0, 3):
        ws = abi.backward_workspace(xd, pad, True)      # sized here, knobs untouched
        gx0, gw0 = abi.backward(god, wd, xd, pad, True, workspace=ws)
        k0 = abi.last_kernel()
        out = {}

        def run():
            try:
                for knob, value in knobs:
                    abi.set_tuning(knob, value)
                ws.fill_(0xA5)
                out["given"] = abi.backward(god, wd, xd, pad, True, workspace=ws)
                out["kernel"] = abi.last_kernel()
                out["own"] = abi.backward(god, wd, xd, pad, True)   # ... and a buffer sized on THIS thread serves its own plan
                out["bytes"] = abi.backward_workspace(xd, pad, True).numel()
            except Exception as e:   # noqa: BLE001 (reported by the assertion below)
                out["err"] = e
            finally:
                for knob, _ in knobs:
                    abi.set_tuning(knob, KNOB_DEFAULTS[knob])

        t = threading.Thread(target=run)
        t.start()
        t.join()
        assert "err" not in out, (case, pad, out.get("err"))
        assert out["bytes"] >= ws.numel(), (case, pad)
        gx_ref = torch.from_numpy(O.backward(go, w, x, pad, 1)[0]).to(tdt)
        _, gw64 = O.backward(go.astype(np.float64), w.astype(np.float64), x.astype(np.float64), pad, 1)
        tol = 1e-5 if tdt == torch.float32 else gw16_tol(torch.finfo(tdt).eps)
        for name, (gx, gw) in (("default", (gx0, gw0)), ("given", out["given"]), ("own", out["own"])):
            got = gx.cpu()   # (torch compares values, whatever the layout)
            if tdt == torch.float32:
                assert torch.equal(got, gx_ref), (case, pad, name, k0, out["kernel"])
            else:
                assert _ulp_close(got, gx_ref, tdt), (case, pad, name, k0, out["kernel"])
            assert rel_err(gw.to(torch.float64).cpu().numpy(), gw64) < tol, (case, pad, name, k0, out["kernel"])
