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


# -- every kernel family has a default-routed shape ----------------------------------------------------------------------------
# (entry point, shiftnd_last_kernel(), dtype, shape, cut, padding, active, channels-last): the smallest problem of each family in
# tools/route_census.py's seeded sample of 6000 problems (gpurun_out/consol/route_census.txt, round 5) -- all knobs untouched.  A
# family that loses its last default-routed shape fails here and should be deleted, not kept "on request" (VERDICT r04 item 7);
# the fused-pool families are pinned by tests/test_pooled_gpu.py.
FAMILY_ROUTES = [
    ("backward", "band_plane_backward", "float32", (2, 1, 1), None, 1, 0, False),
    ("backward", "cl_backward", "float32", (2, 2, 16, 14, 4), None, 0, 1, True),
    ("backward", "cl_tiled_backward", "bfloat16", (1, 64, 32, 31), None, 0, 0, True),
    ("backward", "cl_tiled_backward_3d", "bfloat16", (1, 16, 2, 112, 1), None, 3, 1, True),
    ("backward", "crop_backward", "float64", (1, 1, 7, 112), [[1, 0], [2, 2]], 0, 0, False),
    ("backward", "crop_backward_ragged", "float64", (2, 1, 100, 225), None, 2, 1, False),
    ("backward", "flat_backward", "bfloat16", (2, 2, 7, 11), None, 1, 0, False),
    ("backward", "plane_backward", "float32", (1, 1, 48), None, 2, 0, False),
    ("backward", "plane_backward_ragged", "bfloat16", (1, 2, 16, 28, 28), None, 3, 1, False),   # (added with the kernels, after the census)
    ("backward", "plane_backward_lds", "float64", (1, 2, 14, 3, 24), None, 2, 1, False),
    ("backward", "row_backward", "float64", (1, 16, 1000), None, 1, 0, False),
    ("backward", "slide_backward", "float16", (1, 1, 1, 2, 200), None, 0, 0, False),
    ("backward", "small_plane_backward", "float64", (2, 2, 1, 56, 7), None, 2, 1, False),
    ("backward", "step_backward", "float32", (3, 1, 1, 24), None, 1, 0, False),
    # (round 6: the census no longer reaches it -- 0 of 5 131 backward calls; what is left are index maps beyond LDS: here 13 000 planes of
    #  odd 16-bit rows -- and tensors that are neither contiguous nor channels-last)
    ("backward", "strided_backward", "float16", (1, 1, 13000, 2, 3), None, 1, 1, False),
    ("backward", "sweep_backward", "float16", (2, 3, 16384), None, 4, 1, False),
    ("backward", "walk_backward", "float32", (2, 2, 3, 3, 16), None, 2, 1, False),
    ("backward", "walk_backward16", "bfloat16", (1, 2, 32, 7, 200), None, 1, 1, False),
    ("backward", "walk_backward16_sparse", "float16", (3, 1, 5, 4, 112), None, 2, 0, False),
    ("backward", "walk_backward_sparse", "float32", (1, 3, 12, 16, 12), None, 0, 0, False),
    ("forward", "band_plane_forward", "float16", (1, 1, 1), None, 3, 1, False),
    ("forward", "cl_active_forward", "float32", (2, 2, 16, 14, 4), None, 0, 1, True),
    ("forward", "cl_gather_forward", "float64", (2, 3, 300, 7, 32), None, 4, 0, True),
    ("forward", "cl_tiled_active_forward", "float16", (3, 64, 7, 9), None, 0, 1, True),
    ("forward", "cl_tiled_active_forward_3d", "bfloat16", (1, 16, 2, 112, 1), None, 3, 1, True),
    ("forward", "cl_tiled_forward", "float32", (3, 256, 1, 9), None, 3, 0, True),
    ("forward", "cl_tiled_forward_3d", "bfloat16", (2, 256, 2, 28, 4), None, 1, 0, True),
    ("forward", "crop_active_forward", "float32", (1, 1, 14, 12), [[2, 0], [1, 2]], 0, 1, False),
    ("forward", "crop_gather_forward", "float64", (1, 1, 7, 112), [[1, 0], [2, 2]], 0, 0, False),
    ("forward", "flat_active_forward", "bfloat16", (3, 3, 112, 3), None, 1, 1, False),
    ("forward", "flat_gather_forward", "bfloat16", (2, 2, 7, 11), None, 1, 0, False),
    ("forward", "plane_active_forward", "float64", (1, 16, 96), None, 1, 1, False),
    ("forward", "plane_active_forward_ragged", "float32", (1, 2, 8, 30, 62), None, 1, 1, False),
    ("forward", "plane_gather_forward", "float32", (2, 1, 1), None, 1, 0, False),
    ("forward", "plane_gather_forward_lds", "float16", (2, 1, 384), None, 4, 0, False),
    ("forward", "ragged_active_forward", "float32", (1, 1, 16, 62), None, 0, 1, False),
    ("forward", "ragged_gather_forward", "float64", (1, 3, 32, 31), None, 4, 0, False),
    ("forward", "row_active_forward", "float16", (8, 1, 5000), None, 0, 1, False),
    ("forward", "row_gather_forward", "float64", (1, 16, 1000), None, 1, 0, False),
    ("forward", "slide_forward", "float16", (1, 2, 1, 112, 16), None, 1, 1, False),
    ("forward", "small_plane_forward", "float64", (2, 2, 1, 56, 7), None, 2, 1, False),
    ("forward", "step_active_forward", "float64", (1, 2, 14, 3, 24), None, 2, 1, False),
    ("forward", "step_gather_forward", "float32", (64, 2, 512, 28), None, 0, 0, False),
    ("forward", "step_gather_forward_lds", "bfloat16", (3, 2, 1, 16, 128), None, 3, 0, False),
    ("forward", "step_gather_forward_small", "float16", (3, 3, 100, 224), None, 0, 0, False),
    ("forward", "strided_active_forward", "float16", (1, 1, 70000), [[1, 2]], 2, 1, False),   # (round 6: 11 of 5 131 -- windows on rows whose maps exceed LDS)
    ("forward", "sweep_active_forward", "float32", (8, 1, 2, 8, 40000), None, 4, 1, False),
    ("forward", "sweep_gather_forward", "float64", (8, 16, 28, 5, 32), None, 0, 0, False),
    ("forward", "walk_forward", "float32", (2, 2, 3, 3, 16), None, 2, 1, False),
    ("forward", "walk_forward16", "bfloat16", (1, 2, 32, 7, 200), None, 1, 1, False),
    ("forward_quantized", "band_gather_forward", "uint8", (1, 1, 4096, 5), None, 4, 1, False),
    ("forward_quantized", "bytes_block_forward", "int8", (1, 256, 3), None, 2, 0, False),
    ("forward_quantized", "bytes_gather_forward", "int8", (2, 2, 112), None, 4, 0, False),
    ("forward_quantized", "cl_gather_forward", "int8", (1, 2, 13, 1), None, 1, 0, True),
    ("forward_quantized", "cl_tiled_forward", "int8", (2, 16, 15, 3), None, 1, 0, True),
    ("forward_quantized", "cl_tiled_forward_3d", "int32", (2, 64, 4, 8, 19), None, 4, 0, True),
    ("forward_quantized", "plane_gather_forward", "int32", (2, 3, 112, 12), None, 0, 1, False),
    ("forward_quantized", "rows_gather_forward", "uint8", (1, 2, 3, 64, 256), None, 3, 0, False),
    ("forward_quantized", "step_gather_forward", "int32", (2, 1, 100, 224), None, 3, 1, False),
    ("forward_quantized", "step_gather_forward_small", "int8", (2, 36, 224, 512), None, 0, 0, False),
    ("forward_quantized", "sweep_gather_forward", "int32", (2, 2, 100, 32, 16), [[1, 1], [1, 1], [0, 1]], 2, 0, False),
]


@pytest.mark.parametrize("entry", FAMILY_ROUTES, ids=lambda e: "%s-%s" % (e[0], e[1]))
def test_every_family_has_a_default_route(entry):
    from torchshifts import abi
    kind, kernel, dt, shape, cut, pad, active, cl = entry
    tdt = getattr(torch, dt)
    nd = len(shape) - 2
    abi.set_path_policy(0)
    b, new = abi.check_borders(list(shape), cut, nd) if cut else (None, list(shape))
    torch.manual_seed(1)
    if kind == "forward_quantized":
        if tdt == torch.int32:
            xq = torch.randint(-1000, 1000, shape, dtype=tdt, device=DEV)
        else:
            info = torch.iinfo(tdt)
            xq = torch.randint(info.min, info.max + 1, shape, dtype=tdt, device=DEV)
        wq = torch.randint(118, 139, (shape[1], nd), dtype=torch.uint8, device=DEV)
        out = None
        if cl:
            xq = abi.to_channels_last(xq)
            out = abi.to_channels_last(torch.empty(new, dtype=tdt, device=DEV))
        abi.forward_quantized(xq, wq, 128, 3, pad, b, out=out)
        assert abi.last_kernel() == kernel, (entry, abi.last_kernel())
        return
    x = torch.rand(shape, device=DEV).to(tdt)
    go = torch.rand(new, device=DEV).to(tdt)
    w = ((torch.rand(shape[1], nd, device=DEV) * 2 - 1) * 4).to(tdt)
    out = gx = None
    if cl:
        x, go = abi.to_channels_last(x), abi.to_channels_last(go)
        out, gx = torch.empty_like(go), torch.empty_like(x)
    if kind == "forward":
        abi.forward(x, w, pad, active, b, out=out)
    else:
        abi.backward(go, w, x, pad, active, b, grad_x=gx)
    assert abi.last_kernel() == kernel, (entry, abi.last_kernel())


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
