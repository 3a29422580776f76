"""GPU: the drop-in operator surface (torch.ops.torchshifts.* on HIP tensors, autograd, modules,
QuantizedCUDA) against the golden fixtures and against the same ops on CPU tensors."""
import numpy as np
import pytest
import torch

import torchshifts
from torchshifts import Shift1d, Shift2d, Shift3d
from cases import float_cases, quant_cases, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
OPS = torch.ops.torchshifts


def _op(nd):
    return getattr(OPS, "shift%dd" % nd)


def test_dispatcher_autograd_matches_golden():
    for key, nd, dt, pad, active, crop, x, w, go_full, out_r, gx_r, gw_r in float_cases("g1_float.npz"):
        xt = torch.from_numpy(x.copy()).to(DEV).requires_grad_(True)
        wt = torch.from_numpy(w.copy()).to(DEV).requires_grad_(True)
        b = torch.Tensor() if crop is None else torch.tensor(crop, dtype=torch.long)
        out = _op(nd)(xt, wt, b, pad, bool(active))
        assert out.is_cuda and out.is_contiguous()
        go = torch.from_numpy(np.ascontiguousarray(go_full[tuple(slice(0, s) for s in out.shape)])).to(DEV)
        out.backward(go)
        assert np.array_equal(out.detach().cpu().numpy(), out_r), "forward " + key
        assert np.array_equal(xt.grad.cpu().numpy(), gx_r), "grad_x " + key
        assert np.array_equal(wt.grad.cpu().numpy(), gw_r), "grad_w " + key


def test_quantized_cuda_key():
    tdt = {"quint8": torch.uint8, "qint8": torch.int8, "qint32": torch.int32}
    n = 0
    for key, nd, xname, layout, wname, pad, crop, xq, xzp, wq, wzp, out_r in quant_cases(nds=(2, 3)):
        if wname != "wu8":
            continue
        x = torch._make_per_tensor_quantized_tensor(torch.from_numpy(xq).to(tdt[xname]).to(DEV), 0.05, xzp)
        if layout == "cl":
            x = x.contiguous(memory_format=torch.channels_last)
        elif layout == "cl3d":
            x = x.contiguous(memory_format=torch.channels_last_3d)
        w = torch._make_per_tensor_quantized_tensor(torch.from_numpy(wq).to(torch.uint8).to(DEV), 1.0, wzp)
        b = torch.Tensor() if crop is None else torch.tensor(crop, dtype=torch.long)
        out = _op(nd)(x, w, b, pad, False)
        assert out.is_quantized and out.is_cuda and out.q_zero_point() == xzp
        assert np.array_equal(out.int_repr().cpu().contiguous().numpy(), out_r), key
        n += 1
    assert n == 120
    with pytest.raises(RuntimeError, match="backwards on quantized tensor are not supported"):
        OPS._shift2d_backward(x, w, x, torch.tensor([0, 1, 0, 1, 0, 1], dtype=torch.int32), 0, False)


def test_modules_gpu_vs_cpu():
    """the scenario of the reference's tests/shifts_test.py (fwd + MSE + bwd, SSL and active, emulate_dw),
    with real assertions: HIP == CPU backend"""
    torch.manual_seed(0)
    args = {'kernel_size': 3, 'stride': 1, 'padding': (0, 0)}
    for active in (False, True):
        m = Shift2d(16, init_shift=1, sparsity_term=0., active_flag=active, emulate_dw=dict(args), init_thumb_rule=2)
        x = torch.rand(8, 16, 64, 64)
        t = 10 * torch.rand(8, 16, 62, 62)
        xc = x.clone().requires_grad_(True)
        out_c, loss = m(xc)
        assert loss is None and out_c.shape == (8, 16, 62, 62)
        torch.nn.functional.mse_loss(out_c, t).backward()
        gw_c, gx_c = m.weight.grad.clone(), xc.grad.clone()
        m.zero_grad()
        mg = m.to(DEV)
        xg = x.to(DEV).requires_grad_(True)
        out_g, _ = mg(xg)
        torch.nn.functional.mse_loss(out_g, t.to(DEV)).backward()
        assert torch.equal(out_g.cpu(), out_c.detach())
        assert torch.allclose(xg.grad.cpu(), gx_c, rtol=1e-6, atol=1e-9)
        assert mg.weight.grad is not None and rel_err(mg.weight.grad.cpu().numpy(), gw_c.numpy()) < 1e-5
    # 1-D / 3-D modules, strided emulation (avg-pool tail), sparsity loss
    m3 = Shift3d(4, padding='reflect', active_flag=True, emulate_dw={'kernel_size': 3, 'stride': 2, 'padding': 1}).to(DEV)
    out, loss = m3(torch.rand(2, 4, 8, 8, 8, device=DEV))
    assert out.shape == (2, 4, 4, 4, 4) and loss is not None and loss.is_cuda
    m1 = Shift1d(3, padding='periodic').to(DEV)
    out, _ = m1(torch.rand(2, 3, 64, device=DEV))
    assert out.shape == (2, 3, 64)


def test_quantized_module_gpu():
    from torchshifts.quantized.modules import Shift2d as QShift2d
    torch.manual_seed(1)
    m = Shift2d(8, init_shift=2, sparsity_term=0.)
    q = QShift2d.from_float(m)
    x = torch.rand(4, 8, 24, 24)
    xq = torch.quantize_per_tensor(x, 1 / 255., 0, torch.quint8)
    ref = q(xq)  # QuantizedCPU key
    q = q.to(DEV)  # the quantized weights are a buffer: they move with the module
    assert q.qweight.is_cuda and q.qweight_repr.is_cuda
    out = q(xq.to(DEV))
    assert out.is_cuda and torch.equal(out.int_repr().cpu(), ref.int_repr())


@pytest.mark.parametrize("name,shape", [("Shift1d", (3, 4, 37)), ("Shift2d", (3, 8, 25, 24)), ("Shift3d", (2, 4, 7, 8, 16))])
def test_quantized_module_with_pool_tail_gpu(name, shape):
    """quantized module emulating a strided depthwise conv: shift + avg_pool(stride, ceil_mode) on the QuantizedCUDA
    key equals the QuantizedCPU result (ATen's quantized average pool) bit for bit, ragged last windows included"""
    import torchshifts.quantized.modules as QM
    from torchshifts import abi
    torch.manual_seed(5)
    for stride, zp, qdt, padding in ((2, 0, torch.quint8, 'reflect'), (3, 17, torch.quint8, 'zeros'), (2, -5, torch.qint8, 'periodic'),
                                     (3, 3, torch.qint8, 'border'), (2, 100, torch.quint8, 'symmetric')):
        m = getattr(torchshifts, name)(shape[1], padding=padding, sparsity_term=0.,
                                       emulate_dw={'kernel_size': 3, 'stride': stride, 'padding': 1})
        q = getattr(QM, name).from_float(m)
        x = torch.rand(shape)
        xq = torch.quantize_per_tensor(x, 1 / 255. if qdt == torch.quint8 else 1 / 127., zp, qdt)
        ref = q(xq)
        out = q.to(DEV)(xq.to(DEV))
        assert abi.last_kernel() in ("qpool_forward", "qpool_plane_forward", "qpool_band_forward", "qpool_band_fast")  # shift and pool in ONE pass (round 2: six passes of torch ops)
        assert out.is_cuda and out.is_quantized and out.shape == ref.shape
        assert out.q_zero_point() == ref.q_zero_point() and abs(out.q_scale() - ref.q_scale()) < 1e-12
        assert torch.equal(out.int_repr().cpu(), ref.int_repr()), (name, stride, padding)
    # qint32 tensors and non-contiguous inputs take the two-step route on the same key: same values
    m = getattr(torchshifts, name)(shape[1], padding='reflect', sparsity_term=0., emulate_dw={'kernel_size': 3, 'stride': 2, 'padding': 1})
    q = getattr(QM, name).from_float(m)
    xq = torch.quantize_per_tensor(torch.rand(shape), 1 / 255., 3, torch.quint8)
    ref = q(xq)
    xs = torch.quantize_per_tensor(torch.rand((shape[0] * 2,) + tuple(shape[1:])), 1 / 255., 3, torch.quint8)
    ref_s = q(xs[::2])
    out_s = q.to(DEV)(xs.to(DEV)[::2])
    assert torch.equal(out_s.int_repr().cpu(), ref_s.int_repr())
    # ATen's QuantizedCPU pool rounds with the zero point outside the rounding for channels-last tensors, for every 3-D
    # tensor and for one-channel tensors (its channels-last kernel), inside otherwise: odd zero points tell the two apart
    q = q.cpu()
    if len(shape) > 3:
        fmt = torch.channels_last if len(shape) == 4 else torch.channels_last_3d
        for zp in (3, 101):
            xc = torch.quantize_per_tensor(torch.rand(shape), 1 / 255., zp, torch.quint8).contiguous(memory_format=fmt)
            ref_c = q.cpu()(xc)
            xg = xc.to(DEV).contiguous(memory_format=fmt)  # (a quantized .to(device) does not promise to keep the layout)
            assert xg.is_contiguous(memory_format=fmt) and not xg.is_contiguous()
            out_c = q.to(DEV)(xg)
            assert torch.equal(out_c.int_repr().cpu(), ref_c.int_repr()), (name, "channels-last", zp)
    m1 = getattr(torchshifts, name)(1, padding='zeros', sparsity_term=0., emulate_dw={'kernel_size': 3, 'stride': 3, 'padding': 1})
    q1 = getattr(QM, name).from_float(m1)
    for zp in (3, 101):
        x1 = torch.quantize_per_tensor(torch.rand((shape[0], 1) + tuple(shape[2:])), 1 / 255., zp, torch.quint8)
        ref_1 = q1.cpu()(x1)
        out_1 = q1.to(DEV)(x1.to(DEV))
        assert abi.last_kernel() in ("qpool_forward", "qpool_plane_forward", "qpool_band_forward", "qpool_band_fast")
        assert torch.equal(out_1.int_repr().cpu(), ref_1.int_repr()), (name, "one channel", zp)


def test_error_behaviour_gpu():
    x = torch.rand(2, 3, 8, 8, device=DEV)
    w = torch.rand(3, 2, device=DEV)
    with pytest.raises(RuntimeError, match="same type"):
        OPS.shift2d(x, w.double(), torch.Tensor(), 0, False)
    with pytest.raises(RuntimeError, match="weights must be a CUDA tensor"):
        OPS._shift2d_forward(x, w.cpu(), torch.tensor([0, 8, 0, 8, 0, 1], dtype=torch.int32), [2, 3, 8, 8], 0, False)
    with pytest.raises(AssertionError):
        torchshifts.functional.shift2d_func(x, w.cpu(), 0, False)
    with pytest.raises(RuntimeError, match="not implemented for 'Int'"):
        OPS.shift2d(x.int(), w.int(), torch.Tensor(), 0, False)
    # borders may live on the device (the reference passes them there); result unchanged
    b = torch.tensor([1, 7, 0, 8, 0, 1], dtype=torch.int32)
    a = OPS._shift2d_forward(x, w, b, [2, 3, 6, 8], 2, True)
    c = OPS._shift2d_forward(x, w, b.to(DEV), [2, 3, 6, 8], 2, True)
    assert torch.equal(a, c)


def test_streams_and_graph_capture():
    """kernels launch on the current stream and are capture-safe (no allocation / sync in the C ABI)"""
    from torchshifts import abi
    x = torch.rand(4, 8, 32, 32, device=DEV)
    w = (torch.rand(8, 2, device=DEV) - 0.5) * 4
    ref = abi.forward(x, w, 3, 1)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        out = abi.forward(x, w, 3, 1)
    s.synchronize()
    assert torch.equal(out, ref)
    out2 = torch.empty_like(ref)
    gx, gw = torch.empty_like(x), torch.empty_like(w)
    ws = abi.backward_workspace(x, 3, 1)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        abi.forward(x, w, 3, 1, out=out2)
        abi.backward(ref, w, x, 3, 1, grad_x=gx, grad_w=gw, workspace=ws)
    g.replay()
    torch.cuda.synchronize()
    gx_ref, gw_ref = abi.backward(ref, w, x, 3, 1)
    assert torch.equal(out2, ref) and torch.equal(gx, gx_ref) and torch.equal(gw, gw_ref)


def test_device_borders_stay_sync_free_and_capturable():
    """The reference hands its private ops a 6-int DEVICE tensor (cuda/shifts_cuda.cu:61-67, ops/shifts.cpp:134).  For a window of
    the input's own sizes the borders are implied, so such a caller costs no D2H read: the dispatcher ops capture into a HIP graph
    with device-resident borders (round-5 verdict item 9); a cropped window with device borders still gives the host result."""
    x = torch.rand(4, 8, 32, 32, device=DEV)
    g = torch.rand(4, 8, 32, 32, device=DEV)
    w = (torch.rand(8, 2, device=DEV) - 0.5) * 4
    bh = torch.tensor([0, 32, 0, 32, 0, 1], dtype=torch.int32)
    bd = bh.to(DEV)
    ref = OPS._shift2d_forward(x, w, bh, [4, 8, 32, 32], 3, True)
    gx_ref, gw_ref = OPS._shift2d_backward(g, w, x, bh, 3, True)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):   # warm the allocator's pool of the capture stream
            OPS._shift2d_forward(x, w, bd, [4, 8, 32, 32], 3, True)
            OPS._shift2d_backward(g, w, x, bd, 3, True)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):   # a D2H copy of the borders (or any sync) would fail the capture
        out = OPS._shift2d_forward(x, w, bd, [4, 8, 32, 32], 3, True)
        gx, gw = OPS._shift2d_backward(g, w, x, bd, 3, True)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref) and torch.equal(gx, gx_ref) and torch.equal(gw, gw_ref)
    # quantized forward, same rule
    xq = torch._make_per_tensor_quantized_tensor((torch.rand(2, 8, 16, 16, device=DEV) * 255).to(torch.uint8), 0.1, 3)
    wq = torch.quantize_per_tensor((torch.rand(8, 2) - 0.5) * 4, 1.0, 128, torch.quint8)
    bq = torch.tensor([0, 16, 0, 16, 0, 1], dtype=torch.int32)
    a = OPS._shift2d_forward(xq, wq, bq, [2, 8, 16, 16], 1, False)
    c = OPS._shift2d_forward(xq, wq, bq.to(DEV), [2, 8, 16, 16], 1, False)
    assert torch.equal(a.int_repr(), c.int_repr())
    # a cropped window: device borders are read once (sync), the values are the host-border ones
    bc = torch.tensor([1, 31, 2, 32, 0, 1], dtype=torch.int32)
    a = OPS._shift2d_forward(x, w, bc, [4, 8, 30, 30], 2, False)
    c = OPS._shift2d_forward(x, w, bc.to(DEV), [4, 8, 30, 30], 2, False)
    assert torch.equal(a, c)


def test_pool_op_gpu_vs_cpu():
    """torchshifts::shift{N}d_pool on HIP tensors (fused kernels) == the same op on CPU tensors (the reference's
    shift + avg_pool sequence): fp32 values bit-exact, grad_x bit-exact, grad_w <= 1e-5 relative"""
    import torchshifts.functional as TF
    from torchshifts import abi
    torch.manual_seed(5)
    for nd, shape, pool, crop in [(1, (2, 6, 40), 2, None), (2, (2, 8, 21, 32), (2, 2), None),
                                  (2, (2, 4, 18, 24), (3, 2), [[1, 0], [0, 2]]), (3, (1, 4, 6, 7, 8), 2, None)]:
        fn = getattr(TF, "shift%dd_pool_func" % nd)
        for pad, active in ((0, False), (1, True), (4, False)):
            x = torch.rand(shape)
            w = torch.rand(shape[1], nd) * 5 - 2.5
            b = None if crop is None else torch.tensor(crop)
            xc, wc = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            out_c = fn(xc, wc, pad, active, b, pool)
            g = torch.rand_like(out_c)
            out_c.backward(g)
            xg, wg = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
            out_g = fn(xg, wg, pad, active, b, pool)
            # (3-D interpolating: the pool rides on the walk through the planes)
            fstep = nd == 2 and pool == (2, 2)   # (2-D, 2 x 2 windows: the one-step sweep + pool; round 6: both shifts, any window)
            assert abi.last_kernel() == ("walk_forward_pool" if (nd == 3 and active) else
                                         ("step_gather_forward_pool" if fstep else ("crop_forward3_pool" if nd == 3 else "plane_pool_forward")))
            out_g.backward(g.to(DEV))  # (runs on the autograd thread: its kernel name is checked in test_pooled_gpu.py)
            assert torch.equal(out_g.cpu(), out_c.detach()), (shape, pad, active)
            assert torch.equal(xg.grad.cpu(), xc.grad), (shape, pad, active)
            assert rel_err(wg.grad.cpu().numpy(), wc.grad.numpy()) < 1e-5
    # a layout the fused kernels do not serve (channels-last input) falls back to the two-step sequence, same values
    x = torch.rand(2, 8, 12, 16).to(DEV).contiguous(memory_format=torch.channels_last)
    w = (torch.rand(8, 2) * 4 - 2).to(DEV)
    a = TF.shift2d_pool_func(x, w, 0, False, None, 2)
    bb = TF.shift2d_pool_func(x.contiguous(), w, 0, False, None, 2)
    assert torch.equal(a, bb)


def test_strided_module_uses_fused_pool():
    """a module that emulates a stride-2 depthwise conv runs ONE fused kernel per direction on the GPU and matches
    the CPU module (= the reference's sequence)"""
    from torchshifts import abi
    torch.manual_seed(2)
    m = Shift2d(16, padding='border', sparsity_term=0., emulate_dw={'kernel_size': 3, 'stride': 2, 'padding': 1})
    x = torch.rand(4, 16, 30, 30)
    xc = x.clone().requires_grad_(True)
    out_c, _ = m(xc)
    assert out_c.shape == (4, 16, 15, 15)
    out_c.square().sum().backward()
    gw_c, gx_c = m.weight.grad.clone(), xc.grad.clone()
    m.zero_grad()
    mg = m.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    out_g, _ = mg(xg)
    assert abi.last_kernel() == "step_gather_forward_pool"   # (round 6: windows of any width; 30 fp32 columns are not whole pieces)
    out_g.square().sum().backward()
    assert torch.equal(out_g.cpu(), out_c.detach())
    assert torch.allclose(xg.grad.cpu(), gx_c, rtol=1e-6, atol=1e-9)
    assert rel_err(mg.weight.grad.cpu().numpy(), gw_c.numpy()) < 1e-5


def test_strided_3d_modules_use_the_walk_pool_kernels():
    """Shift3d emulating a stride-2 depthwise conv, sparse (the reference's default) and interpolating: one fused kernel per
    direction on the GPU (the walk through the planes with the pool riding on it) and the CPU module's values (= the reference's
    shift + avg_pool3d sequence): outputs and grad_x bit-exact in fp32, grad_w within 1e-5"""
    from torchshifts import abi
    torch.manual_seed(4)
    for active in (False, True):
        m = Shift3d(8, padding='reflect', sparsity_term=0., active_flag=active, emulate_dw={'kernel_size': 3, 'stride': 2, 'padding': 1})
        with torch.no_grad():
            m.weight.copy_(torch.rand(8, 3) * 5 - 2.5)
        x = torch.rand(2, 8, 6, 11, 16)
        xc = x.clone().requires_grad_(True)
        out_c, _ = m(xc)
        assert out_c.shape == (2, 8, 3, 6, 8)
        out_c.square().sum().backward()
        gw_c, gx_c = m.weight.grad.clone(), xc.grad.clone()
        m.zero_grad()
        mg = m.to(DEV)
        xg = x.to(DEV).requires_grad_(True)
        out_g, _ = mg(xg)
        assert abi.last_kernel() == ("walk_forward_pool" if active else "crop_forward3_pool"), abi.last_kernel()   # (sparse: pooled rows through LDS, round 6)
        out_g.square().sum().backward()
        assert torch.equal(out_g.cpu(), out_c.detach()), active
        assert torch.equal(xg.grad.cpu(), gx_c), active
        assert rel_err(mg.weight.grad.cpu().numpy(), gw_c.numpy()) < 1e-5, active
        m.cpu()


def test_layout_change_kernel():
    """shiftnd_transpose (channels-last <-> contiguous) is a pure permutation for every element size and ragged shape"""
    from torchshifts import abi
    torch.manual_seed(0)
    for shape, dt in [((3, 70, 13, 9), torch.float32), ((2, 64, 16, 16), torch.float16), ((2, 5, 7, 3), torch.int8),
                      ((1, 130, 65, 67), torch.uint8), ((2, 8, 3, 4, 5), torch.float64), ((4, 256, 56, 56), torch.bfloat16),
                      ((2, 3, 100), torch.float32)]:
        x = (torch.rand(shape, device=DEV) * 200 - 100).to(dt)
        if len(shape) == 3:  # [N, C, L]: "channels-last" = [N, L, C] storage
            xc = x.permute(0, 2, 1).contiguous().permute(0, 2, 1)
            assert torch.equal(abi.to_contiguous(xc), x)
            continue
        fmt = torch.channels_last if len(shape) == 4 else torch.channels_last_3d
        xc = x.contiguous(memory_format=fmt)
        a, b = abi.to_contiguous(xc), abi.to_channels_last(x)
        assert a.is_contiguous() and torch.equal(a, x), shape
        assert b.stride() == xc.stride() and torch.equal(b, x), shape


def test_channels_last_through_the_ops():
    """channels-last tensors through the dispatcher ops: float ops return NCHW-contiguous results (like the reference),
    values and gradients equal the contiguous call; the fused-pool op and the modules accept them too"""
    import torchshifts.functional as TF
    torch.manual_seed(1)
    x = torch.rand(4, 24, 20, 28, device=DEV)
    w = (torch.rand(24, 2, device=DEV) * 6 - 3)
    for pad, active in ((0, False), (3, True)):
        xa = x.clone().requires_grad_(True)
        wa = w.clone().requires_grad_(True)
        ref = TF.shift2d_func(xa, wa, pad, active)
        g = torch.rand_like(ref)
        ref.backward(g)
        xb = x.clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        wb = w.clone().requires_grad_(True)
        out = TF.shift2d_func(xb, wb, pad, active)
        assert out.is_contiguous() and torch.equal(out, ref)
        out.backward(g.contiguous(memory_format=torch.channels_last))
        assert torch.equal(xb.grad, xa.grad) and rel_err(wb.grad.cpu().numpy(), wa.grad.cpu().numpy()) < 1e-6
        pa = TF.shift2d_pool_func(x, w, pad, active, None, 2)
        pb = TF.shift2d_pool_func(x.contiguous(memory_format=torch.channels_last), w, pad, active, None, 2)
        assert torch.equal(pa, pb)
    x3 = torch.rand(2, 6, 5, 8, 12, device=DEV)
    w3 = (torch.rand(6, 3, device=DEV) * 4 - 2)
    assert torch.equal(TF.shift3d_func(x3.contiguous(memory_format=torch.channels_last_3d), w3, 2, True),
                       TF.shift3d_func(x3, w3, 2, True))


def test_ndhwc_input_through_autograd_keeps_the_contiguous_copy():
    """round 6: a dense NDHWC (channels_last_3d) input through the public op -- the autograd node changes the layout ONCE and keeps the
    contiguous copy, so the backward runs the contiguous kernel on it (no second transpose of x).  Values: the CPU op's (= the
    reference's) on the same tensors, forward and grad_x bit-exact in fp32, grad_w within 1e-5."""
    from torchshifts import abi
    torch.manual_seed(7)
    x = torch.rand(2, 8, 5, 12, 16)
    w = torch.rand(8, 3) * 5 - 2.5
    for pad, active in ((0, False), (3, True), (2, True)):
        xc, wc = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        out_c = OPS.shift3d(xc, wc, torch.Tensor(), pad, active)
        g = torch.rand_like(out_c)
        out_c.backward(g)
        xg = x.to(DEV).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
        wg = w.to(DEV).requires_grad_(True)
        out_g = OPS.shift3d(xg, wg, torch.Tensor(), pad, active)
        assert not abi.last_kernel().startswith("cl_"), abi.last_kernel()   # the contiguous forward on the copy
        out_g.backward(g.to(DEV))
        assert torch.equal(out_g.cpu(), out_c.detach()), (pad, active)
        assert torch.equal(xg.grad.cpu(), xc.grad), (pad, active)
        assert rel_err(wg.grad.cpu().numpy(), wc.grad.numpy()) < 1e-5
        # the private op without autograd still takes the NDHWC tensor as it lies (direct kernel or its own layout change)
        with torch.no_grad():
            b = torch.tensor([0, 5, 0, 12, 0, 16], dtype=torch.int32)
            o2 = OPS._shift3d_forward(xg.detach(), wg.detach(), b, [2, 8, 5, 12, 16], pad, active)
            assert torch.equal(o2, out_g.detach())
