"""cl_tiled_forward timing: N16 C256 224x224 fp32 channels-last input, NCHW and channels-last output, band-rows sweep.
   SHIFTND_HIP_LIB=variants/<x>.so python3 tools/cl_tiled_bench.py [band_rows ...]"""
import sys
import torch
sys.path.insert(0, "activesparseshifts-pytorch_amd"); sys.path.insert(0, ".")
from torchshifts import abi


def ev(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(it): fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / it)
    return best


dev = "cuda:0"
torch.manual_seed(0)
x = torch.rand(16, 256, 224, 224, device=dev); w = torch.rand(256, 2, device=dev) * 6 - 3
xc = x.contiguous(memory_format=torch.channels_last)
o = torch.empty_like(x); ocl = torch.empty_like(xc); ref = torch.empty_like(x)
abi.forward(x, w, 0, 0, out=ref)
for br in [int(a) for a in sys.argv[1:]] or [0]:
    abi.set_tuning(21, br)
    t1 = ev(lambda: abi.forward(xc, w, 0, 0, out=o)); k1 = abi.last_kernel()
    t2 = ev(lambda: abi.forward(xc, w, 0, 0, out=ocl)); k2 = abi.last_kernel()
    assert torch.equal(o, ref) and torch.equal(ocl, ref)
    print("band_rows %3d  CL->NCHW %.3f ms (%s, %.0f GB/s)   CL->CL %.3f ms (%s)" % (br, t1, k1, 8 * x.numel() / t1 / 1e6, t2, k2))
t0 = ev(lambda: abi.forward(x, w, 0, 0, out=o))
print("NCHW->NCHW %.3f ms (%s)" % (t0, abi.last_kernel()))
# quantized C4 in channels-last (the format is kept) and fp16
xq = torch.randint(0, 255, (128, 512, 56, 56), dtype=torch.uint8, device=dev)
wq = (torch.rand(512, 2, device=dev) * 6 - 3).round().add(128).to(torch.uint8)
xqc = xq.contiguous(memory_format=torch.channels_last)
oq = torch.empty_like(xq); oqc = torch.empty_like(xqc)
t1 = ev(lambda: abi.forward_quantized(xq, wq, 128, 0, 0, out=oq)); k1 = abi.last_kernel()
for br in [int(a) for a in sys.argv[1:]] or [0]:
    abi.set_tuning(21, br)
    t2 = ev(lambda: abi.forward_quantized(xqc, wq, 128, 0, 0, out=oqc)); k2 = abi.last_kernel()
    assert torch.equal(oq, oqc)
    print("C4 quint8 band_rows %3d  NCHW %.3f ms (%s)   CL->CL %.3f ms (%s)  %.0f GB/s" % (br, t1, k1, t2, k2, 2 * xq.numel() / t2 / 1e6))
abi.set_tuning(21, 0)
xh = torch.rand(16, 256, 224, 224, device=dev).half(); wh = (torch.rand(256, 2, device=dev) * 6 - 3).half()
xhc = xh.contiguous(memory_format=torch.channels_last)
oh = torch.empty_like(xh); ohc = torch.empty_like(xhc); oh2 = torch.empty_like(xh)
t1 = ev(lambda: abi.forward(xh, wh, 0, 0, out=oh)); k1 = abi.last_kernel()
t2 = ev(lambda: abi.forward(xhc, wh, 0, 0, out=oh2)); k2 = abi.last_kernel()
t3 = ev(lambda: abi.forward(xhc, wh, 0, 0, out=ohc)); k3 = abi.last_kernel()
assert torch.equal(oh, oh2) and torch.equal(oh, ohc)
print("fp16 N16 C256 224^2  NCHW %.3f ms (%s)   CL->NCHW %.3f ms (%s)   CL->CL %.3f ms (%s)" % (t1, k1, t2, k2, t3, k3))
# bf16 active forward / backward, all tensors channels-last
xb = torch.rand(16, 256, 224, 224, device=dev).bfloat16(); wb = (torch.rand(256, 2, device=dev) * 6 - 3).bfloat16()
gb = torch.rand(16, 256, 224, 224, device=dev).bfloat16()
xbc = xb.contiguous(memory_format=torch.channels_last); gbc = gb.contiguous(memory_format=torch.channels_last)
ob = torch.empty_like(xb); obc = torch.empty_like(xbc); gxb = torch.empty_like(xb); gxbc = torch.empty_like(xbc); gwb = torch.empty_like(wb)
ws = abi.backward_workspace(xb, 0, 1)
for active in (0, 1):
    t1 = ev(lambda: abi.forward(xb, wb, 0, active, out=ob)); k1 = abi.last_kernel()
    t2 = ev(lambda: abi.forward(xbc, wb, 0, active, out=obc)); k2 = abi.last_kernel()
    t3 = ev(lambda: abi.backward(gb, wb, xb, 0, active, grad_x=gxb, grad_w=gwb, workspace=ws)); k3 = abi.last_kernel()
    t4 = ev(lambda: abi.backward(gbc, wb, xbc, 0, active, grad_x=gxbc, grad_w=gwb, workspace=ws)); k4 = abi.last_kernel()
    assert torch.equal(ob, obc) and torch.equal(gxb, gxbc)
    print("bf16 active=%d  fwd NCHW %.3f ms (%s)  NHWC %.3f ms (%s)   bwd NCHW %.3f ms (%s)  NHWC %.3f ms (%s)" % (active, t1, k1, t2, k2, t3, k3, t4, k4))
# fp32 backward: all channels-last, the mixed form (NHWC saved input, NCHW gradient, NHWC grad_x), the contiguous kernels,
# and what the mixed form cost before (transpose x, contiguous backward: grad_x NCHW)
g = torch.rand(16, 256, 224, 224, device=dev); gc = g.contiguous(memory_format=torch.channels_last)
gx = torch.empty_like(x); gxc = torch.empty_like(xc); gw = torch.empty_like(w)
ws = abi.backward_workspace(x, 0, 1)
for active in (0, 1):
    t1 = ev(lambda: abi.backward(g, w, x, 0, active, grad_x=gx, grad_w=gw, workspace=ws)); k1 = abi.last_kernel()
    t2 = ev(lambda: abi.backward(gc, w, xc, 0, active, grad_x=gxc, grad_w=gw, workspace=ws)); k2 = abi.last_kernel()
    ref_gx = gxc.clone()
    t3 = ev(lambda: abi.backward(g, w, xc, 0, active, grad_x=gxc, grad_w=gw, workspace=ws)); k3 = abi.last_kernel()
    assert torch.equal(gxc, ref_gx) and torch.equal(gx, ref_gx)
    t4 = ev(lambda: abi.backward(g, w, abi.to_contiguous(xc), 0, active, grad_x=gx, grad_w=gw, workspace=ws))
    print("fp32 active=%d bwd  NCHW %.3f ms (%s)   all NHWC %.3f ms (%s)   NHWC x + NCHW grad %.3f ms (%s, %.0f GB/s)   transpose + NCHW %.3f ms"
          % (active, t1, k1, t2, k2, t3, k3, 12 * x.numel() / t3 / 1e6, t4))
# periodic padding through the tiled kernels (edge pixels / rows by the element pass) against the channel-fastest kernels
for active in (0, 1):
    t1 = ev(lambda: abi.forward(xc, w, 2, active, out=ocl)); k1 = abi.last_kernel()
    t2 = ev(lambda: abi.backward(gc, w, xc, 2, active, grad_x=gxc, grad_w=gw, workspace=ws)); k2 = abi.last_kernel()
    abi.set_tuning(20, 0)
    t3 = ev(lambda: abi.forward(xc, w, 2, active, out=ocl)); k3 = abi.last_kernel()
    t4 = ev(lambda: abi.backward(gc, w, xc, 2, active, grad_x=gxc, grad_w=gw, workspace=ws)); k4 = abi.last_kernel()
    abi.set_tuning(20, 1)
    print("fp32 periodic active=%d  fwd %.3f ms (%s) vs %.3f ms (%s)   bwd %.3f ms (%s) vs %.3f ms (%s)" % (active, t1, k1, t3, k3, t2, k2, t4, k4))
# backward: rows per band (knob 21) -- fewer, longer bands re-read fewer halo rows
for br in (0, 28, 56, 112, 224):
    abi.set_tuning(21, br)
    t2 = ev(lambda: abi.backward(gc, w, xc, 0, 0, grad_x=gxc, grad_w=gw, workspace=ws))
    t3 = ev(lambda: abi.backward(g, w, xc, 0, 0, grad_x=gxc, grad_w=gw, workspace=ws))
    t4 = ev(lambda: abi.backward(gc, w, xc, 0, 1, grad_x=gxc, grad_w=gw, workspace=ws))
    print("bwd band_rows %3d  all NHWC %.3f ms   NCHW grad %.3f ms   active %.3f ms" % (br, t2, t3, t4))
abi.set_tuning(21, 0)
