"""torch.ao.quantization.convert(..., mapping=torchshifts.quant_mapping) swaps float shift
modules for the quantized ones (the reference's README:87-92 flow; its own mapping table no longer imports on
torch 2.x).  CPU only: exercises the QuantizedCPU key."""
import torch
from torch import nn

import torchshifts
from torchshifts import Shift2d, quant_mapping


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.quant = torch.ao.quantization.QuantStub()
        self.shift = Shift2d(4, padding='border', sparsity_term=0.)
        self.dequant = torch.ao.quantization.DeQuantStub()

    def forward(self, x):
        x = self.quant(x)
        out = self.shift(x)
        out = out[0] if isinstance(out, tuple) else out  # float module returns (out, loss), quantized the tensor
        return self.dequant(out)


def test_convert_with_quant_mapping():
    torch.manual_seed(0)
    net = Net().eval()
    net.shift.weight.data = torch.tensor([[1.2, -0.7], [0.4, 2.6], [-1.5, 0.5], [3.0, -2.2]])
    x = torch.rand(2, 4, 12, 12)
    ref = net(x)
    net.qconfig = torch.ao.quantization.default_qconfig
    torch.ao.quantization.prepare(net, inplace=True)
    net(x)  # calibrate the stubs
    torch.ao.quantization.convert(net, inplace=True, mapping=quant_mapping)
    assert type(net.shift) is torchshifts.quantized.modules.Shift2d
    out = net(x)
    # a pure gather: the quantized result is the float result up to the input's quantisation step
    assert out.shape == ref.shape and (out - ref).abs().max() < 2.0 / 127


def test_quantized_module_state_dict_keeps_weight():
    q = torchshifts.quantized.modules.Shift2d.from_float(Shift2d(3, sparsity_term=0.))
    assert "weight" in q.state_dict()  # the float parameter still checkpoints (qweight is re-derived by from_float)
