"""ResNet backbone with a 3-conv deep stem, as used by DeepLab-v2 in the
reference (`spml/models/backbones/resnet.py`).  Module / parameter names match
the reference so that its checkpoints load unchanged.  The stride-1 units of res4 / res5
(83 % of the step's flops) run on this repo's matrix-core convolutions in training mode
(`spml_amd/mc_bottleneck.py`); everything else goes through PyTorch-ROCm (MIOpen)."""
import math
import os

import torch
import torch.nn as nn
from torch.nn.modules.batchnorm import _BatchNorm

from spml_amd import mc_bottleneck
from spml_amd.nn.batchnorm import BatchNorm2d
from spml_amd.ops import batch_norm_act

BN_MOMENTUM = 3e-4


def _bn(ch):
  return BatchNorm2d(ch, momentum=BN_MOMENTUM)


class Bottleneck(nn.Module):
  """1x1 -> 3x3 (stride / dilation) -> 1x1(x4) residual unit (resnet.py:11-63)."""
  expansion = 4

  def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None):
    super().__init__()
    self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
    self.bn1 = _bn(planes)
    self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation,
                           dilation=dilation, bias=False)
    self.bn2 = _bn(planes)
    self.conv3 = nn.Conv2d(planes, planes * self.expansion, 1, bias=False)
    self.bn3 = _bn(planes * self.expansion)
    self.relu = nn.ReLU(inplace=True)
    self.downsample = downsample
    self.dilation = dilation
    self.stride = stride

  def forward(self, x):
    # stride-1 units with 256-multiple channel counts (all of res4 / res5) in training mode on
    # channels-last GPU activations: convolutions on the matrix cores at fp32-class accuracy,
    # batch norms fused around them (spml_amd/mc_bottleneck.py, csrc/conv.hip)
    if mc_bottleneck.available(self, x):
      return mc_bottleneck.bottleneck_forward(self, x)
    if mc_bottleneck.eval_available(self, x):           # inference: batch norm folded into the convolutions
      return mc_bottleneck.bottleneck_forward_eval(self, x)
    # batch_norm_act = relu(bn(.) [+ identity]): one fused HIP pass pair per batch norm on
    # channels-last GPU activations in training mode, the framework ops otherwise
    if self.downsample is None:
      identity = x
    else:
      identity = batch_norm_act(self.downsample[0](x), self.downsample[1], relu=False)
    y = batch_norm_act(self.conv1(x), self.bn1)
    y = batch_norm_act(self.conv2(y), self.bn2)
    return batch_norm_act(self.conv3(y), self.bn3, residual=identity)


class conv1(nn.Module):
  """Deep stem 3->64->64->128 (stride 2) + BN + ReLU + 3x3/2 max-pool (resnet.py:66-110)."""

  def __init__(self):
    super().__init__()
    self.inplanes = 128
    self.conv1 = nn.Sequential(
        nn.Conv2d(3, 64, 3, stride=2, padding=1, bias=False), _bn(64), nn.ReLU(inplace=True),
        nn.Conv2d(64, 64, 3, stride=1, padding=1, bias=False), _bn(64), nn.ReLU(inplace=True),
        nn.Conv2d(64, 128, 3, stride=1, padding=1, bias=False))
    self.bn1 = _bn(128)
    self.relu = nn.ReLU(inplace=True)
    self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)

  def forward(self, x):
    stem = self.conv1                   # conv, bn, relu, conv, bn, relu, conv
    y = batch_norm_act(stem[0](x), stem[1])
    y = batch_norm_act(stem[3](y), stem[4])
    y = batch_norm_act(stem[6](y), self.bn1)
    if (y.is_cuda and y.dtype == torch.float32 and not y.requires_grad and y.dim() == 4 and y.shape[1] % 4 == 0 and
        y.is_contiguous(memory_format=torch.channels_last) and not y.is_contiguous() and
        os.environ.get('SPML_NO_HIP_MAXPOOL') != '1'):
      # the frozen stem on a channels-last map: own kernel (the framework's NHWC max-pool runs at 1.6 TB/s: 0.43 ms)
      from spml_amd import _ffi
      return _ffi.maxpool3x3s2_nhwc(y)
    return self.maxpool(y)


class ResnetBackbone(nn.Module):
  """conv1 + res2..res5 (resnet.py:113-178); returns the four stage outputs."""

  def __init__(self, blocks, strides, dilations, config=None):
    super().__init__()
    self.inplanes = 128
    self.conv1 = conv1()
    for name, planes, idx in (('res2', 64, 0), ('res3', 128, 1), ('res4', 256, 2), ('res5', 512, 3)):
      setattr(self, name, self._make_layer(planes, blocks[idx], strides[idx], dilations[idx]))
    for m in self.modules():
      if isinstance(m, nn.Conv2d):
        fan = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
        m.weight.data.normal_(0, math.sqrt(2.0 / fan))
      elif isinstance(m, _BatchNorm):
        m.weight.data.fill_(1)
        if m.bias is not None:
          m.bias.data.zero_()

  def _make_layer(self, planes, blocks, stride=1, dilation=1):
    out_ch = planes * Bottleneck.expansion
    downsample = None
    if stride != 1 or self.inplanes != out_ch:
      downsample = nn.Sequential(nn.Conv2d(self.inplanes, out_ch, 1, stride=stride, bias=False),
                                 _bn(out_ch))
    if dilation in (1, 2):
      first = 1
    elif dilation == 4:
      first = 2
    else:
      raise RuntimeError('=> unknown dilation size: {}'.format(dilation))
    layers = [Bottleneck(self.inplanes, planes, stride, dilation=first, downsample=downsample)]
    self.inplanes = out_ch
    layers += [Bottleneck(self.inplanes, planes, dilation=dilation) for _ in range(1, blocks)]
    return nn.Sequential(*layers)

  def forward(self, x):
    x = self.conv1(x)
    res2 = self.res2(x)
    res3 = self.res3(res2)
    res4 = self.res4(res3)
    res5 = self.res5(res4)
    return res2, res3, res4, res5
