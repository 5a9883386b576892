"""Atrous spatial pyramid head of DeepLab-v2 (`spml/models/heads/spp.py:8-43`):
four dilated 3x3 branches (6/12/18/24) whose outputs are SUMMED."""
import torch.nn as nn


class ASPP(nn.Module):

  def __init__(self, in_channels, out_channels, bn=True, relu=True):
    super().__init__()
    for i, dilation in enumerate((6, 12, 18, 24), start=1):
      branch = [nn.Conv2d(in_channels, out_channels, 3, 1, padding=dilation, dilation=dilation,
                          bias=not bn)]
      if bn:
        branch.append(nn.BatchNorm2d(out_channels))
      if relu:
        branch.append(nn.ReLU(inplace=True))
      setattr(self, 'aspp_%d' % i, nn.Sequential(*branch))

  def forward(self, x):
    return self.aspp_1(x) + self.aspp_2(x) + self.aspp_3(x) + self.aspp_4(x)


class PSPP(nn.Module):
  """Pyramid pooling head of PSPNet (`spml/models/heads/spp.py:46-86`): average pools to
  1/2/3/6 bins, 1x1 conv (+BN+ReLU) each, upsampled and concatenated with the input, then
  a 3x3 conv."""

  def __init__(self, in_channels, out_channels, bn=True, relu=True):
    super().__init__()

    def block(in_c, out_c, k, size):
      layers = [nn.AdaptiveAvgPool2d(size)] if size else []
      layers.append(nn.Conv2d(in_c, out_c, k, 1, (k - 1) // 2, 1, bias=not bn))
      if bn:
        layers.append(nn.BatchNorm2d(out_c))
      if relu:
        layers.append(nn.ReLU(inplace=True))
      return nn.Sequential(*layers)

    self.pspp_1 = block(in_channels, out_channels, 1, 1)
    self.pspp_2 = block(in_channels, out_channels, 1, 2)
    self.pspp_3 = block(in_channels, out_channels, 1, 3)
    self.pspp_4 = block(in_channels, out_channels, 1, 6)
    self.conv = block(in_channels + out_channels * 4, out_channels, 3, None)

  def forward(self, x):
    import torch
    import torch.nn.functional as F
    size = x.shape[-2:]
    pooled = [F.interpolate(branch(x), size=size, mode='bilinear')
              for branch in (self.pspp_1, self.pspp_2, self.pspp_3, self.pspp_4)]
    return self.conv(torch.cat([x] + pooled, dim=1))
