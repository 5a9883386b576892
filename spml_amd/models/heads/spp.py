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
