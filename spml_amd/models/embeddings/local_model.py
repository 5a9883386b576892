"""Location (and optionally smoothed colour) features per pixel
(`spml/models/embeddings/local_model.py`)."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.parameter import Parameter

import spml_amd.utils.segsort.common as segsort_common


class GaussianConv2d(nn.Module):
  """Fixed depth-wise blur (local_model.py:13-35; the reference's kernel is a
  normalised distance map, kept as is)."""

  def __init__(self, in_channels, out_channels, ksize=5):
    super().__init__()
    w = (np.arange(ksize, dtype=np.float32) - ksize // 2) ** 2
    w = np.sqrt(w[None, :] + w[:, None])
    w = np.reshape(w, (1, 1, ksize, ksize)) / w.sum()
    # a fixed kernel: keeps the reference's state-dict key, but it is only ever used under
    # no_grad, so it must not ask for a gradient (DistributedDataParallel would wait for one)
    self.weight = Parameter(torch.Tensor(w).expand(out_channels, -1, -1, -1).contiguous(),
                            requires_grad=False)
    self._in_channels = in_channels

  def forward(self, x):
    with torch.no_grad():
      return F.conv2d(x, self.weight, groups=self._in_channels)


class LocationColorNetwork(nn.Module):
  """`[N,H,W,2(+3)]`: (y, x) in [-0.5, 0.5] then optional colours (local_model.py:38-119)."""

  def __init__(self, use_color=True, use_location=True, norm_color=True, smooth_ksize=None):
    super().__init__()
    self._use_color, self._use_location = use_color, use_location
    self._norm_color, self._smooth_ksize = norm_color, smooth_ksize
    self.smooth_kernel = GaussianConv2d(3, 3, smooth_ksize) if smooth_ksize else nn.Identity()

  def __repr__(self):
    return 'LocationColorNetwork(use_color={}, use_location={}, smooth_ksize={})'.format(
        self._use_color, self._use_location, self._smooth_ksize)

  def forward(self, x, size=None):
    n, c, h, w = x.shape
    if size:
      h, w = size
    feats = []
    if self._use_location:
      loc = segsort_common.generate_location_features((h, w), x.device, 'float') - 0.5
      feats.append(loc.unsqueeze(0).expand(n, h, w, 2))
    if self._use_color:
      x = self.smooth_kernel(x)
      if size:
        x = F.interpolate(x, size=size, mode='bilinear')
      col = x.permute(0, 2, 3, 1).contiguous()
      if self._norm_color:
        col = col - col.view(n, -1, c).mean(dim=1).view(n, 1, 1, c)
        col = col / col.view(n, -1, c).abs().max(dim=1)[0].view(n, 1, 1, c)
      feats.append(col)
    return torch.cat(feats, dim=-1)
