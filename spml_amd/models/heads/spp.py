"""Atrous spatial pyramid head of DeepLab-v2 (`spml/models/heads/spp.py:8-43`):
four dilated 3x3 branches (6/12/18/24) whose outputs are SUMMED."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from spml_amd._cache import BoundedCache

from spml_amd.nn.batchnorm import BatchNorm2d


class _DilatedSum(torch.autograd.Function):
  """sum_i conv2d(x, w_i, b_i, dilation = padding = d_i) for 3x3 weights, on the matrix-core kernels.  The data
  gradient -- four convolutions of the SAME output gradient, summed -- is one 36-tap launch
  (`spml_conv_hl8_pyramid_f32`) instead of four library calls and three additions over the 2048-channel tensor.
  Narrow heads (64 output channels): forward = one 1x1 convolution with 36 x 64 columns + a gather of every
  pixel's taps (`spml_conv_tap_gather_f32`), weight gradients = one launch on tiles of four taps x 64 channels
  (`spml_conv_wgrad_pyramid_hl8_f32`).  Wide heads (256-multiple channels): the 36-tap launch forward, one
  weight-gradient call per branch."""

  @staticmethod
  def forward(ctx, x, dilations, *params):
    from spml_amd import _ffi
    ws, bs = params[0::2], params[1::2]
    cout, cin = ws[0].shape[0], ws[0].shape[1]
    n, _, h, w = x.shape
    # wide heads (256-multiple output channels, e.g. the 512-d embedding of BASELINE config 5): forward and
    # weight gradients on the matrix-core kernels too.  The 64-channel head: forward on the 36-tap kernel with
    # one workgroup per (256-pixel tile, dilation group) and taps that only see padding skipped (3.9 ms against
    # 4.9 ms for the four library calls + three additions, tools/bench_conv.py --narrow; SPML_ASPP_FWD_MC=0
    # keeps the library); its weight gradients are ONE launch whose 256-column tiles are four taps x 64 channels
    # (`spml_conv_wgrad_pyramid_hl8_f32`; SPML_ASPP_WGRAD_MC=0 keeps the four library calls)
    ctx.wide = _ffi.conv_hl8_supported(cin, cout, 9) and _ffi.conv_wgrad_hl8_supported(cin, cout, 9)
    fwd_mc = ctx.wide or (_ffi.conv_hl8_supported(cin, cout, 9) and os.environ.get('SPML_ASPP_FWD_MC') != '0')
    ctx.narrow_wgrad = (not ctx.wide and _ffi.conv_wgrad_pyramid_hl8_supported(cin, cout, len(ws)) and
                        all(1 <= d <= 255 for d in dilations) and os.environ.get('SPML_ASPP_WGRAD_MC') != '0')
    ctx.xh = None
    if fwd_mc or ctx.narrow_wgrad:
      xh = getattr(x, '_spml_hl8', None) or _ffi.hl8_from_f32(x)
      ctx.xh = xh if (ctx.wide or ctx.narrow_wgrad) else None
    if fwd_mc and not ctx.wide and os.environ.get('SPML_ASPP_FWD_GEMM') != '0' and \
        _ffi.conv_hl8_pyramid_forward_gemm_supported(cin, cout, len(ws)) and all(1 <= d <= 255 for d in dilations):
      # narrow head: one 1x1 convolution with 36 x 64 columns + a gather of every pixel's taps (x is streamed 9
      # times instead of 36; SPML_ASPP_FWD_GEMM=0: the 36-tap launch on 64-column tiles)
      out = _ffi.conv_hl8_pyramid_forward_gemm(xh, ws, bs, dilations, n, h, w)
    elif fwd_mc:
      out = _ffi.conv_hl8_pyramid_forward(xh, ws, bs, dilations, n, h, w)
    else:
      out = None
      for wt, b, d in zip(ws, bs, dilations):
        y = F.conv2d(x, wt, b, 1, d, d)
        out = y if out is None else out.add_(y)
    ctx.dilations = dilations
    ctx.save_for_backward(x, *ws)
    ctx.has_bias = [b is not None for b in bs]
    return out

  @staticmethod
  def backward(ctx, dy):
    from spml_amd import _ffi
    x, ws = ctx.saved_tensors[0], ctx.saved_tensors[1:]
    dy = dy.contiguous(memory_format=torch.channels_last)
    n, _, h, w = x.shape
    grads = []
    any_w = any(ctx.needs_input_grad[2 + 2 * i] for i in range(len(ws)))
    own = ctx.wide or ctx.narrow_wgrad
    dyh = _ffi.hl8_from_f32(dy) if (ctx.needs_input_grad[0] or (own and any_w)) else None
    db_all = dy.sum(dim=(0, 2, 3)) if own and any(ctx.has_bias) else None
    dws = _ffi.conv_wgrad_pyramid_hl8(dyh, ctx.xh, n, h, w, ctx.dilations) if ctx.narrow_wgrad and any_w else None
    for i, (wt, d) in enumerate(zip(ws, ctx.dilations)):
      need_w, need_b = ctx.needs_input_grad[2 + 2 * i], ctx.has_bias[i] and ctx.needs_input_grad[3 + 2 * i]
      dw = db = None
      if own:
        if need_w:
          dw = dws[i] if ctx.narrow_wgrad else _ffi.conv_wgrad_hl8(dyh, ctx.xh, n, h, w, 9, d)
        if need_b:
          db = db_all
      elif need_w or need_b:
        _, dw, db = torch.ops.aten.convolution_backward(
            dy, x, wt, [wt.shape[0]] if ctx.has_bias[i] else None, [1, 1], [d, d], [d, d], False, [0, 0], 1,
            [False, bool(need_w), bool(need_b)])
      grads += [dw, db]
    dx = None
    if ctx.needs_input_grad[0]:
      dx = _ffi.conv_hl8_pyramid_dgrad(dyh, ws, ctx.dilations, n, h, w)
    return (dx, None) + tuple(grads)


def _dilated_sum_available(x, convs):
  if os.environ.get('SPML_NO_MC_CONV') == '1' or os.environ.get('SPML_NO_ASPP_DGRAD') == '1' or not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
    return False
  from spml_amd import _ffi
  if not x.is_contiguous(memory_format=torch.channels_last):
    return False
  # (forward-only calls -- eval / no_grad: stage 2, the inference drivers -- take the own forward too: one 1x1
  # convolution with 36 x 64 columns + tap gather, 2.0 against 4.8 ms for the four library calls at batch 16, and
  # run-to-run stable, which MIOpen's dilated 3x3 2048 -> 64 forward is not; SPML_ASPP_EVAL_MC=0 keeps them on the
  # library unless the deterministic mode is on)
  if not ((x.requires_grad and torch.is_grad_enabled()) or _ffi.deterministic() or
          os.environ.get('SPML_ASPP_EVAL_MC') != '0'):
    return False
  c0 = convs[0]
  return (len(convs) <= 4 and all(
      c.kernel_size == (3, 3) and c.stride == (1, 1) and c.groups == 1 and c.padding == c.dilation and
      c.dilation[0] == c.dilation[1] and c.in_channels == c0.in_channels and c.out_channels == c0.out_channels
      for c in convs) and _ffi.conv_hl8_supported(c0.out_channels, c0.in_channels, 9))


class ASPP(nn.Module):

  def __init__(self, in_channels, out_channels, bn=True, relu=True):
    super().__init__()
    for i, dilation in enumerate((6, 12, 18, 24), start=1):
      branch = [nn.Conv2d(in_channels, out_channels, 3, 1, padding=dilation, dilation=dilation,
                          bias=not bn)]
      if bn:
        branch.append(BatchNorm2d(out_channels))
      if relu:
        branch.append(nn.ReLU(inplace=True))
      setattr(self, 'aspp_%d' % i, nn.Sequential(*branch))

  def forward(self, x):
    branches = (self.aspp_1, self.aspp_2, self.aspp_3, self.aspp_4)
    if all(len(b) == 1 for b in branches):            # DeepLab-v2: bare convolutions, summed
      convs = [b[0] for b in branches]
      if _dilated_sum_available(x, convs):
        params = []
        for c in convs:
          params += [c.weight, c.bias]
        return _DilatedSum.apply(x, tuple(c.dilation[0] for c in convs), *params)
    return self.aspp_1(x) + self.aspp_2(x) + self.aspp_3(x) + self.aspp_4(x)


_pool_matrices = BoundedCache(8)       # (constants of the map size; bounded: spml_amd/_cache.py)


def _pool_matrix(h, w, sizes, device):
  """[sum b*b, h*w] averaging matrix of AdaptiveAvgPool2d(b) for every b in `sizes` (bin i of a side of
  length n covers [floor(i n / b), ceil((i + 1) n / b)), as the framework op)."""
  def make():
    rows = []
    for b in sizes:
      for i in range(b):
        y0, y1 = (i * h) // b, -((-(i + 1) * h) // b)
        for j in range(b):
          x0, x1 = (j * w) // b, -((-(j + 1) * w) // b)
          r = torch.zeros(h, w)
          r[y0:y1, x0:x1] = 1.0 / ((y1 - y0) * (x1 - x0))
          rows.append(r.reshape(-1))
    return torch.stack(rows).to(device)
  return _pool_matrices.get_or_make((h, w, tuple(sizes), str(device)), make)


def _pyramid_pool_available(x, branches):
  if os.environ.get('SPML_NO_PYRAMID_POOL_GEMM') == '1':
    return False
  return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and
          x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous() and
          all(isinstance(b[0], nn.AdaptiveAvgPool2d) and isinstance(b[0].output_size, int) for b in branches))


def _pyramid_pool(x, sizes):
  """AdaptiveAvgPool2d(b)(x) for every b in `sizes` from one pass over a channels-last x."""
  n, c, h, w = x.shape
  flat = x.permute(0, 2, 3, 1).reshape(n, h * w, c)                 # a view of the channels-last storage
  out = torch.matmul(_pool_matrix(h, w, sizes, x.device), flat)    # [n, sum b*b, c]
  res, lo = [], 0
  for b in sizes:
    res.append(out[:, lo:lo + b * b].reshape(n, b, b, c).permute(0, 3, 1, 2))   # NCHW view, channels-last memory
    lo += b * b
  return res


_up_matrices = BoundedCache(16)


def _upsample_gemm(f, size):
  """F.interpolate(f, size, mode='bilinear') for a small map f [N, C, b, b] as one GEMM with the [H*W, b*b]
  interpolation matrix (the framework op applied to the unit maps, cached); channels-last in and out."""
  n, c, bh, bw = f.shape
  h, w = int(size[0]), int(size[1])
  def make():
    eye = torch.eye(bh * bw, device=f.device).view(bh * bw, 1, bh, bw)
    return F.interpolate(eye, size=(h, w), mode='bilinear').reshape(bh * bw, h * w).t().contiguous()
  m = _up_matrices.get_or_make((bh, bw, h, w, str(f.device)), make)
  flat = f.permute(0, 2, 3, 1).reshape(n, bh * bw, c)              # [N, b*b, C] (a copy only if f is not channels-last)
  return torch.matmul(m, flat).reshape(n, h, w, c).permute(0, 3, 1, 2)


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
        layers.append(BatchNorm2d(out_c))
      if relu:
        layers.append(nn.ReLU(inplace=True))
      return nn.Sequential(*layers)

    self.pspp_1 = block(in_channels, out_channels, 1, 1)
    self.pspp_2 = block(in_channels, out_channels, 1, 2)
    self.pspp_3 = block(in_channels, out_channels, 1, 3)
    self.pspp_4 = block(in_channels, out_channels, 1, 6)
    self.conv = block(in_channels + out_channels * 4, out_channels, 3, None)

  def forward(self, x):
    size = x.shape[-2:]
    branches = (self.pspp_1, self.pspp_2, self.pspp_3, self.pspp_4)
    if _pyramid_pool_available(x, branches):
      # the four adaptive average pools as ONE plain GEMM over the channels-last map (the framework's NHWC
      # adaptive pool kernel runs at 130 GB/s: 14 ms per step at 8 x 2048 x 97 x 97, a third of it per level)
      feats = [branch[1:](p) for branch, p in zip(branches, _pyramid_pool(x, [b[0].output_size for b in branches]))]
    else:
      feats = [branch(x) for branch in branches]
    if _pyramid_pool_available(x, branches):
      # ... and back up as plain GEMMs too: the framework's bilinear backward scatter-adds 154 MB of output
      # gradient into a b x b map with atomics (1.1 ms per branch at 8 x 512 x 97 x 97)
      pooled = [_upsample_gemm(f, size) for f in feats]
    else:
      pooled = [F.interpolate(f, size=size, mode='bilinear') for f in feats]
    cat = torch.cat([x] + pooled, dim=1)
    from spml_amd import mc_bottleneck
    if (len(self.conv) == 3 and isinstance(self.conv[1], nn.modules.batchnorm._BatchNorm) and isinstance(self.conv[2], nn.ReLU) and
        mc_bottleneck.conv_bn_act_available(self.conv[0], self.conv[1], cat)):
      return mc_bottleneck.conv_bn_act(self.conv[0], self.conv[1], cat)       # conv + bn + relu, matrix cores
    return self.conv(cat)
