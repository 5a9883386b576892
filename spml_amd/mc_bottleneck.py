"""Stride-1 bottleneck unit of the ResNet backbone (spml/models/backbones/resnet.py:11-63) on the
matrix-core convolutions of `csrc/conv.hip`.

The unit's three (four with a downsample branch) convolutions run as split-f16 implicit GEMMs at
fp32-class accuracy, and the batch norms between them hand the activations over in the split
("hl8") form directly:

    x (fp32 + hl8) -conv1-> a1 -bn1,relu-> y1 (hl8 only) -conv2-> a2 -bn2,relu-> y2 (hl8 only)
      -conv3-> a3 -bn3 (+ identity), relu-> out (fp32 + hl8)

Backward mirrors it: every batch-norm backward writes the gradient of its convolution as hl8
(scaled by a bound derived from the per-channel extremes of the reduction pass), the data
gradients come from the same GEMM kernel on transposed weights, the gradient of the residual
branch is accumulated in the epilogue of conv1's data gradient.  One autograd node per unit.
SyncBatchNorm: per-rank statistics are combined between the statistics pass and the apply pass
(all_gather / all_reduce of [C]-sized vectors), as in `ops._BatchNormAct`.
"""
import os

import torch
import torch.distributed as dist

from spml_amd import _ffi
from spml_amd._cache import BoundedCache


def _group_of(bn):
  if isinstance(bn, torch.nn.SyncBatchNorm) and dist.is_available() and dist.is_initialized():
    return bn.process_group if bn.process_group is not None else dist.group.WORLD
  return None


def available(block, x, allow_forward_only=True):
  """Training-mode, channels-last fp32 GPU input, stride 1, channel counts the kernels tile."""
  if os.environ.get('SPML_NO_MC_CONV') == '1' or not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
    return False
  if not block.training or block.stride != 1 or not x.is_contiguous(memory_format=torch.channels_last):
    return False
  convs = [block.conv1, block.conv2, block.conv3] + ([block.downsample[0]] if block.downsample is not None else [])
  # units with 128-multiple channel counts only (res3) run the 128-wide tiles; SPML_MC_NARROW_UNITS=0 keeps
  # them on the framework convolutions
  if os.environ.get('SPML_MC_NARROW_UNITS') == '0' and any(
      (c.in_channels & 255) or (c.out_channels & 255) for c in convs):
    return False
  bns = [block.bn1, block.bn2, block.bn3] + ([block.downsample[1]] if block.downsample is not None else [])
  # a frozen unit behind a frozen input (res2 of the training recipes: in no optimizer group, batch norms still in
  # training mode) only ever runs forward: its channel counts need no data- / weight-gradient tiles.  Opt-in
  # (SPML_MC_FROZEN_UNITS=1): at 64 channels the 64-column tiles are no faster than the library -- 122.0 / 120.5 ms
  # per step with res2 on this path against 120.0 / 120.2 without, alternating runs on one box
  forward_only = allow_forward_only and not (torch.is_grad_enabled() and (x.requires_grad or any(
      p.requires_grad for m in convs + bns for p in m.parameters())))
  if forward_only and any(c.out_channels & 127 for c in convs) and os.environ.get('SPML_MC_FROZEN_UNITS') != '1':
    return False
  for c in convs:
    taps = c.kernel_size[0] * c.kernel_size[1]
    if c.stride != (1, 1) or c.groups != 1 or c.bias is not None or taps not in (1, 9):
      return False
    if taps == 9 and (c.padding != c.dilation or c.dilation[0] != c.dilation[1]):
      return False
    if not _ffi.conv_hl8_supported(c.in_channels, c.out_channels, taps):
      return False
    if not forward_only and not (_ffi.conv_hl8_supported(c.out_channels, c.in_channels, taps) and
                                 _ffi.conv_wgrad_hl8_supported(c.in_channels, c.out_channels, taps)):
      return False
  return all(b.affine and b.track_running_stats and b.momentum is not None for b in bns)


def eval_available(block, x):
  """Inference (eval mode, no autograd): same shape conditions as the training path."""
  if block.training or torch.is_grad_enabled():
    return False
  block.training = True                  # reuse the shape / layout checks of the training path
  try:
    return available(block, x, allow_forward_only=False)
  finally:
    block.training = False


def _touch(bn):
  """The fused kernels update the running statistics through raw pointers, which does not bump the
  tensors' version counters; the eval-mode fold below keys its cache on them."""
  if bn.running_mean is not None:
    torch.autograd.graph.increment_version((bn.running_mean, bn.running_var))


import weakref

_folded = weakref.WeakKeyDictionary()       # conv module -> (version key, folded weights, bias); dies with the module


def _fold(conv, bn):
  """Batch norm in eval mode folded into the convolution: weight * (gamma * invstd)[co] in both hl8
  layouts + bias = beta - mean * gamma * invstd; cached per module (weakly: not carried by deepcopy / torch.save of
  the model) until a parameter or statistic
  changes (version counters; every in-library update of the statistics goes through `_touch`)."""
  ver = tuple(t._version for t in (conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var)) + \
      (conv.weight.data_ptr(), bn.running_mean.data_ptr(), bn.eps)
  hit = _folded.get(conv)
  if hit is not None and hit[0] == ver:
    return hit[1], hit[2]
  scale = bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps)
  w = conv.weight.detach() * scale.view(-1, 1, 1, 1)
  bias = (bn.bias.detach() - bn.running_mean * scale).contiguous()
  wf, _ = _ffi.hl8_weight(w)
  _folded[conv] = (ver, wf, bias)
  return wf, bias


def bottleneck_forward_eval(block, x):
  """Inference forward of one Bottleneck: three (four) matrix-core convolutions with the folded batch
  norm, ReLU and the residual add in their epilogues; one conversion pass between them."""
  n, cin, h, w = x.shape
  xh = getattr(x, '_spml_hl8', None) or _ffi.hl8_from_f32(x)
  dil = block.conv2.dilation[0]
  if block.downsample is not None:
    wd, bd = _fold(block.downsample[0], block.downsample[1])
    identity, _ = _ffi.conv_hl8_affine(xh, wd, bd, n, h, w, 1, relu=False, want_hl8=False)
  else:
    identity = x
  w1, b1 = _fold(block.conv1, block.bn1)
  w2, b2 = _fold(block.conv2, block.bn2)
  w3, b3 = _fold(block.conv3, block.bn3)
  _, y1 = _ffi.conv_hl8_affine(xh, w1, b1, n, h, w, 1)
  _, y2 = _ffi.conv_hl8_affine(y1, w2, b2, n, h, w, 9, dil)
  out, oh = _ffi.conv_hl8_affine(y2, w3, b3, n, h, w, 1, addend=identity)
  out._spml_hl8 = oh
  return out


def bump_counters(counters):
  """`num_batches_tracked += 1` for every batch norm of a unit in one launch."""
  if counters:
    torch._foreach_add_(counters, 1)


class _Bn(object):
  """Forward half of one batch norm inside the unit + what its backward needs."""

  def __init__(self, bn, a, rows, channels, residual=None, residual_bound=None, relu=True, want_f32=False,
               want_hl8=True, chunk_stats=None, counters=None, gathered=None):
    # gathered: (the ranks' [world, 3, C] statistics, this rank's [5, C]) if the caller has already exchanged them
    # (`sync_pair`: two independent batch norms share ONE all_gather)
    # counters (a list): the caller bumps `num_batches_tracked` of all its batch norms with ONE launch
    # (`bump_counters`) instead of one tiny kernel per layer (141 per training step)
    group = _group_of(bn)
    world = dist.get_world_size(group) if group is not None else 1
    self.count, self.group, self.world, self.relu = rows * world, group, world, relu
    if world == 1:                       # everything inside the library: three launches
      if bn.num_batches_tracked is not None:
        if counters is not None:
          counters.append(bn.num_batches_tracked)
        else:
          bn.num_batches_tracked.add_(1)
      self.y, self.yh, self.bound, self.mask, self.saved = _ffi.bn_fwd_hl8(
          a, rows, channels, residual, residual_bound, bn.weight, bn.bias, bn.running_mean, bn.running_var,
          bn.momentum, bn.eps, relu, want_f32, want_hl8, relu, chunk_stats=chunk_stats)
      _touch(bn)
      return
    # SyncBatchNorm: (count, mean, M2) of the ranks are gathered, pooled and finalised in one launch;
    # the channel extremes stay local (they only have to bound this rank's tensor)
    if gathered is None:
      st = _ffi.bn_stats_ext(a, rows, channels, chunk_stats=chunk_stats)
      allst = torch.empty((world * 3, channels), dtype=st.dtype, device=st.device)
      dist.all_gather_into_tensor(allst, st[:3], group=group)       # one contiguous [world, 3, C] block
      allst = allst.view(world, 3, channels)
    else:
      allst, st = gathered
    if bn.num_batches_tracked is not None:
      if counters is not None:
        counters.append(bn.num_batches_tracked)
      else:
        bn.num_batches_tracked.add_(1)
    # (the pooled row count stays on the device: the ranks' counts may differ, batchnorm.py:124-145)
    mean, invstd, self.count = _ffi.bn_finalize_ranks(allst, bn.eps, bn.momentum, bn.running_mean, bn.running_var)
    _touch(bn)
    cmax, cmin = st[3], st[4]
    self.y, self.yh, self.bound, self.mask = _ffi.bn_act_apply_hl8(
        a, rows, channels, residual, residual_bound, mean, invstd, bn.weight, bn.bias, cmax, cmin, relu,
        want_f32, want_hl8, want_mask=relu)
    self.saved = (mean, invstd, cmax, cmin)


def _bn_backward(dy, a, rows, channels, gamma, saved, count, group, world, mask, want_dres=False,
                 want_dx_f32=False):
  """-> (dx fp32 | None, dx hl8, d_residual | None, d_gamma, d_beta)"""
  if world == 1:
    dxh, dres, d_gamma, d_beta = _ffi.bn_bwd_hl8(dy, mask, a, rows, channels, saved, gamma, want_dres=want_dres)
    return None, dxh, dres, d_gamma, d_beta
  mean, invstd, cmax, cmin = saved
  st = _ffi.bn_act_bwd_reduce_ext(dy, None, mask, a, rows, channels, mean, invstd)   # (sum dz, sum dz*xhat, max|dz|)
  local = st[:2].clone()                           # local sums: DDP averages parameter gradients
  d_beta, d_gamma = local[0], local[1]
  dist.all_reduce(st[:2], group=group)
  dx, dxh, dres = _ffi.bn_act_bwd_apply_hl8(dy, None, mask, a, rows, channels, mean, invstd, gamma, st[0], st[1],
                                            st[2], cmax, cmin, count, want_dx_f32=want_dx_f32, want_dx_hl8=True,
                                            want_dres=want_dres)
  return dx, dxh, dres, d_gamma, d_beta


def sync_pair(bn_a, a_a, chunk_a, bn_b, a_b, chunk_b, rows, channels):
  """Two synchronised batch norms over independent tensors of the same width (a unit's third convolution and its
  downsample convolution): the local statistics of both travel in ONE all_gather of [3, 2 C] per rank instead of two
  of [3, C].  -> (gathered_a, gathered_b) [world, 3, C] each, or (None, None) on the unsynchronised path."""
  group = _group_of(bn_a)
  if group is None or _group_of(bn_b) is not group or dist.get_world_size(group) == 1:
    return None, None
  world = dist.get_world_size(group)
  st_a = _ffi.bn_stats_ext(a_a, rows, channels, chunk_stats=chunk_a)
  st_b = _ffi.bn_stats_ext(a_b, rows, channels, chunk_stats=chunk_b)
  loc = torch.cat([st_a[:3], st_b[:3]], dim=1)                       # [3, 2 C]
  allst = torch.empty((world * 3, 2 * channels), dtype=loc.dtype, device=loc.device)
  dist.all_gather_into_tensor(allst, loc, group=group)
  allst = allst.view(world, 3, 2 * channels)
  return (allst[:, :, :channels].contiguous(), st_a), (allst[:, :, channels:].contiguous(), st_b)


def _bn_backward_pair(d_out, mask, a3, gamma3, saved3, ad, gammad, savedd, rows, channels, count, group):
  """Backward of a unit's third batch norm (+ identity, ReLU) and of its downsample batch norm, synchronised: both
  see dz = d_out * mask (the downsample branch IS the residual), so both (sum dz, sum dz x-hat) pairs are reduced
  before ONE all_reduce of [2, 2 C], and the gradient of the residual branch is never materialised.
  -> (da3 hl8, dad hl8, d_gamma3, d_beta3, d_gammad, d_betad)"""
  mean3, invstd3, cmax3, cmin3 = saved3
  meand, invstdd, cmaxd, cmind = savedd
  st3 = _ffi.bn_act_bwd_reduce_ext(d_out, None, mask, a3, rows, channels, mean3, invstd3)
  std = _ffi.bn_act_bwd_reduce_ext(d_out, None, mask, ad, rows, channels, meand, invstdd)
  both = torch.cat([st3[:2], std[:2]], dim=1)                  # [2, 2 C]
  local = both.clone()                                        # local sums: DDP averages parameter gradients
  dist.all_reduce(both, group=group)
  c = channels
  _, da3, _ = _ffi.bn_act_bwd_apply_hl8(d_out, None, mask, a3, rows, c, mean3, invstd3, gamma3, both[0, :c].contiguous(),
                                        both[1, :c].contiguous(), st3[2], cmax3, cmin3, count)
  _, dad, _ = _ffi.bn_act_bwd_apply_hl8(d_out, None, mask, ad, rows, c, meand, invstdd, gammad, both[0, c:].contiguous(),
                                        both[1, c:].contiguous(), std[2], cmaxd, cmind, count)
  return da3, dad, local[1, :c], local[0, :c], local[1, c:], local[0, c:]


_side_streams = BoundedCache(16)      # one HIP stream per device, created once (a device resource, not call state)


def _side_stream(device):
  """Second stream for the weight gradients: they are off the backward's critical path, and the
  HBM-bound batch-norm passes of the next convolution overlap with them on the same CUs."""
  if os.environ.get('SPML_WGRAD_STREAM') == '0':
    return None
  return _side_streams.get_or_make(device.index, lambda: torch.cuda.Stream(device=device))


def _wgrad(side, dy, x, n, h, w, taps, dil=1):
  """conv_wgrad_hl8 on the side stream (after everything queued so far on the current one)."""
  if side is None:
    return _ffi.conv_wgrad_hl8(dy, x, n, h, w, taps, dil)
  main = torch.cuda.current_stream()
  side.wait_stream(main)
  with torch.cuda.stream(side):
    dw = _ffi.conv_wgrad_hl8(dy, x, n, h, w, taps, dil)
  # both halves of dy are freed by the caller before the side stream is joined: the 512-byte
  # bound block is read by the LAST kernel of the side stream (conv_wgrad_reduce) and would
  # otherwise be handed to the next batch-norm backward's `bound` on the main stream
  dy.data.record_stream(side)
  dy.bound.record_stream(side)
  dw.record_stream(main)
  return dw


class _Unit(torch.autograd.Function):

  @staticmethod
  def forward(ctx, block, x, xh_data, xh_bound, w1, g1, b1, w2, g2, b2, w3, g3, b3, wd, gd, bd):
    n, cin, h, w = x.shape
    rows = n * h * w
    dil = block.conv2.dilation[0]
    width, cout = w1.shape[0], w3.shape[0]
    xh = _ffi.Hl8(xh_data, xh_bound, rows, cin) if xh_data is not None else _ffi.hl8_from_f32(x)
    wset = _ffi.hl8_weight_set([w1, w2, w3] + ([wd] if wd is not None else []))
    (w1f, w1t), (w2f, w2t), (w3f, w3t) = wset[:3]
    # every convolution feeds a batch norm: where the tiling allows it (256-column tiles) the convolution's
    # epilogue leaves the chunk statistics and the batch norm does not read the tensor for them
    conv = _ffi.conv_hl8_stats
    a1, s1 = conv(xh, w1f, n, h, w, 1)
    counters = []
    n1 = _Bn(block.bn1, a1, rows, width, chunk_stats=s1, counters=counters)
    a2, s2 = conv(n1.yh, w2f, n, h, w, 9, dil)
    n2 = _Bn(block.bn2, a2, rows, width, chunk_stats=s2, counters=counters)
    a3, s3 = conv(n2.yh, w3f, n, h, w, 1)
    nd = ad = wdt = None
    ex3 = None                             # (exchanged statistics of bn3, if it shares the exchange)
    if wd is not None:
      wdf, wdt = wset[3]
      ad, sd = conv(xh, wdf, n, h, w, 1)
      # the statistics of the third convolution and of the downsample convolution are independent: one exchange
      ex3, exd = sync_pair(block.bn3, a3, s3, block.downsample[1], ad, sd, rows, cout)
      nd = _Bn(block.downsample[1], ad, rows, cout, relu=False, want_f32=True, want_hl8=False, chunk_stats=sd,
               counters=counters, gathered=exd)
      residual, res_bound = nd.y, nd.bound
    else:
      residual, res_bound = x, xh.bound
    n3 = _Bn(block.bn3, a3, rows, cout, residual=residual, residual_bound=res_bound, want_f32=True, chunk_stats=s3,
             counters=counters, gathered=ex3)
    bump_counters(counters)
    ctx.block, ctx.geom = block, (n, cin, h, w, dil, width, cout)
    ctx.bn_meta = [(m.count, m.group, m.world) for m in (n1, n2, n3)] + \
        ([(nd.count, nd.group, nd.world)] if nd is not None else [])
    tensors = [xh.data, xh.bound, a1, n1.yh.data, n1.yh.bound, a2, n2.yh.data, n2.yh.bound, a3, n3.mask,
               n1.mask, n2.mask,
               w1t.data, w1t.bound, w2t.data, w2t.bound, w3t.data, w3t.bound, g1, g2, g3]
    tensors += list(n1.saved) + list(n2.saved) + list(n3.saved)
    if nd is not None:
      tensors += [ad, wdt.data, wdt.bound, gd] + list(nd.saved)
    ctx.save_for_backward(*tensors)
    ctx.mark_non_differentiable(n3.yh.data, n3.yh.bound)
    ctx.set_materialize_grads(False)       # no zero-filled 0.3-0.5 GB "gradients" for the hl8 side outputs
    return n3.y, n3.yh.data, n3.yh.bound

  @staticmethod
  def backward(ctx, d_out, _unused_data, _unused_bound):
    if d_out is None:
      return (None,) * 16
    t = ctx.saved_tensors
    n, cin, h, w, dil, width, cout = ctx.geom
    rows = n * h * w
    (xh_d, xh_b, a1, y1_d, y1_b, a2, y2_d, y2_b, a3, m3, m1, m2, w1t_d, w1t_b, w2t_d, w2t_b, w3t_d, w3t_b, g1, g2,
     g3) = t[:21]
    s1, s2, s3 = t[21:25], t[25:29], t[29:33]
    has_ds = len(t) > 33
    H = _ffi.Hl8
    xh, y1h, y2h = H(xh_d, xh_b, rows, cin), H(y1_d, y1_b, rows, width), H(y2_d, y2_b, rows, width)
    w1t, w2t, w3t = H(w1t_d, w1t_b, cin, width), H(w2t_d, w2t_b, width, 9 * width), H(w3t_d, w3t_b, width, cout)
    if not d_out.is_contiguous(memory_format=torch.channels_last):
      d_out = d_out.contiguous(memory_format=torch.channels_last)
    need_x = ctx.needs_input_grad[1]
    m = ctx.bn_meta
    # bn3 (+ identity, relu)
    # the residual branch's gradient d_out * mask3 is only materialised for the downsample branch; the
    # plain identity takes it in the epilogue of conv1's data gradient (masked addend)
    dwd = dgd = dbd = dad = None
    pair = has_ds and m[2][2] > 1 and m[3][1] is m[2][1]      # synchronised: both reductions before ONE all_reduce
    if pair:
      da3, dad, dg3, db3, dgd, dbd = _bn_backward_pair(d_out, m3, a3, g3, s3, t[33], t[36], t[37:41], rows, cout,
                                                       m[2][0], m[2][1])
      dres = None
    else:
      _, da3, dres, dg3, db3 = _bn_backward(d_out, a3, rows, cout, g3, s3, m[2][0], m[2][1], m[2][2], m3,
                                            want_dres=has_ds)
    side = _side_stream(d_out.device)
    dw3 = _wgrad(side, da3, y2h, n, h, w, 1)
    dy2 = _ffi.conv_hl8(da3, w3t, n, h, w, 1)
    del da3
    _, da2, _, dg2, db2 = _bn_backward(dy2, a2, rows, width, g2, s2, m[1][0], m[1][1], m[1][2], m2)
    dw2 = _wgrad(side, da2, y1h, n, h, w, 9, dil)
    dy1 = _ffi.conv_hl8(da2, w2t, n, h, w, 9, dil)
    del da2, dy2
    _, da1, _, dg1, db1 = _bn_backward(dy1, a1, rows, width, g1, s1, m[0][0], m[0][1], m[0][2], m1)
    dw1 = _wgrad(side, da1, xh, n, h, w, 1)
    dx = None
    if has_ds:
      ad, wdt_d, wdt_b, gd = t[33:37]
      sd = t[37:41]
      if not pair:
        _, dad, _, dgd, dbd = _bn_backward(dres, ad, rows, cout, gd, sd, m[3][0], m[3][1], m[3][2], None)
      dwd = _wgrad(side, dad, xh, n, h, w, 1)
      if need_x:
        dx = _ffi.conv_hl8(da1, w1t, n, h, w, 1)
        dx = _ffi.conv_hl8(dad, H(wdt_d, wdt_b, cin, cout), n, h, w, 1, addend=dx)
    elif need_x:
      dx = _ffi.conv_hl8(da1, w1t, n, h, w, 1, addend=d_out, addend_mask=m3)
    if side is not None:
      torch.cuda.current_stream().wait_stream(side)      # the weight gradients are consumed on this stream
    return (None, dx, None, None, dw1, dg1, db1, dw2, dg2, db2, dw3, dg3, db3, dwd, dgd, dbd)


class _ConvBnAct(torch.autograd.Function):
  """relu(bn(conv(x))) for ONE stride-1 convolution (1x1 or dilated 3x3) followed by a training-mode batch
  norm, on the same kernels as the units: the 3x3 convolution 4096 -> 512 that closes the pyramid-pooling head
  (spml/models/heads/spp.py:46-86) is 30 % of the DensePose step on the fp32 library (22 + 22 + 20 ms)."""

  @staticmethod
  def forward(ctx, conv, bn, x, weight, gamma, beta):
    n, cin, h, w = x.shape
    rows, cout = n * h * w, weight.shape[0]
    taps, dil = weight.shape[2] * weight.shape[3], conv.dilation[0]
    xh = getattr(x, '_spml_hl8', None) or _ffi.hl8_from_f32(x)
    (wf, wt), = _ffi.hl8_weight_set([weight])
    a, st = _ffi.conv_hl8_stats(xh, wf, n, h, w, taps, dil)
    nb = _Bn(bn, a, rows, cout, relu=True, want_f32=True, want_hl8=False, chunk_stats=st)
    ctx.geom = (n, cin, h, w, taps, dil, cout)
    ctx.bn_meta = (nb.count, nb.group, nb.world)
    ctx.save_for_backward(xh.data, xh.bound, a, nb.mask, wt.data, wt.bound, gamma, *nb.saved)
    return nb.y

  @staticmethod
  def backward(ctx, dy):
    n, cin, h, w, taps, dil, cout = ctx.geom
    rows = n * h * w
    xh_d, xh_b, a, mask, wt_d, wt_b, gamma = ctx.saved_tensors[:7]
    saved = ctx.saved_tensors[7:11]
    if not dy.is_contiguous(memory_format=torch.channels_last):
      dy = dy.contiguous(memory_format=torch.channels_last)
    count, group, world = ctx.bn_meta
    _, da, _, dg, db = _bn_backward(dy, a, rows, cout, gamma, saved, count, group, world, mask)
    side = _side_stream(dy.device)
    dw = _wgrad(side, da, _ffi.Hl8(xh_d, xh_b, rows, cin), n, h, w, taps, dil)
    dx = None
    if ctx.needs_input_grad[2]:
      dx = _ffi.conv_hl8(da, _ffi.Hl8(wt_d, wt_b, cin, taps * cout), n, h, w, taps, dil)
    if side is not None:
      torch.cuda.current_stream().wait_stream(side)
    return None, None, dx, dw, dg, db


def conv_bn_act_available(conv, bn, x):
  """One stride-1 convolution + training-mode batch norm + ReLU on the matrix-core path: channels-last fp32
  GPU input, channel counts the forward, data-gradient and weight-gradient kernels tile."""
  if os.environ.get('SPML_NO_MC_CONV') == '1' or not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
    return False
  if not (bn.training and torch.is_grad_enabled() and x.is_contiguous(memory_format=torch.channels_last)):
    return False
  k = conv.kernel_size
  taps = k[0] * k[1]
  cin, cout = conv.in_channels, conv.out_channels
  return (conv.bias is None and conv.groups == 1 and conv.stride == (1, 1) and k in ((1, 1), (3, 3)) and
          conv.dilation[0] == conv.dilation[1] and conv.padding == tuple(d * (k[0] // 2) for d in conv.dilation) and
          bn.affine and bn.track_running_stats and
          _ffi.conv_hl8_supported(cin, cout, taps) and _ffi.conv_hl8_supported(cout, cin, taps) and
          _ffi.conv_wgrad_hl8_supported(cin, cout, taps))


def conv_bn_act(conv, bn, x):
  return _ConvBnAct.apply(conv, bn, x, conv.weight, bn.weight, bn.bias)


def bottleneck_forward(block, x):
  """Forward of one Bottleneck through the matrix-core path; the output tensor carries its hl8
  copy (attribute `_spml_hl8`) for the next unit."""
  xh = getattr(x, '_spml_hl8', None)
  ds = block.downsample
  out, oh, ob = _Unit.apply(
      block, x, None if xh is None else xh.data, None if xh is None else xh.bound,
      block.conv1.weight, block.bn1.weight, block.bn1.bias, block.conv2.weight, block.bn2.weight, block.bn2.bias,
      block.conv3.weight, block.bn3.weight, block.bn3.bias,
      None if ds is None else ds[0].weight, None if ds is None else ds[1].weight, None if ds is None else ds[1].bias)
  out._spml_hl8 = _ffi.Hl8(oh, ob, out.shape[0] * out.shape[2] * out.shape[3], out.shape[1])
  return out
