"""Differentiable operators over the C-ABI of libspml_hip.so.

Every function here takes GPU tensors and launches hand-written gfx950 kernels
through `spml_amd._ffi`; there is no CPU path (a CPU tensor or a missing library
raises `SpmlHipError`).  `torch.autograd.Function` is used only to connect the
forward and backward kernels."""
import torch
import torch.nn.functional as F

from . import _ffi

NLL_LABEL, NLL_TAGSET, NLL_PLAIN, NLL_CODE32 = 0, 1, 2, 4


def _f32c(t):
  t = t if t.dtype == torch.float32 else t.float()
  return t if t.is_contiguous() else t.contiguous()


def _i64c(t):
  t = t if t.dtype == torch.int64 else t.long()
  return t if t.is_contiguous() else t.contiguous()


# ---------------------------------------------------------------------------
class _NormalizeRows(torch.autograd.Function):

  @staticmethod
  def forward(ctx, x):
    x = _f32c(x)
    ctx.save_for_backward(x)
    return _ffi.normalize_rows(x)

  @staticmethod
  def backward(ctx, dy):
    (x,) = ctx.saved_tensors
    return _ffi.normalize_rows_bwd(x, _f32c(dy))


def normalize_rows(x):
  """x / max(||x||, 1e-12) over the last dim (general/common.py:101-120)."""
  return _NormalizeRows.apply(x)


# ---------------------------------------------------------------------------
class _NormalizeConcatLoc(torch.autograd.Function):

  @staticmethod
  def forward(ctx, emb, loc, row_map, num_rows):
    # a channels-last map is streamed as it is (row-wise kernels); anything else goes through NCHW
    if not (emb.is_cuda and _ffi.k1_channels_last(emb, 2 if loc is None else int(loc.shape[-1]))):
      emb = _f32c(emb)
    loc = None if loc is None else _f32c(loc)
    ctx.save_for_backward(emb, loc, row_map)
    out_emb, out_loc = _ffi.normalize_concat_loc(emb, loc, row_map, num_rows)
    return out_emb, out_loc

  @staticmethod
  def backward(ctx, d_emb_rows, d_loc_rows):
    emb, loc, row_map = ctx.saved_tensors
    d_emb_rows = None if d_emb_rows is None else _f32c(d_emb_rows)
    d_loc_rows = None if d_loc_rows is None else _f32c(d_loc_rows)
    return _ffi.normalize_concat_loc_bwd(emb, loc, row_map, d_emb_rows, d_loc_rows), None, None, None


K1_MAX_GRAD_CHANNELS = 512      # spml_normalize_concat_*_bwd_f32 keep the gradient rows in registers (csrc/normalize.hip)


def normalize_concat_loc(emb_nchw, loc=None, row_map=None, num_rows=None):
  """K1: NCHW map -> (unit rows [P',C], unit rows with location [P',C+2]).  A map that needs a gradient may have at
  most 512 channels: refused here, in the forward, not first in the backward."""
  if emb_nchw.requires_grad and torch.is_grad_enabled() and emb_nchw.shape[1] > K1_MAX_GRAD_CHANNELS:
    raise _ffi.SpmlHipError('the K1 backward kernels take at most %d channels (got %d)' % (
        K1_MAX_GRAD_CHANNELS, emb_nchw.shape[1]))
  return _NormalizeConcatLoc.apply(emb_nchw, loc, row_map, num_rows)


# ---------------------------------------------------------------------------
class _SegmentPrototypes(torch.autograd.Function):

  @staticmethod
  def forward(ctx, x, ids, m):
    x, ids = _f32c(x), _i64c(ids)
    protos, sums = _ffi.segment_sum_normalize(x, ids, m)
    ctx.save_for_backward(sums, ids)
    ctx.p = x.shape[0]
    return protos

  @staticmethod
  def backward(ctx, d_protos):
    sums, ids = ctx.saved_tensors
    return _ffi.segment_sum_normalize_bwd(_f32c(d_protos), sums, ids, ctx.p), None, None


def segment_prototypes(x, ids, m):
  """A4: normalize(scatter_add(x by ids)) -> [m, D]."""
  return _SegmentPrototypes.apply(x, ids, int(m))


# ---------------------------------------------------------------------------
class _SegSortNLL(torch.autograd.Function):

  @staticmethod
  def forward(ctx, emb, own, px_code, protos, pr_code, kappa, mode, m_grad):
    emb, protos = _f32c(emb), _f32c(protos)
    own, px_code, pr_code = _i64c(own), _i64c(px_code), _i64c(pr_code)
    nll, stats = _ffi.segsort_nll_fwd(emb, own, px_code, protos, pr_code, kappa, mode)
    ctx.save_for_backward(emb, own, px_code, protos, pr_code, stats)
    ctx.kappa, ctx.mode, ctx.m_grad = kappa, mode, m_grad
    return nll

  @staticmethod
  def backward(ctx, d_nll):
    emb, own, px_code, protos, pr_code, stats = ctx.saved_tensors
    d_emb, d_protos = _ffi.segsort_nll_bwd(emb, own, px_code, protos, pr_code, ctx.kappa,
                                           ctx.mode, stats, _f32c(d_nll), ctx.m_grad)
    return d_emb, None, None, d_protos, None, None, None, None


def segsort_nll(emb, own, px_code, protos, pr_code, kappa, mode=NLL_LABEL, proto_grad_rows=None):
  """A9/A10: per-pixel NCA negative log-likelihood [P].  `proto_grad_rows`: only the
  first that many prototypes need a gradient (the rest is e.g. a detached memory bank)."""
  if emb.shape[0] == 0:
    return emb.new_zeros((0,))
  m_grad = -1 if proto_grad_rows is None else int(proto_grad_rows)
  return _SegSortNLL.apply(emb, own, px_code, protos, pr_code, float(kappa), int(mode), m_grad)


# ---------------------------------------------------------------------------
def kmeans(x, seg_offsets, max_seg_len, k, labels_init, iterations, want_centroids=False):
  """A6 over a ragged batch (no gradient: labels are discrete)."""
  with torch.no_grad():
    return _ffi.kmeans_run(_f32c(x.detach()), _i64c(seg_offsets), int(max_seg_len), int(k),
                           _i64c(labels_init), int(iterations), want_centroids)


def kmeans_assign(x, seg_offsets, max_seg_len, centroids):
  with torch.no_grad():
    return _ffi.kmeans_assign(_f32c(x.detach()), _i64c(seg_offsets), int(max_seg_len),
                              _f32c(centroids.detach()))


def topk_affinity(q, protos, k, q_group=None, pr_group=None, pr_valid=None, masked_value=-2.0):
  with torch.no_grad():
    return _ffi.topk_affinity(
        _f32c(q.detach()), _f32c(protos.detach()), int(k),
        None if q_group is None else _i64c(q_group),
        None if pr_group is None else _i64c(pr_group),
        None if pr_valid is None else pr_valid.to(torch.uint8).contiguous(), masked_value)


def kmeans_init_grid(h, w, ky, kx, device):
  return _ffi.kmeans_init_grid(int(h), int(w), int(ky), int(kx), device)


# ---------------------------------------------------------------------------
def merge_bn_statistics(counts, means, m2s):
  """Chan's parallel-variance merge of per-rank batch-norm statistics: `[world, C]` counts,
  means and sums of squared deviations -> (total count [C], mean [C], M2 [C])."""
  total = counts.sum(0)
  mean = (means * counts).sum(0) / total
  m2 = (m2s + counts * (means - mean) ** 2).sum(0)
  return total, mean, m2


class _BatchNormAct(torch.autograd.Function):
  """relu?(batch_norm_train(x) [+ residual]) on channels-last fp32 activations: one statistics
  pass + one apply pass forward, one reduction pass + one apply pass backward."""

  @staticmethod
  def forward(ctx, x, residual, weight, bias, running_mean, running_var, momentum, eps, relu, group):
    import torch.distributed as dist
    n, c, h, w = x.shape
    count = n * h * w
    world = dist.get_world_size(group) if group is not None else 1
    ctx.count, ctx.group, ctx.world = count * world, group, world
    ctx.has_res = residual is not None
    if world == 1:        # everything inside the library: 3 launches, no framework ops in between
      y, mean, invstd = _ffi.bn_act_fwd(x, residual, weight, bias, running_mean, running_var, momentum,
                                        eps, relu)
      ctx.save_for_backward(x, y if relu else None, mean, invstd, weight)
      return y
    # SyncBatchNorm: the ranks' (count, mean, M2) are gathered, pooled (Chan) and finalised in one launch
    mean, m2 = _ffi.bn_stats(x)
    stats = torch.stack([torch.full_like(mean, float(count)), mean, m2])
    allst = torch.empty((world * 3, stats.shape[1]), dtype=stats.dtype, device=stats.device)
    dist.all_gather_into_tensor(allst, stats, group=group)        # one contiguous [world, 3, C] block
    allst = allst.view(world, 3, stats.shape[1])
    # the pooled row count = sum of the gathered counts (the ranks' batches may differ,
    # lib/nn/sync_batchnorm/batchnorm.py:124-145); it stays on the device for the backward
    mean, invstd, ctx.count = _ffi.bn_finalize_ranks(allst, eps, momentum if running_mean is not None else 0.0,
                                                     running_mean, running_var)
    y = _ffi.bn_act_apply(x, residual, mean, invstd, weight, bias, relu)
    ctx.save_for_backward(x, y if relu else None, mean, invstd, weight)
    return y

  @staticmethod
  def backward(ctx, dy):
    import torch.distributed as dist
    x, y, mean, invstd, weight = ctx.saved_tensors
    dy = dy if dy.is_contiguous(memory_format=torch.channels_last) else \
        dy.contiguous(memory_format=torch.channels_last)
    need_x, need_res = ctx.needs_input_grad[0], ctx.has_res and ctx.needs_input_grad[1]
    if ctx.world == 1:
      dx, dres, d_weight, d_bias = _ffi.bn_act_bwd(dy, y, x, mean, invstd, weight, want_dx=need_x,
                                                   want_dres=need_res)
      return dx, dres, d_weight, d_bias, None, None, None, None, None, None
    s0, s1 = _ffi.bn_act_bwd_reduce(dy, y, x, mean, invstd)
    d_weight, d_bias = s1.clone(), s0.clone()          # local sums: DDP averages parameter gradients
    if ctx.world > 1:
      both = torch.stack([s0, s1])
      dist.all_reduce(both, group=ctx.group)
      s0, s1 = both[0].contiguous(), both[1].contiguous()
    dx, dres = _ffi.bn_act_bwd_apply(dy, y, x, mean, invstd, weight, s0, s1, ctx.count, want_dx=need_x,
                                     want_dres=need_res)
    return dx, dres, d_weight, d_bias, None, None, None, None, None, None


def fused_bn_act_available(x, bn):
  """The fused kernels take fp32 channels-last GPU activations of a batch norm in training mode."""
  import os
  if os.environ.get('SPML_NO_FUSED_BN') == '1':
    return False
  return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and bn.training and bn.affine and
          bn.track_running_stats and x.shape[1] % 4 == 0 and
          x.is_contiguous(memory_format=torch.channels_last) and x.shape[0] * x.shape[2] * x.shape[3] > 1)


def batch_norm_act(x, bn, relu=True, residual=None):
  """`relu(bn(x) + residual)` (the three forms of the reference's bottleneck unit,
  spml/models/backbones/resnet.py:42-63) through the fused kernels when they apply, through the
  framework ops otherwise (eval mode, NCHW, CPU).  `bn` is the nn.BatchNorm2d / SyncBatchNorm
  module that owns the parameters and running statistics."""
  if fused_bn_act_available(x, bn) and (residual is None or
                                        residual.is_contiguous(memory_format=torch.channels_last)):
    group = None
    if isinstance(bn, torch.nn.SyncBatchNorm):
      import torch.distributed as dist
      if dist.is_available() and dist.is_initialized():
        group = bn.process_group if bn.process_group is not None else dist.group.WORLD
    if bn.num_batches_tracked is not None:
      bn.num_batches_tracked.add_(1)
    momentum = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
    y = _BatchNormAct.apply(x, residual, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                            momentum, bn.eps, bool(relu), group)
    # the kernels update the running statistics through raw pointers: bump the version counters
    # (the eval-mode fold of `mc_bottleneck._fold` keys its cache on them)
    torch.autograd.graph.increment_version((bn.running_mean, bn.running_var))
    return y
  from spml_amd.nn.batchnorm import native_batch_norm
  y = native_batch_norm(bn, x)          # never the library's batch-norm kernels (spml_amd/nn/batchnorm.py)
  if residual is not None:
    y = y + residual
  return torch.relu(y) if relu else y


# ---------------------------------------------------------------------------
class _UpsampleCrossEntropy(torch.autograd.Function):
  """mean over the counted pixels of CE(bilinear_upsample(logits), labels) without the full-resolution
  logits (spml/models/predictions/segsort_softmax.py:112-131)."""

  @staticmethod
  def forward(ctx, logits, labels, ignore_index):
    nhwc = logits.permute(0, 2, 3, 1).contiguous()       # (a view for channels-last logits)
    labels = _i64c(labels)
    result, lse = _ffi.upsample_ce_fwd(nhwc, labels, ignore_index)
    ctx.save_for_backward(nhwc, labels, lse, result)
    ctx.ignore_index = ignore_index
    return result[2]

  @staticmethod
  def backward(ctx, d_loss):
    nhwc, labels, lse, result = ctx.saved_tensors
    scale = (d_loss.to(torch.float32) / result[1]).reshape(1).contiguous()
    d = _ffi.upsample_ce_bwd(nhwc, labels, lse, ctx.ignore_index, scale)
    return d.permute(0, 3, 1, 2), None, None


def upsample_cross_entropy_available(logits, labels):
  return (logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 4 and labels.dim() == 3 and
          labels.shape[0] == logits.shape[0] and _ffi.upsample_ce_supported(logits.shape[1]))


def upsample_cross_entropy(logits, labels, ignore_index):
  """CrossEntropyLoss(ignore_index)(F.interpolate(logits, labels.shape[-2:], mode='bilinear'), labels);
  labels must lie in [0, C) or equal ignore_index."""
  return _UpsampleCrossEntropy.apply(logits, labels, int(ignore_index))


# ---------------------------------------------------------------------------
# Deterministic mode: bilinear up-sampling whose BACKWARD has a fixed summation order
# ---------------------------------------------------------------------------
_interp_cache = {}


def _interp_matrix(n_in, n_out, device, dtype=torch.float32):
  """[n_out, n_in] matrix of F.interpolate(mode='bilinear', align_corners=False) along one axis (its response to
  unit impulses: the same arithmetic as the operator itself)."""
  key = (n_in, n_out, str(device), dtype)
  m = _interp_cache.get(key)
  if m is None:
    eye = torch.eye(n_in, device=device, dtype=dtype).view(1, 1, n_in, n_in)
    m = F.interpolate(eye, size=(n_out, n_in), mode='bilinear')[0, 0].contiguous()
    _interp_cache[key] = m
  return m


class _UpsampleBilinearDetBwd(torch.autograd.Function):
  """F.interpolate(x, size, mode='bilinear') with the input gradient as two matrix products (the operator is
  separable and linear: dX = Wh^T dY Ww).  The framework's backward scatters every output-gradient pixel into its
  four sources with fp32 atomics -- run to run the bits of dX differ; a GEMM has a fixed order."""

  @staticmethod
  def forward(ctx, x, size):
    ctx.in_hw = tuple(x.shape[-2:])
    ctx.size = tuple(size)
    return F.interpolate(x, size=size, mode='bilinear')

  @staticmethod
  def backward(ctx, g):
    wh = _interp_matrix(ctx.in_hw[0], ctx.size[0], g.device, g.dtype)       # [Ho, Hi]
    ww = _interp_matrix(ctx.in_hw[1], ctx.size[1], g.device, g.dtype)       # [Wo, Wi]
    gx = torch.matmul(wh.t(), torch.matmul(g.contiguous(), ww))    # [N, C, Hi, Wi]
    return gx, None


def upsample_bilinear(x, size=None, scale_factor=None):
  """`F.interpolate(x, size=size | scale_factor=scale_factor, mode='bilinear')`; in the library's deterministic mode
  (on GPU tensors that need a gradient) the backward is the fixed-order form above."""
  if size is None:
    size = (int(x.shape[-2] * scale_factor), int(x.shape[-1] * scale_factor))
  if x.is_cuda and x.requires_grad and _ffi.deterministic():
    return _UpsampleBilinearDetBwd.apply(x, tuple(size))
  return F.interpolate(x, size=size, mode='bilinear')
