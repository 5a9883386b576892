"""Differentiable operators over the C-ABI of libspml_hip.so.

Every function here takes GPU tensors and launches hand-written gfx950 kernels
through `spml_amd._ffi`; there is no CPU path (a CPU tensor or a missing library
raises `SpmlHipError`).  `torch.autograd.Function` is used only to connect the
forward and backward kernels."""
import torch

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


def normalize_concat_loc(emb_nchw, loc=None, row_map=None, num_rows=None):
  """K1: NCHW map -> (unit rows [P',C], unit rows with location [P',C+2])."""
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
