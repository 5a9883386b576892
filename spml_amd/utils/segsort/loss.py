"""MI355X mirror of `spml/utils/segsort/loss.py`: SegSort / Set-SegSort
pixel-to-segment contrastive (NCA) losses on the fused gfx950 NLL kernels."""
import torch
from torch.nn.modules.loss import _Loss

from spml_amd import ops


def _mode(group_mode, base, codes32=False):
  return (base | (0 if group_mode == 'segsort+' else ops.NLL_PLAIN) |
          (ops.NLL_CODE32 if codes32 else 0))


def _bit_weights(num, device):
  return torch.bitwise_left_shift(torch.ones((num,), dtype=torch.long, device=device),
                                  torch.arange(num, dtype=torch.long, device=device))


def pack_tag_sets(tags):
  """Multi-hot `[N,T]` tags -> one 64-bit set per row (T <= 63; wider sets: `pack_tag_set_pair`)."""
  if tags.dim() == 1:
    return tags
  if tags.shape[1] > 63:
    raise ValueError('tag sets wider than 63 classes need pack_tag_set_pair (pixel and prototype tags together)')
  return ((tags != 0).long() * _bit_weights(tags.shape[1], tags.device).view(1, -1)).sum(1)


def pack_tag_set_pair(pixel_tags, prototype_tags):
  """Multi-hot `[P,T]` / `[M,T]` tags of any width -> packed 64-bit sets `[P]`, `[M]` with the same
  positive-set predicate as the reference's `mm(pixel_tags, prototype_tags.t()) > 0`
  (segsort/loss.py:95-130, which has no width limit).  Two rows share a tag only through a class that
  occurs on BOTH sides, so only those columns are packed (in column order); more than 63 such classes in
  one call (e.g. a 150-class dataset whose global batch + memory bank shows > 63 classes at once) is not
  supported by the 64-bit kernels and raises."""
  t = pixel_tags.shape[1]
  if t <= 63:
    return pack_tag_sets(pixel_tags), pack_tag_sets(prototype_tags)
  active = (pixel_tags != 0).any(0) & (prototype_tags != 0).any(0)
  rank = torch.cumsum(active.long(), 0) - 1
  n_active = int(rank[-1]) + 1                      # one host sync, wide tag sets only
  if n_active > 63:
    raise ValueError('%d classes are shared between pixels and prototypes in this call; the packed '
                     'tag-set kernels take at most 63' % n_active)
  weights = torch.where(active, torch.bitwise_left_shift(torch.ones_like(rank), rank.clamp(min=0)),
                        torch.zeros_like(rank))
  pack = lambda tags: ((tags != 0).long() * weights.view(1, -1)).sum(1)
  return pack(pixel_tags), pack(prototype_tags)


def _calculate_log_likelihood(embeddings, semantic_labels, instance_labels, prototypes,
                              prototype_semantic_labels, concentration, group_mode,
                              prototype_grad_rows=None, codes32=False):
  """Per-pixel NLL `[P,1]`, integer labels (loss.py:15-82)."""
  embeddings = embeddings.reshape(-1, embeddings.shape[-1])
  prototypes = prototypes.reshape(-1, prototypes.shape[-1])
  nll = ops.segsort_nll(embeddings, instance_labels.reshape(-1), semantic_labels.reshape(-1),
                        prototypes, prototype_semantic_labels.reshape(-1), concentration,
                        _mode(group_mode, ops.NLL_LABEL, codes32), prototype_grad_rows)
  return nll.view(-1, 1)


def _one_hot_calculate_log_likelihood(embeddings, semantic_labels, instance_labels, prototypes,
                                      prototype_semantic_labels, concentration, group_mode,
                                      prototype_grad_rows=None, codes32=False):
  """Per-pixel NLL `[P,1]`, multi-hot tag sets: positives share a tag (loss.py:85-130)."""
  embeddings = embeddings.reshape(-1, embeddings.shape[-1])
  prototypes = prototypes.reshape(-1, prototypes.shape[-1])
  # multi-hot tags over <= 32 classes always pack into 32 bits
  codes32 = codes32 or (semantic_labels.dim() == 2 and semantic_labels.shape[1] <= 32 and
                        prototype_semantic_labels.dim() == 2 and
                        prototype_semantic_labels.shape[1] <= 32)
  if semantic_labels.dim() == 2 and prototype_semantic_labels.dim() == 2:
    px_sets, pr_sets = pack_tag_set_pair(semantic_labels, prototype_semantic_labels)
  else:
    px_sets, pr_sets = pack_tag_sets(semantic_labels), pack_tag_sets(prototype_semantic_labels)
  nll = ops.segsort_nll(embeddings, instance_labels.reshape(-1), px_sets, prototypes, pr_sets, concentration,
                        _mode(group_mode, ops.NLL_TAGSET, codes32), prototype_grad_rows)
  return nll.view(-1, 1)


class _NcaLoss(_Loss):
  _kernel = None
  _name = ''

  def __init__(self, concentration=10, group_mode='segsort+', size_average=None, reduce=None,
               reduction='mean'):
    super().__init__(size_average, reduce, reduction)
    self.concentration = concentration
    self.group_mode = group_mode

  def __repr__(self):
    return '{}(concentration={:.2f}, group_mode={})'.format(
        self._name, self.concentration, self.group_mode)

  def forward(self, embeddings, semantic_labels, instance_labels, prototypes,
              prototype_semantic_labels, prototype_weights=None, prototype_grad_rows=None,
              codes32=False):
    """Extensions: `prototype_grad_rows` -- only the first that many prototypes need a
    gradient (rows of a detached memory bank appended after them are skipped in the backward
    kernel); `codes32` -- the caller promises that every label / packed tag set fits in 32
    bits (class ids, segment ids, sets over <= 32 classes), the kernels then compare 32-bit
    words (same results, ~10 % faster)."""
    nll = type(self)._kernel(embeddings, semantic_labels, instance_labels, prototypes,
                             prototype_semantic_labels, self.concentration, self.group_mode,
                             prototype_grad_rows, codes32)
    if self.reduction == 'mean':
      return torch.mean(nll)
    if self.reduction == 'sum':
      return torch.sum(nll)
    return nll


class SegSortLoss(_NcaLoss):
  """NCA loss with integer semantic labels (loss.py:133-190)."""
  _kernel = staticmethod(_calculate_log_likelihood)
  _name = 'SegSortLoss'


class SetSegSortLoss(_NcaLoss):
  """NCA loss with tag sets (loss.py:193-251); labels may be multi-hot `[.,T]`
  (as in the reference) or already packed 64-bit sets `[.]`."""
  _kernel = staticmethod(_one_hot_calculate_log_likelihood)
  _name = 'SetSegSortLoss'
