"""MI355X mirror of `spml/utils/general/common.py` (same names, arguments and
error behaviour).  Float work on GPU tensors runs in libspml_hip.so; the pure
index helpers are device-agnostic torch calls."""
import torch
import torch.nn.functional as F

from spml_amd import ops


def resize_labels(labels, size):
  """Nearest-neighbour resize of `[N,H,W]` long labels (general/common.py:11-26)."""
  n, h, w = labels.shape
  out = F.interpolate(labels.view(n, 1, h, w).float(), size=size, mode='nearest')
  return out.squeeze_(1).long()


def calculate_principal_components(embeddings, num_components=3):
  """First right-singular vectors of the centred rows (general/common.py:29-42)."""
  centred = embeddings - embeddings.mean(0, keepdim=True)
  _, _, v = torch.svd(centred)
  return v[:, :num_components]


def pca(embeddings, num_components=3, principal_components=None):
  """Project the last dim onto principal components (general/common.py:45-73)."""
  shape = embeddings.shape
  flat = embeddings.reshape(-1, shape[-1])
  if principal_components is None:
    principal_components = calculate_principal_components(flat, num_components)
  return (flat @ principal_components).view(*shape[:-1], num_components)


def one_hot(labels, max_label=None):
  """int64 one-hot on a new last axis (general/common.py:76-98)."""
  if max_label is None:
    max_label = labels.max() + 1
  flat = labels.reshape(-1, 1)
  out = torch.zeros((flat.shape[0], int(max_label)), dtype=torch.long, device=labels.device)
  out.scatter_(1, flat, 1)
  return out.view(*labels.shape, int(max_label))


def normalize_embedding(embeddings, eps=1e-12):
  """L2-normalise the last dim, `x / max(|x|, eps)` (general/common.py:101-120).
  HIP kernel (spml_normalize_rows_f32), differentiable."""
  if eps != 1e-12:
    raise ValueError('the gfx950 kernel is built for eps = 1e-12 (the reference default)')
  return ops.normalize_rows(embeddings)


def segment_mean(x, index):
  """tf.segment_mean: per-index mean of rows, empty index -> 0 (general/common.py:123-147)."""
  x = x.reshape(-1, x.shape[-1])
  index = index.reshape(-1)
  m = int(index.max()) + 1
  tot = torch.zeros((m, x.shape[-1]), dtype=torch.float, device=x.device)
  tot.index_add_(0, index, x.float())
  cnt = torch.zeros((m,), dtype=torch.float, device=x.device)
  cnt.index_add_(0, index, torch.ones_like(index, dtype=torch.float))
  cnt = torch.where(cnt == 0, torch.ones_like(cnt), cnt)
  return tot / cnt.view(-1, 1)
