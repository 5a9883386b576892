"""MI355X mirror of `spml/utils/segsort/common.py`.

Same public names, arguments, return values and error behaviour as the
reference; GPU tensors are processed by the gfx950 kernels of libspml_hip.so
(K1 normalise/transposition, fused spherical k-means, segment prototypes).
There is no CPU fallback: float work on a CPU tensor raises SpmlHipError."""
import os

import torch

from spml_amd._cache import BoundedCache

import spml_amd.utils.general.common as common_utils
from spml_amd import _ffi, ops


def _unique_inverse(keys, with_uniq=True, padded=False):
  """`torch.unique(keys, return_inverse=True)` for the label algebra: GPU tensors go through the
  hash-set kernel (`spml_relabel_unique_i64`: no sort of the pixels, one host read of the count when the
  distinct keys themselves are wanted, none otherwise); CPU tensors through torch.  `padded`: the
  distinct keys come in a buffer as long as `keys`, filled up with INT64_MAX (no host read)."""
  if keys.is_cuda and os.environ.get('SPML_NO_RELABEL') != '1':      # (A/B switch: torch.unique on the GPU)
    uniq, inv, _ = _ffi.relabel_unique(keys, with_uniq=with_uniq, padded=padded)
    return uniq, inv.view(keys.shape)
  uniq, inv = torch.unique(keys, return_inverse=True)
  if padded:
    fill = uniq.new_full((keys.numel() - uniq.numel(),), torch.iinfo(torch.int64).max)
    return torch.cat([uniq, fill]), inv
  return (uniq if with_uniq else None), inv


def calculate_prototypes_from_labels(embeddings, labels, max_label=None):
  """Mean direction per label (segsort/common.py:11-41): scatter-sum of the
  rows by label, L2-normalised; a label without pixels gives a zero vector."""
  embeddings = embeddings.reshape(-1, embeddings.shape[-1])
  labels = labels.reshape(-1)
  if max_label is None:
    max_label = int(labels.max()) + 1
  return ops.segment_prototypes(embeddings, labels, int(max_label))


def find_nearest_prototypes(embeddings, prototypes):
  """argmax_k <embedding, prototype_k>, ties to the lowest k (segsort/common.py:44-64)."""
  embeddings = embeddings.reshape(-1, prototypes.shape[-1])
  p = embeddings.shape[0]
  off = torch.tensor([0, p], dtype=torch.int64, device=embeddings.device)
  return ops.kmeans_assign(embeddings, off, max(p, 1), prototypes.reshape(1, -1, prototypes.shape[-1]))


def kmeans_with_initial_labels(embeddings, initial_labels, max_label=None, iterations=10):
  """vMF k-means from given labels: `iterations` x (M-step, E-step)
  (segsort/common.py:67-97), one fused HIP pass per iteration."""
  if max_label is None:
    max_label = int(initial_labels.max()) + 1
  p = embeddings.shape[0]
  off = torch.tensor([0, p], dtype=torch.int64, device=embeddings.device)
  return ops.kmeans(embeddings, off, max(p, 1), int(max_label), initial_labels.reshape(-1),
                    iterations)


def initialize_cluster_labels(num_clusters, img_dimensions, device):
  """Uniform grid initialisation (segsort/common.py:129-153)."""
  device = torch.device(device)
  if device.type != 'cuda':
    y = torch.linspace(0, num_clusters[0] - 1, img_dimensions[0], device=device).round_().long()
    x = torch.linspace(0, num_clusters[1] - 1, img_dimensions[1], device=device).round_().long()
    return y.view(-1, 1) + (y.max() + 1) * x.view(1, -1)
  return ops.kmeans_init_grid(img_dimensions[0], img_dimensions[1], num_clusters[0],
                              num_clusters[1], device)


def kmeans(embeddings, num_clusters, iterations=10):
  """k-means over a whole `[B,H,W,C]` batch as ONE point set, grid-initialised
  (segsort/common.py:100-126)."""
  shape = embeddings.shape
  labels = initialize_cluster_labels(num_clusters, [shape[1], shape[2]], embeddings.device)
  labels = labels.view(1, shape[1], shape[2]).expand(shape[0], -1, -1).reshape(-1)
  labels = kmeans_with_initial_labels(embeddings.reshape(-1, shape[3]), labels,
                                      iterations=iterations)
  return labels.view(shape[0], shape[1], shape[2])


def generate_location_features(img_dimensions, device, feature_type='int'):
  """`[H,W,2]` grid, channel 0 = y, channel 1 = x (segsort/common.py:156-189)."""
  if feature_type == 'int':
    y = torch.arange(img_dimensions[0], device=device)
    x = torch.arange(img_dimensions[1], device=device)
  elif feature_type == 'float':
    y = torch.linspace(0, 1, img_dimensions[0], device=device)
    x = torch.linspace(0, 1, img_dimensions[1], device=device)
  else:
    raise ValueError('Type of location features should be either int or float.')
  gy, gx = torch.meshgrid(y, x, indexing='ij')
  return torch.stack([gy, gx], dim=2)


def prepare_prototype_labels(semantic_labels, instance_labels, offset=256):
  """Dense re-index of (instance, semantic) pairs (segsort/common.py:192-218)."""
  panoptic = semantic_labels + instance_labels * offset
  uniq, inverse = _unique_inverse(panoptic)
  return uniq % offset, inverse


def find_majority_label_index(semantic_labels, cluster_labels):
  """Pixels agreeing with their cluster's majority class (segsort/common.py:221-267)."""
  sem = semantic_labels.reshape(-1)
  clu = cluster_labels.reshape(-1)
  n_clu = int(clu.max()) + 1
  n_cls = int(sem.max()) + 1
  hist = torch.zeros((n_clu * n_cls,), dtype=torch.long, device=sem.device)
  hist.index_add_(0, clu * n_cls + sem, torch.ones_like(sem))
  major = torch.argmax(hist.view(n_clu, n_cls), dim=1)
  keep = (major[clu] == sem).nonzero()
  return keep, major


_grid_ids = BoundedCache(8)


def _dense_grid(num_clusters, h, w, dev):
  """Dense ids of the grid initialisation and their number: a constant of (grid, map size) --
  computed once per shape (its `unique` and `max` would otherwise cost two host syncs per call); at most 8 shapes
  are kept (spml_amd/_cache.py)."""
  def make():
    grid = initialize_cluster_labels(num_clusters, (h, w), dev).reshape(-1)
    _, grid = torch.unique(grid, return_inverse=True)
    return (grid, int(grid.max()) + 1)
  return _grid_ids.get_or_make((int(num_clusters[0]), int(num_clusters[1]), int(h), int(w), str(dev)), make)


def _dense_ids_per_image(cluster_indices):
  """`unique(return_inverse)` per image (segsort/common.py:341-344)."""
  n = cluster_indices.shape[0]
  flat = cluster_indices.reshape(n, -1)
  ids, ks = [], []
  for b in range(n):
    _, inv = torch.unique(flat[b], return_inverse=True)
    ids.append(inv)
    ks.append(int(inv.max()) + 1)
  return torch.stack(ids), ks


def segment_by_kmeans(embeddings, labels=None, num_clusters=[5, 5], cluster_indices=None,
                      local_features=None, ignore_index=None, iterations=10, shard_id=None):
  """Per-image spherical k-means over an NCHW embedding map
  (segsort/common.py:270-408).

  Returns `(embeddings [P',C], embeddings_with_loc [P',C+2], labels [P'],
  cluster_indices [P'], batch_indices [P'])`, P' = pixels whose label is not
  `ignore_index`, image-major.  `shard_id` stands for the reference's
  `tensor.device.index` (common.py:376), which there numbers the replicas of ONE process:
  batch ids are `image + N * shard_id`.  Here one process drives one GPU, so the default
  is the torch.distributed rank (unique across nodes and under HIP_VISIBLE_DEVICES, where
  every process sees its GPU as cuda:0), or the GPU ordinal without a process group."""
  n, c, h, w = embeddings.shape
  dev = embeddings.device
  hw = h * w
  if shard_id is None:
    if torch.distributed.is_available() and torch.distributed.is_initialized():
      shard_id = torch.distributed.get_rank()
    else:
      shard_id = dev.index or 0

  if labels is None:
    labels = torch.zeros((n, h, w), dtype=torch.long, device=dev)
  flat_labels = labels.reshape(-1)

  # ---- initial cluster ids (dense per image) ----
  if cluster_indices is None:
    grid, k_grid = _dense_grid(num_clusters, h, w, dev)
    ks = [k_grid] * n
    init = grid.view(1, hw).expand(n, hw)
  else:
    init, ks = _dense_ids_per_image(cluster_indices)

  # ---- ignore-pixel removal folded into the K1 kernel through a row map ----
  if ignore_index is not None:
    keep = flat_labels != ignore_index
    counts = keep.view(n, hw).sum(1)
    row_map = torch.where(keep, torch.cumsum(keep, 0) - 1, torch.full_like(flat_labels, -1))
  else:
    keep = None
    counts = torch.full((n,), hw, dtype=torch.long, device=dev)
    row_map = None
  seg_off = torch.zeros(n + 1, dtype=torch.long, device=dev)
  seg_off[1:] = torch.cumsum(counts, 0)
  seg_host = seg_off.tolist()                    # the one host sync: P' sizes the outputs
  rows = seg_host[-1]
  image_sizes = [seg_host[i + 1] - seg_host[i] for i in range(n)]

  if local_features is not None and local_features.shape[-1] > 8:
    # more local channels than the K1 kernel takes: normalise, concatenate, normalise
    emb_rows, _ = ops.normalize_concat_loc(embeddings, None, row_map, rows)
    loc_rows = local_features.reshape(n * hw, -1)
    if keep is not None:
      loc_rows = loc_rows[keep]
    emb_loc_rows = ops.normalize_rows(torch.cat([emb_rows, loc_rows.float()], -1))
  else:
    # (y, x) location, or location + colours of the DensePose recipe (5 channels)
    loc = None
    if local_features is not None:
      loc = local_features.expand(n, h, w, local_features.shape[-1]).float().contiguous()
    emb_rows, emb_loc_rows = ops.normalize_concat_loc(embeddings, loc, row_map, rows)

  if keep is None:
    kept_labels, kept_init = flat_labels, init.reshape(-1)
  else:
    # compaction through the row map (the number of kept rows is already known: no second sync)
    dst = torch.where(keep, row_map, torch.full_like(row_map, rows))
    both = torch.stack([flat_labels, init.reshape(-1)])
    packed = both.new_empty((2, rows + 1)).scatter_(1, dst.view(1, -1).expand(2, -1), both)
    kept_labels, kept_init = packed[0, :rows], packed[1, :rows]

  # ---- k-means: one ragged launch when every image has the same K ----
  if rows == 0:
    clu = kept_init
  elif len(set(ks)) == 1:
    clu = ops.kmeans(emb_loc_rows, seg_off, hw, ks[0], kept_init, iterations)
  else:
    # per image: different K per image
    parts = []
    for b in range(n):
      lo, hi = seg_host[b], seg_host[b + 1]
      if hi == lo:
        parts.append(kept_init[lo:hi])
        continue
      off = torch.tensor([0, hi - lo], dtype=torch.long, device=dev)
      parts.append(ops.kmeans(emb_loc_rows[lo:hi], off, hw, ks[b], kept_init[lo:hi], iterations))
    clu = torch.cat(parts)

  batch = torch.repeat_interleave(
      torch.arange(n, device=dev, dtype=torch.long) + n * shard_id, counts, output_size=rows)
  # pixels kept per image, already on the host: rides on the returned tensor so that the predictors need
  # not count them again on the device (`cluster_image_sizes` of the embedding models' outputs)
  batch._spml_image_sizes = image_sizes

  # ---- label algebra (common.py:398-405): unique(batch * div + cluster) -> dense ids, then
  # prepare_prototype_labels(labels, ids, labels.max() + 1) = unique(labels + ids * offset).  The two
  # dense re-indexings compose into ONE over (batch, cluster, label) in lexicographic order: the
  # key (batch * div + cluster) * offset + label sorts exactly like the reference's second key ----
  if rows == 0:
    return emb_rows, emb_loc_rows, kept_labels, clu, batch
  div = clu.max() + 1
  offset = kept_labels.max() + 1
  _, clu = _unique_inverse((batch * div + clu) * offset + kept_labels, with_uniq=False)
  return emb_rows, emb_loc_rows, kept_labels, clu, batch
