"""CPU oracle for the SPML pixel-to-segment contrastive hot path.

TEST INFRASTRUCTURE ONLY.  This file is a CPU restatement of the reference
algorithm (twke18/SPML, ``spml/utils/segsort``, ``spml/utils/general/common.py``,
``spml/models/utils.py`` and the loss assembly of ``spml/models/predictions``).
It is the checker the HIP path is compared with.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; nothing under ``spml_amd/`` does, and the product path raises when
the HIP library is missing instead of falling back to this file.

Parity status: PINNED.  ``tools/gen_golden.py`` imports the reference's own
leaf functions from ``/root/reference`` (in the build container, never copied),
runs them on seeded inputs and stores inputs + outputs under ``tests/golden``;
``tests/test_oracle_golden.py`` checks every function below against those
vectors (indices exact, floats to 1e-6) -- including the restatements of the inference
scripts' arithmetic blocks (prototype.py:134-205 -> n2_window.npz, pseudo_camrw_crf.py:139-164
-> n3_randomwalk.npz: the generator exec's those source lines from the reference tree) and,
through oracle/cpu_step.py, two whole training steps of the reference's model classes and
lib.nn.optimizer.SGD (h01_step*.npz).

Why torch-on-CPU and not numpy/C: the reference *is* a chain of ATen calls
(``mm``, ``scatter_add_``, ``argmax``, ``unique``, ``argsort``, ``topk``,
``round_``); their tie-breaking, ordering and rounding rules are the
specification.  Re-using the same primitives on CPU, in
fp32, keeps the oracle an op-for-op restatement and makes the "reference CPU
path timed beside the GPU" baseline a fair one (same BLAS, same threading).

Every function cites the reference file:line it restates (paths relative to
the reference root).
"""

from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ---------------------------------------------------------------------------
# spml/utils/general/common.py
# ---------------------------------------------------------------------------

def normalize_embedding(x: Tensor, eps: float = 1e-12) -> Tensor:
  """x / max(||x||_2, eps) over the last dim (general/common.py:101-120)."""
  n = x.norm(dim=-1, keepdim=True)
  n = torch.where(n >= eps, n, torch.full_like(n, eps))
  return x / n


def one_hot(labels: Tensor, max_label: Optional[int] = None) -> Tensor:
  """int64 one-hot along a new last axis (general/common.py:76-98)."""
  if max_label is None:
    max_label = int(labels.max()) + 1
  flat = labels.reshape(-1, 1)
  out = torch.zeros((flat.shape[0], int(max_label)), dtype=torch.long)
  out.scatter_(1, flat, 1)
  return out.view(*labels.shape, int(max_label))


def resize_labels(labels: Tensor, size: Sequence[int]) -> Tensor:
  """Nearest-neighbour label resize through float (general/common.py:11-26)."""
  n, h, w = labels.shape
  r = F.interpolate(labels.view(n, 1, h, w).float(), size=tuple(size),
                    mode='nearest')
  return r[:, 0].long()


def segment_mean(x: Tensor, index: Tensor) -> Tensor:
  """Per-index mean of rows, empty index -> 0 (general/common.py:123-147)."""
  x = x.reshape(-1, x.shape[-1])
  index = index.reshape(-1)
  m = int(index.max()) + 1
  tot = torch.zeros((m, x.shape[-1]), dtype=torch.float)
  tot.index_add_(0, index, x.float())
  cnt = torch.zeros((m,), dtype=torch.float)
  cnt.index_add_(0, index, torch.ones(index.shape[0], dtype=torch.float))
  cnt = torch.where(cnt == 0, torch.ones_like(cnt), cnt)
  return tot / cnt.view(-1, 1)


# ---------------------------------------------------------------------------
# spml/utils/segsort/common.py
# ---------------------------------------------------------------------------

def calculate_prototypes_from_labels(emb: Tensor, labels: Tensor,
                                     max_label: Optional[int] = None) -> Tensor:
  """Mean direction per label: row scatter-sum then L2 normalise
  (segsort/common.py:11-41).  Rows are accumulated in pixel order, fp32; a
  label with no pixel yields the zero vector (0 / 1e-12)."""
  emb = emb.reshape(-1, emb.shape[-1])
  labels = labels.reshape(-1)
  if max_label is None:
    max_label = int(labels.max()) + 1
  sums = torch.zeros((int(max_label), emb.shape[-1]), dtype=emb.dtype)
  sums.index_add_(0, labels, emb)
  return normalize_embedding(sums)


def find_nearest_prototypes(emb: Tensor, protos: Tensor) -> Tensor:
  """argmax_k <emb, proto_k>, ties -> lowest k (segsort/common.py:44-64)."""
  emb = emb.reshape(-1, protos.shape[-1])
  return torch.argmax(emb @ protos.t(), dim=1)


def kmeans_with_initial_labels(emb: Tensor, initial_labels: Tensor,
                               max_label: Optional[int] = None,
                               iterations: int = 10,
                               trace: Optional[list] = None) -> Tensor:
  """Spherical (vMF) k-means: `iterations` x (M-step, E-step), starting from
  given labels (segsort/common.py:67-97).  If ``trace`` is a list, a dict per
  iteration with the prototypes, labels and top-2 margin is appended."""
  if max_label is None:
    max_label = int(initial_labels.max()) + 1
  labels = initial_labels
  for _ in range(iterations):
    protos = calculate_prototypes_from_labels(emb, labels, max_label)
    sims = emb @ protos.t()
    labels = torch.argmax(sims, dim=1)
    if trace is not None:
      if sims.shape[1] >= 2:
        top2 = torch.topk(sims, 2, dim=1).values
        margin = top2[:, 0] - top2[:, 1]
      else:
        margin = torch.full((sims.shape[0],), float('inf'))
      trace.append({'prototypes': protos, 'labels': labels, 'margin': margin})
  return labels


def initialize_cluster_labels(num_clusters: Sequence[int],
                              img_dimensions: Sequence[int]) -> Tensor:
  """Uniform grid init, x-major numbering, round-half-even of a linspace
  (segsort/common.py:129-153)."""
  ky, kx = int(num_clusters[0]), int(num_clusters[1])
  h, w = int(img_dimensions[0]), int(img_dimensions[1])
  y = torch.linspace(0, ky - 1, h).round().long().view(-1, 1)
  x = torch.linspace(0, kx - 1, w).round().long().view(1, -1)
  return y + (y.max() + 1) * x


def generate_location_features(img_dimensions: Sequence[int],
                               feature_type: str = 'int') -> Tensor:
  """[H, W, 2] grid, channel 0 = y, channel 1 = x (segsort/common.py:156-189)."""
  h, w = int(img_dimensions[0]), int(img_dimensions[1])
  if feature_type == 'int':
    yy, xx = torch.arange(h), torch.arange(w)
  elif feature_type == 'float':
    yy, xx = torch.linspace(0, 1, h), torch.linspace(0, 1, w)
  else:
    raise ValueError('Type of location features should be either int or float.')
  gy, gx = torch.meshgrid(yy, xx, indexing='ij')
  return torch.stack([gy, gx], dim=2)


def prepare_prototype_labels(semantic_labels: Tensor, instance_labels: Tensor,
                             offset: int = 256) -> Tuple[Tensor, Tensor]:
  """Dense re-index of (instance, semantic) pairs through a sorted unique
  (segsort/common.py:192-218)."""
  pan = semantic_labels + instance_labels * offset
  uniq, inv = torch.unique(pan, return_inverse=True)
  return uniq % offset, inv


def find_majority_label_index(semantic_labels: Tensor, cluster_labels: Tensor
                              ) -> Tuple[Tensor, Tensor]:
  """Pixels that agree with their cluster's majority class
  (segsort/common.py:221-267)."""
  sem = semantic_labels.reshape(-1)
  clu = cluster_labels.reshape(-1)
  n_clu = int(clu.max()) + 1
  n_cls = int(sem.max()) + 1
  hist = torch.zeros((n_clu, n_cls), dtype=torch.long)
  hist.index_put_((clu, sem), torch.ones_like(sem), accumulate=True)
  major = torch.argmax(hist, dim=1)
  keep = (major[clu] == sem).nonzero()
  return keep, major


def segment_by_kmeans(embeddings: Tensor,
                      labels: Optional[Tensor] = None,
                      num_clusters: Sequence[int] = (5, 5),
                      cluster_indices: Optional[Tensor] = None,
                      local_features: Optional[Tensor] = None,
                      ignore_index: Optional[int] = None,
                      iterations: int = 10,
                      gpu_id: int = 0,
                      trace: Optional[list] = None):
  """Per-image spherical k-means over an NCHW embedding map
  (segsort/common.py:270-408).

  ``gpu_id`` stands for ``tensor.device.index`` at common.py:376 (``None`` on
  CPU in the reference, which makes it raise there; rank r passes r).
  Returns (embeddings [P',C], embeddings_with_loc [P',C+2], labels [P'],
  cluster_indices [P'], batch_indices [P'])."""
  emb = embeddings.permute(0, 2, 3, 1).contiguous()
  n, h, w, c = emb.shape
  emb = normalize_embedding(emb)

  if local_features is None:
    loc = generate_location_features((h, w), 'float') - 0.5
    local_features = loc.view(1, h, w, 2).expand(n, h, w, 2)
  if cluster_indices is None:
    cluster_indices = initialize_cluster_labels(num_clusters, (h, w))
    cluster_indices = cluster_indices.view(1, h, w).expand(n, h, w)
  if labels is None:
    labels = torch.zeros((n, h, w), dtype=torch.long)

  out_lab, out_clu, out_bat, out_emb, out_loc = [], [], [], [], []
  for b in range(n):
    lab_b = labels[b].reshape(-1)
    _, clu_b = torch.unique(cluster_indices[b].reshape(-1), return_inverse=True)
    k_b = int(clu_b.max()) + 1
    emb_b = emb[b].reshape(-1, c)
    loc_b = local_features[b].reshape(-1, local_features.shape[-1])
    embloc_b = normalize_embedding(torch.cat([emb_b, loc_b], dim=-1))

    if ignore_index is not None:
      keep = (lab_b != ignore_index).nonzero().view(-1)
      lab_b, clu_b = lab_b[keep], clu_b[keep]
      emb_b, embloc_b = emb_b[keep], embloc_b[keep]

    if emb_b.shape[0] > 0:
      img_trace = [] if trace is not None else None
      clu_b = kmeans_with_initial_labels(embloc_b, clu_b, k_b, iterations,
                                         trace=img_trace)
      if trace is not None:
        trace.append(img_trace)

    out_lab.append(lab_b)
    out_clu.append(clu_b)
    out_bat.append(torch.full_like(clu_b, b + n * gpu_id))
    out_emb.append(emb_b)
    out_loc.append(embloc_b)

  lab = torch.cat(out_lab)
  clu = torch.cat(out_clu)
  bat = torch.cat(out_bat)
  div = clu.max() + 1
  _, clu = torch.unique(bat * div + clu, return_inverse=True)
  _, clu = prepare_prototype_labels(lab, clu, int(lab.max()) + 1)
  return torch.cat(out_emb), torch.cat(out_loc), lab, clu, bat


# ---------------------------------------------------------------------------
# spml/utils/segsort/loss.py
# ---------------------------------------------------------------------------

def _nca_from_masks(sim: Tensor, own: Tensor, pos_mask: Tensor, neg_mask: Tensor
                    ) -> Tensor:
  """Shared tail of loss.py:56-82 / :106-130 ('segsort+' mode).

  sim [P,M] = exp(kappa * cos); own [P] = index of the pixel's own segment.
  pos = sum_{mask} sim - sim[p, own]  (that order: sum first, subtract after),
  replaced by sim[p, own] when pos <= 0; den = sum_{neg} sim + pos."""
  self_sim = sim.gather(1, own.view(-1, 1))
  pos = (sim * pos_mask.float()).sum(1, keepdim=True) - self_sim
  num = torch.where(pos > 0, pos, self_sim)
  den = (sim * neg_mask.float()).sum(1, keepdim=True) + num
  return -(num / den).log()


def segsort_nll(emb: Tensor, sem: Tensor, own: Tensor, protos: Tensor,
                proto_sem: Tensor, concentration: float) -> Tensor:
  """Per-pixel NCA negative log-likelihood, integer labels
  (loss.py:15-82, group_mode 'segsort+').  Returns [P, 1]."""
  emb = emb.reshape(-1, emb.shape[-1])
  protos = protos.reshape(-1, protos.shape[-1])
  sim = ((emb @ protos.t()) * concentration).exp()
  same = sem.view(-1, 1) == proto_sem.view(1, -1)
  return _nca_from_masks(sim, own, same, ~same)


def set_segsort_nll(emb: Tensor, tags: Tensor, own: Tensor, protos: Tensor,
                    proto_tags: Tensor, concentration: float) -> Tensor:
  """Per-pixel NCA NLL with multi-hot tag sets: positives share >= 1 tag
  (loss.py:85-130).  Returns [P, 1]."""
  emb = emb.reshape(-1, emb.shape[-1])
  protos = protos.reshape(-1, protos.shape[-1])
  sim = ((emb @ protos.t()) * concentration).exp()
  aff = tags.float() @ proto_tags.t().float()
  return _nca_from_masks(sim, own, aff > 0, aff == 0)


def _reduce(x: Tensor, reduction: str) -> Tensor:
  if reduction == 'mean':
    return x.mean()
  if reduction == 'sum':
    return x.sum()
  return x


def segsort_loss(emb, sem, own, protos, proto_sem, concentration,
                 reduction: str = 'mean') -> Tensor:
  """SegSortLoss.forward (loss.py:149-190)."""
  return _reduce(segsort_nll(emb, sem, own, protos, proto_sem, concentration),
                 reduction)


def set_segsort_loss(emb, tags, own, protos, proto_tags, concentration,
                     reduction: str = 'mean') -> Tensor:
  """SetSegSortLoss.forward (loss.py:209-251)."""
  return _reduce(set_segsort_nll(emb, tags, own, protos, proto_tags,
                                 concentration), reduction)


# ---------------------------------------------------------------------------
# spml/utils/segsort/eval.py
# ---------------------------------------------------------------------------

def top_k_ranking(emb: Tensor, labels: Tensor, protos: Tensor,
                  proto_labels: Tensor, top_k: int = 3) -> Tuple[Tensor, Tensor]:
  """Top-k retrieval by cosine affinity (eval.py:9-52).  The reference takes
  the first k columns of a full descending argsort."""
  emb = emb.reshape(-1, emb.shape[-1])
  protos = protos.reshape(-1, protos.shape[-1])
  order = torch.argsort(emb @ protos.t(), dim=1, descending=True)[:, :top_k]
  hit = labels.view(-1, 1) == proto_labels.view(1, -1)
  acc = hit.gather(1, order).float().mean()
  return acc, proto_labels.view(-1)[order.reshape(-1)].view(-1, top_k)


def majority_label_from_topk(top_k_labels: Tensor,
                             num_classes: Optional[int] = None) -> Tensor:
  """Most frequent label among the k retrieved (eval.py:55-70)."""
  votes = one_hot(top_k_labels, num_classes).sum(dim=1)
  return torch.argmax(votes, dim=1)


def segsort_predictions(cluster_embedding: Tensor, cluster_index: Tensor,
                        memory_prototypes: Tensor, memory_labels: Tensor,
                        top_k: int = 20) -> Tuple[Tensor, Tensor]:
  """Semantic prediction by nearest-neighbour retrieval (`Segsort.predictions`,
  spml/models/predictions/segsort.py:68-125): prototypes of the dense-reindexed
  clusters, top-20 memory prototypes per segment (queried in min(10, N-1) groups),
  majority vote, scattered back to the pixels."""
  _, clu = torch.unique(cluster_index, return_inverse=True)
  n = int(clu.max()) + 1
  protos = calculate_prototypes_from_labels(cluster_embedding, clu, n)
  dummy = torch.zeros(n, dtype=torch.long)
  pred = torch.zeros((n,), dtype=torch.long)
  topk = torch.zeros((n, top_k), dtype=torch.long)
  groups = min(10, n - 1)
  r = n // groups
  bounds = [i * r for i in range(groups)] + [n]
  for i in range(groups):
    st, ed = bounds[i], bounds[i + 1]
    _, labs = top_k_ranking(protos[st:ed], dummy[st:ed], memory_prototypes, memory_labels, top_k)
    pred[st:ed] = majority_label_from_topk(labs)
    topk[st:ed] = labs
  return torch.gather(pred, 0, clu), torch.index_select(topk, 0, clu)


def load_memory_banks(memory_dir: str) -> Tuple[Tensor, Tensor]:
  """Concatenate the per-image prototype files of a memory-bank directory
  (spml/utils/segsort/others.py:11-41; files written by
  pyscripts/inference/prototype.py:207-211 as a pickled dict with keys
  'prototype' [M,C] float32 and 'prototype_label' [M] int64), sorted by name."""
  import glob
  import os
  import numpy as np
  paths = sorted(glob.glob(os.path.join(memory_dir, '*.npy')))
  assert len(paths) > 0, 'No memory stored in the directory'
  protos, labels = [], []
  for path in paths:
    d = np.load(path, allow_pickle=True).item()
    protos.append(d['prototype'])
    labels.append(d['prototype_label'])
  return (torch.FloatTensor(np.concatenate(protos, 0)),
          torch.LongTensor(np.concatenate(labels, 0)))


# ---------------------------------------------------------------------------
# DensePose recipe (N4): spml/models/embeddings/resnet_pspnet_densepose.py and
# spml/models/predictions/segsort_softmax_densepose.py
# ---------------------------------------------------------------------------

def densepose_generate_clusters(embeddings: Tensor, semantic_labels: Tensor,
                                instance_labels: Tensor, local_features: Tensor,
                                num_clusters: Sequence[int], label_divisor: int,
                                semantic_ignore_index: int = 255, iterations: int = 10):
  """`ResnetPspnet.generate_clusters` of the DensePose variant
  (resnet_pspnet_densepose.py:90-160): k-means on embedding + 5 local channels, then the
  embedding-with-local-features is rebuilt from `0.1 * embedding` (:128-139)."""
  labels = semantic_labels * label_divisor + instance_labels
  ignore_index = labels.max() + 1
  labels = labels.masked_fill(semantic_labels == semantic_ignore_index, ignore_index)
  emb, _, lab, clu, bat = segment_by_kmeans(
      embeddings, labels, num_clusters, local_features=local_features,
      ignore_index=ignore_index, iterations=iterations)
  valid = (semantic_labels != semantic_ignore_index).view(-1).nonzero().view(-1)
  local = local_features.reshape(-1, local_features.shape[-1])[valid]
  emb_loc = normalize_embedding(torch.cat([emb * 0.1, local], dim=-1))
  return {'cluster_embedding': emb, 'cluster_embedding_with_loc': emb_loc,
          'cluster_semantic_label': lab // label_divisor,
          'cluster_instance_label': lab % label_divisor,
          'cluster_index': clu, 'cluster_batch_index': bat}


def densepose_propagated_tags(prototypes_with_loc: Tensor, proto_sem: Tensor, proto_bat: Tensor,
                              num_classes: int, label_divisor: int) -> Tensor:
  """Class tags every segment inherits from its nearest labelled segment of the same
  image (top-1, cosine >= 0.95); segments without one get every tag
  (segsort_softmax_densepose.py:154-167)."""
  tags = gather_multiset_labels_per_batch_by_nearest_neighbor(
      prototypes_with_loc, prototypes_with_loc, proto_sem, proto_bat, proto_bat,
      num_classes=num_classes, top_k=1, threshold=0.95, label_divisor=label_divisor)
  untagged = torch.max(tags, dim=1, keepdim=True)[0] == 0
  return tags.masked_fill(untagged.expand(-1, num_classes), 1)


def densepose_losses(datas: Dict[str, Tensor], targets: Dict[str, object], num_classes: int,
                     label_divisor: int, sem_ann: Tuple[float, float], sem_occ: Tuple[float, float],
                     img_sim: Tuple[float, float], softmax_ce: Tensor):
  """`SegsortSoftmax.losses` of segsort_softmax_densepose.py:101-240 given the
  cross-entropy of its classifier head (`softmax_ce`, :108-121).  Each of sem_ann / sem_occ
  / img_sim is (concentration, weight)."""
  clu = datas['cluster_index']
  emb = datas['cluster_embedding']
  sem = datas['cluster_semantic_label']
  protos = targets['prototype']
  protos_loc = targets['prototype_with_loc']
  p_sem = targets['prototype_semantic_label']
  p_bat = targets['prototype_batch_index']
  mem_p = targets.get('memory_prototype', [])
  mem_pl = targets.get('memory_prototype_with_loc', [])
  mem_sem = targets.get('memory_prototype_semantic_label', [])
  mem_bat = targets.get('memory_prototype_batch_index', [])
  if mem_p and mem_sem and mem_bat:                               # :135-152
    protos = torch.cat([protos] + list(mem_p))
    protos_loc = torch.cat([protos_loc] + list(mem_pl))
    p_sem = torch.cat([p_sem] + list(mem_sem))
    p_bat = torch.cat([p_bat] + list(mem_bat))
  p_tags = densepose_propagated_tags(protos_loc, p_sem, p_bat, num_classes, label_divisor)
  tags = p_tags[clu]
  px = (sem < num_classes).nonzero().view(-1)
  pr = (p_sem < num_classes).nonzero().view(-1)
  ids = torch.arange(protos.shape[0])
  ids = ids.masked_fill(p_sem >= num_classes, int(ids.max()) + 1)
  _, ids = torch.unique(ids, return_inverse=True)
  new_clu = ids[clu]
  l_ann = softmax_ce + segsort_loss(emb[px], sem[px], new_clu[px], protos[pr], p_sem[pr], sem_ann[0])
  l_ann = l_ann * sem_ann[1]
  l_occ = None                      # sem_occ_loss_types 'none' (the shipped point recipe): no term
  if sem_occ is not None:
    l_occ = set_segsort_loss(emb, tags, clu, protos, p_tags, sem_occ[0]) * sem_occ[1]
  acc, _ = top_k_ranking(protos, p_sem, protos, p_sem, 5)
  ins, bat = datas['cluster_instance_label'], datas['cluster_batch_index']
  terms = []
  for b in torch.unique(bat):                                     # :206-232, no location
    sel = (bat == b).nonzero().view(-1)
    e, lab, c = emb[sel], ins[sel], clu[sel]
    p_lab, c = prepare_prototype_labels(lab, c, int(lab.max()) + 1)
    terms.append(segsort_loss(e, lab, c, calculate_prototypes_from_labels(e, c), p_lab, img_sim[0]))
  l_img = sum(terms) / len(terms) * img_sim[1]
  return l_ann, l_occ, l_img, acc


# ---------------------------------------------------------------------------
# pyscripts/inference/prototype.py (N2: full-resolution embedding + k-means)
# ---------------------------------------------------------------------------

def sliding_window_ends(pad_size: int, crop_size: int, stride: int):
  """End coordinates of the crops along one axis (prototype.py:134-142):
  ceil((pad - crop) / stride) + 1 windows, ends = linspace(crop, pad, n) cast to int32."""
  import math
  import numpy as np
  n = math.ceil(1.0 * (pad_size - crop_size) / stride) + 1
  return np.linspace(crop_size, pad_size, n, dtype=np.int32)


def full_resolution_embedding(embed_fn, image: Tensor, crop_size: Sequence[int],
                              stride: Sequence[int]) -> Tensor:
  """Sliding-window embedding of a padded image `[1,3,Hp,Wp]` (prototype.py:134-181):
  every crop is embedded (`embed_fn(crop) -> [1,C,h,w]`, i.e.
  `generate_embeddings(..., resize_as_input=True)['embedding']`), L2-normalised over the
  channels, summed into the full map; the sum is divided by the per-pixel crop count."""
  pad_h, pad_w = image.shape[-2:]
  crop_h, crop_w = crop_size
  ends_h = sliding_window_ends(pad_h, crop_h, stride[0])
  ends_w = sliding_window_ends(pad_w, crop_w, stride[1])
  acc = None
  counts = torch.zeros(1, 1, pad_h, pad_w)
  with torch.no_grad():
    for eh in ends_h:
      for ew in ends_w:
        sh, sw = int(eh) - crop_h, int(ew) - crop_w
        emb = embed_fn(image[:, :, sh:int(eh), sw:int(ew)])
        emb = normalize_embedding(emb.permute(0, 2, 3, 1).contiguous()).permute(0, 3, 1, 2)
        if acc is None:
          acc = torch.zeros(1, emb.shape[1], pad_h, pad_w)
        acc[:, :, sh:int(eh), sw:int(ew)] += emb
        counts[:, :, sh:int(eh), sw:int(ew)] += 1
  return acc / counts


def full_resolution_prototypes(embed_fn, image: Tensor, semantic_label: Tensor,
                               crop_size: Sequence[int], stride: Sequence[int],
                               num_clusters: Sequence[int], label_divisor: int,
                               semantic_ignore_index: int = 255, iterations: int = 10):
  """One image of the memory-bank pass (prototype.py:107-211).  `image` is the padded
  input `[1,3,Hp,Wp]`, `semantic_label` `[h,w]` the labels of the un-padded region
  (top-left).  Returns (prototypes [M,C], prototype majority labels [M],
  cluster index map [h,w])."""
  h, w = semantic_label.shape[-2:]
  pad_h, pad_w = image.shape[-2:]
  fake = torch.full((1, pad_h, pad_w), semantic_ignore_index, dtype=torch.long)
  fake[:, :h, :w] = 0                                    # prototype.py:117-131
  emb = full_resolution_embedding(embed_fn, image, crop_size, stride)
  labels = fake * label_divisor + fake                   # resnet_deeplab.py:104-117
  ignore_index = labels.max() + 1
  labels = labels.masked_fill(fake == semantic_ignore_index, ignore_index)
  cl_emb, _, _, cl_idx, _ = segment_by_kmeans(
      emb, labels, num_clusters, ignore_index=ignore_index, iterations=iterations)
  protos = calculate_prototypes_from_labels(cl_emb, cl_idx)
  _, proto_labels = find_majority_label_index(semantic_label, cl_idx)
  return protos, proto_labels, cl_idx.view(h, w)


def predict_full_resolution(embed_fn, image: Tensor, valid_hw: Sequence[int], crop_size: Sequence[int],
                            stride: Sequence[int], num_clusters: Sequence[int], label_divisor: int,
                            memory_prototypes: Tensor, memory_labels: Tensor,
                            semantic_ignore_index: int = 255, iterations: int = 10):
  """One image of the kNN label inference (pyscripts/inference/inference.py:145-237): sliding-window
  embedding of the padded `image` [1,3,Hp,Wp] (:162-210), k-means over the whole image with the
  padding (outside the top-left `valid_hw` region) ignored (:145-156, :212-220), nearest-neighbour
  retrieval of every segment's prototype in the memory bank + majority vote (:223-227,
  `Segsort.predictions`) -> (label map [h,w], top-k labels [h*w,20], cluster index [h*w])."""
  h, w = valid_hw
  pad_h, pad_w = image.shape[-2:]
  fake = torch.full((1, pad_h, pad_w), semantic_ignore_index, dtype=torch.long)
  fake[:, :h, :w] = 0
  emb = full_resolution_embedding(embed_fn, image, crop_size, stride)
  labels = fake * label_divisor + fake                   # resnet_deeplab.py:104-117
  ignore_index = labels.max() + 1
  labels = labels.masked_fill(fake == semantic_ignore_index, ignore_index)
  cl_emb, _, _, cl_idx, _ = segment_by_kmeans(
      emb, labels, num_clusters, ignore_index=ignore_index, iterations=iterations)
  pred, topk = segsort_predictions(cl_emb, cl_idx, memory_prototypes, memory_labels)
  return pred.view(h, w), topk, cl_idx


def affinity_random_walk(embs_list: List[Tensor], cam: Tensor, walk_steps: int = 6,
                         return_transition: bool = False):
  """Random walk of class activation maps over the pixel affinity
  (pyscripts/inference/pseudo_camrw_crf.py:143-164, WALK_STEPS = 6; the same lines are in
  pseudo_softmaxrw_crf.py:135-170).  `embs_list`: one `[1,C,h,w]` embedding per augmented
  view (already resized to 1/8 resolution), `cam` `[K,h,w]`."""
  affs = []
  for embs in embs_list:
    embs = embs / torch.norm(embs, dim=1)
    flat = embs.view(embs.shape[1], -1)
    affs.append(torch.matmul(flat.t(), flat).mul_(5).add_(-5).exp_())
  aff = torch.mean(torch.stack(affs, dim=0), dim=0)
  aff_mat = aff ** 20
  trans = aff_mat / torch.sum(aff_mat, dim=0, keepdim=True)
  first = trans
  for _ in range(walk_steps):
    trans = torch.matmul(trans, trans)
  out = torch.matmul(cam.reshape(cam.shape[0], -1), trans).view(cam.shape)
  return (out, first) if return_transition else out


# ---------------------------------------------------------------------------
# spml/models/utils.py
# ---------------------------------------------------------------------------

def gather_clustering_and_update_prototypes(
    embeddings: List[Tensor], embeddings_with_loc: List[Tensor],
    cluster_indices: List[Tensor], batch_indices: List[Tensor],
    semantic_labels: List[Tensor], instance_labels: List[Tensor]):
  """Concatenate every shard's clustering, re-index segments globally, split
  them by (batch, semantic, instance) label and compute the prototypes
  (models/utils.py:41-131).  Lists are per device in the reference; here per
  shard.  Returns the same six lists (prototype tensors replicated)."""
  sections = [int(c.shape[0]) for c in cluster_indices]
  emb = torch.cat(embeddings)
  emb_loc = torch.cat(embeddings_with_loc)
  clu = torch.cat(cluster_indices)
  bat = torch.cat(batch_indices)
  sem = torch.cat(semantic_labels)
  ins = torch.cat(instance_labels)

  div = clu.max() + 1
  _, clu = torch.unique(bat * div + clu, return_inverse=True)

  lab_div = max(int(ins.max()) + 1, int(sem.max()) + 1)
  lab = bat * lab_div ** 2 + sem * lab_div + ins
  proto_lab, new_clu = prepare_prototype_labels(lab, clu, int(lab.max()) + 1)
  proto_bat = proto_lab // lab_div ** 2
  proto_sem = (proto_lab % lab_div ** 2) // lab_div
  proto_ins = proto_lab % lab_div

  protos = calculate_prototypes_from_labels(emb, new_clu)
  protos_loc = calculate_prototypes_from_labels(emb_loc, new_clu)

  n = len(sections)
  return ([protos] * n, [protos_loc] * n, [proto_sem] * n, [proto_ins] * n,
          [proto_bat] * n, list(torch.split(new_clu, sections)))


def gather_multiset_labels_per_batch_by_nearest_neighbor(
    emb: Tensor, protos: Tensor, proto_sem: Tensor, emb_batch: Tensor,
    proto_batch: Tensor, num_classes: int = 21, top_k: int = 3,
    threshold: float = 0.95, label_divisor: int = 255) -> Tensor:
  """Tag propagation: for each query, the classes of its top-k most similar
  labelled prototypes of the same image, kept when cos >= threshold
  (models/utils.py:157-223).  Returns multi-hot [Q, num_classes] int64."""
  emb = emb.reshape(-1, emb.shape[-1])
  protos = protos.reshape(-1, emb.shape[-1])
  q = emb.shape[0]
  ok = (emb_batch.view(-1, 1) == proto_batch.view(1, -1)) & \
       (proto_sem < num_classes).view(1, -1)
  d = emb @ protos.t()
  d = torch.where(ok, d, d.min() - 1)
  nn_d, nn_i = torch.topk(d, top_k, dim=1)
  lab = proto_sem.view(1, -1).expand(q, -1).gather(1, nn_i)
  lab = lab.masked_fill(nn_d < threshold, num_classes)
  hot = one_hot(lab, num_classes + 1).sum(dim=1)
  return (hot > 0).long()[:, :num_classes]


# ---------------------------------------------------------------------------
# Loss assembly of spml/models/predictions/segsort.py (the three contrastive
# terms + retrieval accuracy; the softmax variant adds a conv head on top).
# ---------------------------------------------------------------------------

def segsort_losses(datas: Dict[str, Tensor], targets: Dict[str, object],
                   num_classes: int,
                   sem_ann: Optional[Tuple[float, float]] = None,
                   sem_occ: Optional[Tuple[float, float]] = None,
                   img_sim: Optional[Tuple[float, float]] = None):
  """Segsort.losses (segsort.py:127-243).  Each of ``sem_ann``/``sem_occ``/
  ``img_sim`` is (concentration, weight) or None.  Returns
  (sem_ann_loss, sem_occ_loss, img_sim_loss, accuracy)."""
  l_ann = l_occ = l_img = acc = None

  if sem_ann is not None or sem_occ is not None:
    clu = datas['cluster_index']
    emb = datas['cluster_embedding']
    sem = datas['cluster_semantic_label']
    bat = datas['cluster_batch_index']
    protos = targets['prototype']
    p_sem = targets['prototype_semantic_label']
    p_bat = targets['prototype_batch_index']

    # image tags, background column dropped (segsort.py:147-151)
    tags = targets['semantic_tag'][:, 1:num_classes][bat]
    p_tags = targets['prototype_semantic_tag'][:, 1:num_classes]

    mem_p = targets.get('memory_prototype', [])
    mem_sem = targets.get('memory_prototype_semantic_label', [])
    mem_bat = targets.get('memory_prototype_batch_index', [])
    mem_tag = targets.get('memory_prototype_semantic_tag', [])
    if mem_p and mem_sem and mem_tag and mem_bat:   # segsort.py:162-183
      protos = torch.cat([protos] + list(mem_p))
      p_sem = torch.cat([p_sem] + list(mem_sem))
      p_tags = torch.cat([p_tags] + [t[:, 1:num_classes] for t in mem_tag])
      p_bat = torch.cat([p_bat] + list(mem_bat))

    # labelled pixels / prototypes and the index remap (segsort.py:185-195)
    px = (sem < num_classes).nonzero().view(-1)
    pr = (p_sem < num_classes).nonzero().view(-1)
    ids = torch.arange(protos.shape[0])
    ids = ids.masked_fill(p_sem >= num_classes, int(ids.max()) + 1)
    _, ids = torch.unique(ids, return_inverse=True)
    new_clu = ids[clu]

    l_ann = segsort_loss(emb[px], sem[px], new_clu[px], protos[pr], p_sem[pr],
                         sem_ann[0]) * sem_ann[1]
    l_occ = set_segsort_loss(emb, tags, clu, protos, p_tags,
                             sem_occ[0]) * sem_occ[1]
    acc, _ = top_k_ranking(protos, p_sem, protos, p_sem, 5)

  if img_sim is not None:                           # segsort.py:221-241
    clu = datas['cluster_index']
    emb = datas['cluster_embedding_with_loc']
    ins = datas['cluster_instance_label']
    bat = datas['cluster_batch_index']
    terms = []
    for b in torch.unique(bat):
      sel = (bat == b).nonzero().view(-1)
      e, lab, c = emb[sel], ins[sel], clu[sel]
      p_lab, c = prepare_prototype_labels(lab, c, int(lab.max()) + 1)
      pr = calculate_prototypes_from_labels(e, c)
      terms.append(segsort_loss(e, lab, c, pr, p_lab, img_sim[0]))
    l_img = sum(terms) / len(terms) * img_sim[1]

  return l_ann, l_occ, l_img, acc
