"""Full-resolution inference pieces of the memory-bank pass
(`pyscripts/inference/prototype.py:107-211`, the same loop is in `inference.py:162-220`):
sliding-window embedding with overlap averaging, k-means over the whole image at full
resolution (the real consumer of the 513x513xC k-means kernel), prototypes + majority
labels, memory-bank files.  SURVEY.md 8(f) rows N1/N2.

The per-crop normalise + accumulate is one HIP kernel (`spml_window_accumulate_f32`),
clustering and prototypes are the same kernels the training path uses."""
import math

import numpy as np
import torch

import spml_amd.utils.segsort.common as segsort_common
import spml_amd.utils.segsort.others as segsort_others
from spml_amd import _ffi


def sliding_window_ends(pad_size, crop_size, stride):
  """End coordinates of the crops along one axis (prototype.py:134-142)."""
  n = math.ceil(1.0 * (pad_size - crop_size) / stride) + 1
  return np.linspace(crop_size, pad_size, n, dtype=np.int32)


def embed_full_resolution(embedding_model, image, crop_size, stride):
  """Sliding-window embedding `[1,C,Hp,Wp]` of a padded image `[1,3,Hp,Wp]`
  (prototype.py:134-181): crops are embedded with `generate_embeddings(...,
  resize_as_input=True)`, normalised over the channels and overlap-averaged."""
  if image.dim() != 4 or image.shape[0] != 1:
    raise ValueError('embed_full_resolution expects one image [1,3,H,W]')
  pad_h, pad_w = image.shape[-2:]
  crop_h, crop_w = crop_size
  ends_h = sliding_window_ends(pad_h, crop_h, stride[0])
  ends_w = sliding_window_ends(pad_w, crop_w, stride[1])
  acc = None
  counts = torch.zeros((pad_h, pad_w), dtype=torch.float32, device=image.device)
  windows = [(int(eh) - crop_h, int(ew) - crop_w) for eh in ends_h for ew in ends_w]
  # the crops of one image go through the network together (eval mode: every sample is independent, the
  # reference's one-crop-at-a-time loop gives the same embeddings) -- in the memory format of the model
  first = next(embedding_model.parameters(), None)
  nhwc = first is not None and first.is_cuda and first.dim() == 4 and \
      first.is_contiguous(memory_format=torch.channels_last) and not first.is_contiguous()
  group = 8
  with torch.no_grad():
    for g0 in range(0, len(windows), group):
      part = windows[g0:g0 + group]
      crops = torch.cat([image[:, :, sh:sh + crop_h, sw:sw + crop_w] for sh, sw in part], 0)
      if nhwc:
        crops = crops.contiguous(memory_format=torch.channels_last)
      embs = embedding_model.generate_embeddings({'image': crops}, resize_as_input=True)['embedding']
      for (sh, sw), emb in zip(part, embs):
        emb = emb.float().contiguous()
        if acc is None:
          acc = torch.zeros((emb.shape[0], pad_h, pad_w), dtype=torch.float32, device=image.device)
        _ffi.window_accumulate(emb, acc, counts, sh, sw)
    acc /= counts
  return acc.unsqueeze(0)


def full_resolution_prototypes(embedding_model, image, semantic_label, crop_size, stride,
                               semantic_ignore_index=255):
  """One image of the memory-bank pass (prototype.py:107-211): padded `image`
  `[1,3,Hp,Wp]`, `semantic_label` `[h,w]` of the un-padded (top-left) region ->
  (prototypes [M,C], majority label per prototype [M], cluster index map [h,w])."""
  h, w = semantic_label.shape[-2:]
  pad_h, pad_w = image.shape[-2:]
  fake = torch.full((1, pad_h, pad_w), semantic_ignore_index, dtype=torch.long,
                    device=image.device)
  fake[:, :h, :w] = 0                 # clustering ignores the padding (prototype.py:117-131)
  embeddings = embed_full_resolution(embedding_model, image, crop_size, stride)
  with torch.no_grad():
    out = embedding_model.generate_clusters(embeddings, fake, fake)
    prototypes = segsort_common.calculate_prototypes_from_labels(
        out['cluster_embedding'], out['cluster_index'])
    _, prototype_labels = segsort_common.find_majority_label_index(
        semantic_label.to(image.device), out['cluster_index'])
  return prototypes, prototype_labels, out['cluster_index'].view(h, w)


def drop_ignored_memory(prototypes, prototype_labels, semantic_ignore_index=255):
  """The memory bank without the prototypes of the ignore class (inference.py:99-111)."""
  keep = torch.nonzero(prototype_labels != semantic_ignore_index).view(-1)
  return prototypes.index_select(0, keep), prototype_labels.index_select(0, keep)


def predict_full_resolution(embedding_model, prediction_model, image, valid_hw, crop_size, stride,
                            memory_prototypes, memory_prototype_labels, semantic_ignore_index=255):
  """One image of the kNN label inference (`pyscripts/inference/inference.py:145-237`), the composition
  of the two rows above: sliding-window embedding of the padded `image` `[1,3,Hp,Wp]` (:162-210),
  k-means at full resolution with the padding outside the top-left `valid_hw` region ignored
  (:145-156, :212-220: the 513x513xC kernel's consumer), then `prediction_model(..., with_loss=False,
  with_prediction=True)` against the memory bank (:223-227: prototypes of the segments, top-20
  retrieval, majority vote, scatter to the pixels).  Returns a dict with `semantic_prediction` `[h,w]`
  (what :231-237 turns into the label image), `semantic_score` (the retrieved labels, `[h*w,20]`)
  and `cluster_index` `[h*w]`."""
  h, w = valid_hw
  pad_h, pad_w = image.shape[-2:]
  fake = torch.full((1, pad_h, pad_w), semantic_ignore_index, dtype=torch.long, device=image.device)
  fake[:, :h, :w] = 0
  embeddings = {'embedding': embed_full_resolution(embedding_model, image, crop_size, stride)}
  with torch.no_grad():
    embeddings.update(embedding_model.generate_clusters(embeddings['embedding'], fake, fake))
    outputs = prediction_model(
        embeddings,
        {'semantic_memory_prototype': memory_prototypes,
         'semantic_memory_prototype_label': memory_prototype_labels},
        with_loss=False, with_prediction=True)
  return {'semantic_prediction': outputs['semantic_prediction'].view(h, w),
          'semantic_score': outputs['semantic_score'], 'cluster_index': embeddings['cluster_index']}


def save_image_memory(path, prototypes, prototype_labels):
  """`np.save` of `{'prototype', 'prototype_label'}` (prototype.py:207-211)."""
  segsort_others.save_memory_bank(path, prototypes, prototype_labels)


def affinity_random_walk(embs_list, cam, walk_steps=6, scale=5.0, power=20):
  """Random walk of class activation maps `[K,h,w]` over the pixel affinity of one image
  (pseudo_camrw_crf.py:143-164).  `embs_list`: one `[1,C,h,w]` embedding
  per augmented view at 1/8 resolution.  The affinity, its mean over the views, the 20th
  power and the column normalisation are one fused HIP kernel; the walk is six fp32
  library GEMMs (rocBLAS through torch.matmul)."""
  with torch.no_grad():
    views = []
    for embs in embs_list:
      embs = embs / torch.norm(embs, dim=1)
      views.append(embs.reshape(embs.shape[1], -1))
    emb = torch.stack(views, 0).float().contiguous()          # [B,C,n]
    trans = _ffi.affinity_transition(emb, scale, power)
    for _ in range(walk_steps):
      trans = torch.matmul(trans, trans)
    return torch.matmul(cam.reshape(cam.shape[0], -1).float(), trans).view(cam.shape)
