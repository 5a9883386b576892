"""MI355X mirror of `spml/models/utils.py`.

The reference gathers every GPU's pixels onto one anchor GPU, recomputes the
prototypes there and copies them back (models/utils.py:41-131).  In the
one-process-per-GPU layout each rank computes the prototypes of its OWN images
(a segment never spans images, an image never spans ranks) and only the
prototypes are exchanged -- see `spml_amd.parallel`.  The functions below keep
the reference's list-based signatures: a list holds one entry per shard living
in THIS process (normally one)."""
import torch

import spml_amd.utils.general.common as common_utils
import spml_amd.utils.segsort.common as segsort_common
from spml_amd import ops


def get_params(model, prefixs, suffixes, exclude=None):
  """Trainable parameters of the named sub-modules whose name starts/ends with
  one of `suffixes` (models/utils.py:12-38)."""
  for name, module in model.named_modules():
    if name not in prefixs:
      continue
    for n, p in module.named_parameters():
      n = '.'.join([name, n])
      if isinstance(exclude, list) and n in exclude:
        continue
      if isinstance(exclude, str) and exclude in n:
        continue
      for suffix in suffixes:
        if (n.split('.')[-1].startswith(suffix) or n.endswith(suffix)) and p.requires_grad:
          yield p


def local_prototypes(embeddings, embeddings_with_loc, cluster_indices, batch_indices,
                     semantic_labels, instance_labels):
  """Segments -> prototypes for one set of pixels (the body of
  models/utils.py:94-116).  Returns (prototypes, prototypes_with_loc,
  proto_semantic, proto_instance, proto_batch, updated_cluster_indices)."""
  # unique(batch * divisor + cluster) followed by prepare_prototype_labels(lab, ids, lab.max() + 1)
  # (models/utils.py:94-111) as ONE dense re-indexing of (batch, cluster, lab): see segment_by_kmeans
  divisor = cluster_indices.max() + 1
  clu = batch_indices * divisor + cluster_indices
  lab_div = torch.maximum(instance_labels.max() + 1, semantic_labels.max() + 1)
  lab = batch_indices * lab_div ** 2 + semantic_labels * lab_div + instance_labels
  proto_lab, new_clu = segsort_common.prepare_prototype_labels(lab, clu, lab.max() + 1)
  proto_bat = proto_lab // lab_div ** 2
  proto_sem = (proto_lab % lab_div ** 2) // lab_div
  proto_ins = proto_lab % lab_div

  m = proto_lab.shape[0]
  protos = ops.segment_prototypes(embeddings, new_clu, m)
  protos_loc = ops.segment_prototypes(embeddings_with_loc, new_clu, m)
  return protos, protos_loc, proto_sem, proto_ins, proto_bat, new_clu


def gather_clustering_and_update_prototypes(embeddings, embeddings_with_loc, cluster_indices,
                                            batch_indices, semantic_labels, instance_labels,
                                            anchor_device=None):
  """List-in / list-out mirror of models/utils.py:41-131 for the shards of this
  process: concatenates them, re-indexes the segments, splits them by
  (batch, semantic, instance) and computes both prototype sets."""
  sections = [int(c.shape[0]) for c in cluster_indices]
  out = local_prototypes(torch.cat(list(embeddings)), torch.cat(list(embeddings_with_loc)),
                         torch.cat(list(cluster_indices)), torch.cat(list(batch_indices)),
                         torch.cat(list(semantic_labels)), torch.cat(list(instance_labels)))
  n = len(sections)
  return ([out[0]] * n, [out[1]] * n, [out[2]] * n, [out[3]] * n, [out[4]] * n,
          list(torch.split(out[5], sections)))


def gather_and_update_datas(datas, anchor_device=None):
  """Concatenate per-shard tensors and hand every shard the result
  (models/utils.py:134-154)."""
  merged = torch.cat(list(datas), 0)
  return [merged for _ in datas]


def gather_multiset_labels_per_batch_by_nearest_neighbor(
    embeddings, prototypes, semantic_prototype_labels, batch_embedding_labels,
    batch_prototype_labels, num_classes=21, top_k=3, threshold=0.95, label_divisor=255):
  """Tag propagation by nearest labelled segments of the same image
  (models/utils.py:157-223): multi-hot `[num_pixels, num_classes]`."""
  embeddings = embeddings.reshape(-1, embeddings.shape[-1])
  prototypes = prototypes.reshape(-1, embeddings.shape[-1])
  n = embeddings.shape[0]
  valid = semantic_prototype_labels < num_classes
  idx, val = ops.topk_affinity(embeddings, prototypes, top_k, batch_embedding_labels,
                               batch_prototype_labels, valid, masked_value=-2.0)
  labs = semantic_prototype_labels.reshape(-1)[idx.reshape(-1)].view(n, top_k)
  labs = labs.masked_fill(val < threshold, num_classes)
  hot = common_utils.one_hot(labs, num_classes + 1).sum(dim=1)
  return (hot > 0).long()[:, :num_classes]
