"""MI355X mirror of `spml/utils/segsort/eval.py` (top-k retrieval)."""
import torch

import spml_amd.utils.general.common as common_utils
from spml_amd import ops


def top_k_ranking(embeddings, labels, prototypes, prototype_labels, top_k=3):
  """Top-k prototypes by cosine affinity and the label hit rate (eval.py:9-52);
  fused MFMA + register top-k instead of a full argsort of the `[Q,M]` matrix."""
  embeddings = embeddings.reshape(-1, embeddings.shape[-1])
  prototypes = prototypes.reshape(-1, prototypes.shape[-1])
  if top_k > prototypes.shape[0]:
    raise IndexError('top_k larger than the number of prototypes')
  idx, _ = ops.topk_affinity(embeddings, prototypes, top_k)
  top_k_labels = prototype_labels.reshape(-1)[idx.reshape(-1)].view(-1, top_k)
  accuracy = (top_k_labels == labels.reshape(-1, 1)).float().mean()
  return accuracy, top_k_labels


def majority_label_from_topk(top_k_labels, num_classes=None):
  """Most frequent label among the k retrieved (eval.py:55-70)."""
  votes = common_utils.one_hot(top_k_labels, num_classes).sum(dim=1)
  return torch.argmax(votes, 1)
