"""SegSort + softmax head for the DensePose recipe (counterpart of
`spml/models/predictions/segsort_softmax_densepose.py`).

Differences to `SegsortSoftmax`:
  * the co-occurrence term does not use image-level tags: every segment inherits the class
    of its most similar LABELLED segment of the same image (cosine >= 0.95 on the
    prototypes with location + colour, top-1; `models/utils.py:157-223`), segments without
    any such neighbour match everything (segsort_softmax_densepose.py:154-172);
  * the memory bank needs no tag entry (:135-152) but its prototypes with location;
  * the per-image term runs on the embedding WITHOUT local features (:212)."""
import torch

import spml_amd.models.utils as model_utils
import spml_amd.utils.segsort.loss as segsort_loss
from spml_amd.models.predictions.segsort_softmax import SegsortSoftmax


class SegsortSoftmaxDensepose(SegsortSoftmax):

  img_sim_embedding_key = 'cluster_embedding'

  def _memory_bank_ready(self, targets):
    return all(targets.get(k, []) for k in (
        'memory_prototype', 'memory_prototype_semantic_label', 'memory_prototype_batch_index'))

  def _occurrence_sets(self, targets, use_memory, clu, bat, p_sem, p_bat):
    protos_loc = targets['prototype_with_loc']
    if use_memory:
      protos_loc = torch.cat([protos_loc] + list(targets.get('memory_prototype_with_loc', [])), 0)
    tags = model_utils.gather_multiset_labels_per_batch_by_nearest_neighbor(
        protos_loc, protos_loc, p_sem, p_bat, p_bat, num_classes=self.num_classes, top_k=1,
        threshold=0.95, label_divisor=self.label_divisor)
    untagged = tags.max(dim=1, keepdim=True)[0] == 0
    tags = tags.masked_fill(untagged.expand(-1, self.num_classes), 1)
    sets, _ = segsort_loss.pack_tag_set_pair(tags, tags)
    return sets[clu], sets


def segsort(config):
  return SegsortSoftmaxDensepose(config)
