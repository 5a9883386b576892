"""PSPNet embedding network of the DensePose recipe (counterpart of
`spml/models/embeddings/resnet_pspnet_densepose.py`).

Local features are location + smoothed, normalised colour (5 channels: the K1 kernel
takes them as such, the k-means runs on C+5 channels), and after the clustering the
embedding-with-local-features is rebuilt from the embedding scaled by 0.1
(resnet_pspnet_densepose.py:128-139) -- that copy, not the k-means input, feeds the
nearest-neighbour label propagation of the predictor."""
import torch

import spml_amd.utils.general.common as common_utils
from spml_amd.models.embeddings.local_model import LocationColorNetwork
from spml_amd.models.embeddings.resnet_pspnet import ResnetPspnet

_EMBEDDING_SQUEEZE = 0.1


class ResnetPspnetDensepose(ResnetPspnet):

  def __init__(self, backbone_depth, strides, dilations, config):
    super().__init__(backbone_depth, strides, dilations, config)
    self.lfn = LocationColorNetwork(use_color=True, use_location=True, norm_color=True,
                                    smooth_ksize=5)

  def generate_clusters(self, embeddings, semantic_labels, instance_labels, local_features=None):
    out = super().generate_clusters(embeddings, semantic_labels, instance_labels, local_features)
    if local_features is not None:
      local = local_features.reshape(-1, local_features.shape[-1])
      if semantic_labels is not None:            # the rows segment_by_kmeans kept
        keep = (semantic_labels != self.semantic_ignore_index).reshape(-1).nonzero().view(-1)
        local = local[keep]
      squeezed = torch.cat([out['cluster_embedding'] * _EMBEDDING_SQUEEZE, local], dim=-1)
      out['cluster_embedding_with_loc'] = common_utils.normalize_embedding(squeezed)
    return out


def resnet_101_pspnet(config):
  return ResnetPspnetDensepose([3, 4, 23, 3], [1, 2, 1, 1], [1, 1, 2, 4], config)


def resnet_50_pspnet(config):
  return ResnetPspnetDensepose([3, 4, 6, 3], [1, 2, 1, 1], [1, 1, 2, 4], config)
