"""PSPNet pixel-embedding network (`spml/models/embeddings/resnet_pspnet.py`): same
backbone, clustering and forward as the DeepLab variant, pyramid-pooling head
(2048 -> 512 -> embedding_dim) instead of ASPP.  SURVEY.md 8(f) row N4."""
import torch.nn as nn
import torch.nn.functional as F

import spml_amd.models.utils as model_utils
from spml_amd import ops
from spml_amd.models.backbones.resnet import ResnetBackbone
from spml_amd.models.embeddings.base_model import ResnetBase
from spml_amd.models.embeddings.local_model import LocationColorNetwork
from spml_amd.models.embeddings.resnet_deeplab import ResnetDeeplab
from spml_amd.models.heads.spp import PSPP


class ResnetPspnet(ResnetBase):

  def __init__(self, backbone_depth, strides, dilations, config):
    super().__init__()
    self.resnet_backbone = ResnetBackbone(backbone_depth, strides, dilations, config)
    self.pspp = nn.Sequential(PSPP(2048, 512, bn=True, relu=True),
                              nn.Conv2d(512, config.network.embedding_dim, 1, 1, 0, 1, bias=True))
    self.lfn = LocationColorNetwork(use_color=False, use_location=True, norm_color=False,
                                    smooth_ksize=None)
    self.label_divisor = config.network.label_divisor
    self.num_classes = config.dataset.num_classes
    self.semantic_ignore_index = config.dataset.semantic_ignore_index
    self.kmeans_num_clusters = config.network.kmeans_num_clusters
    self.kmeans_iterations = config.network.kmeans_iterations
    self.initialize()

  def generate_embeddings(self, datas, targets=None, resize_as_input=False):
    """image -> {'embedding', 'local_feature'} (resnet_pspnet.py:56-88)."""
    _, _, _, res5 = self.resnet_backbone(datas['image'])
    emb = ops.upsample_bilinear(self.pspp(res5), scale_factor=2)
    if resize_as_input:
      emb = F.interpolate(emb, size=datas['image'].shape[-2:], mode='bilinear')
    local = self.lfn(datas['image'], size=emb.shape[-2:])
    return {'embedding': emb, 'local_feature': local}

  # per-image spherical k-means and the forward pass are the DeepLab variant's
  # (resnet_pspnet.py:90-186 repeats resnet_deeplab.py:90-180 verbatim)
  generate_clusters = ResnetDeeplab.generate_clusters
  forward = ResnetDeeplab.forward

  def initialize(self):
    pass

  def get_params_lr(self):
    """LR groups (resnet_pspnet.py:188-220): res3-5 weights x1 / biases x2 (no decay),
    pspp weights x10 / biases x20; conv1 and res2 are in no group (frozen)."""
    stages = ['resnet_backbone.res3', 'resnet_backbone.res4', 'resnet_backbone.res5']
    groups = []
    for prefixes, w_lr, b_lr in ((stages, 1, 2), (['pspp'], 10, 20)):
      groups.append({'params': list(model_utils.get_params(self, prefixes, ['weight'])),
                     'lr': w_lr})
      groups.append({'params': list(model_utils.get_params(self, prefixes, ['bias'])),
                     'lr': b_lr, 'weight_decay': 0})
    return groups

  def name_mapping(self, name, resume=False):
    if resume:
      return name[len('module.'):] if name.startswith('module.') else name
    if name.startswith('conv1') or name.startswith('bn1'):
      return 'resnet_backbone.conv1.' + name
    for src, dst in (('layer1', 'res2'), ('layer2', 'res3'), ('layer3', 'res4'), ('layer4', 'res5')):
      name = name.replace(src, 'resnet_backbone.' + dst)
    return name


def resnet_101_pspnet(config):
  """PSPNet / ResNet-101 (resnet_pspnet.py:234-237)."""
  return ResnetPspnet([3, 4, 23, 3], [1, 2, 1, 1], [1, 1, 2, 4], config)


def resnet_50_pspnet(config):
  """PSPNet / ResNet-50 (resnet_pspnet.py:240-243)."""
  return ResnetPspnet([3, 4, 6, 3], [1, 2, 1, 1], [1, 1, 2, 4], config)
