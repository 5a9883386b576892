"""Checkpoint loading shared by the embedding networks (counterpart of
`spml/models/embeddings/base_model.py`): torchvision-style ResNet names are translated to
the local module tree, and mismatches only warn -- a partially matching checkpoint (e.g.
ImageNet weights without the segmentation head) must still load."""
import warnings

import torch.nn as nn

_STAGES = {'layer1': 'res2', 'layer2': 'res3', 'layer3': 'res4', 'layer4': 'res5'}


def _strip_module(name):
  prefix = 'module.'
  return name[len(prefix):] if name.startswith(prefix) else name


class ResnetBase(nn.Module):

  def name_mapping(self, name, resume=False):
    """Checkpoint key -> key of this model (base_model.py:11-24).  `resume`: the file was
    written by this code (possibly under DataParallel), only the 'module.' prefix goes."""
    if resume:
      return _strip_module(name)
    if name.startswith(('conv1', 'bn1')):
      return 'resnet_backbone.conv1.' + name
    for src, dst in _STAGES.items():
      name = name.replace(src, 'resnet_backbone.%s.layers' % dst)
    return name

  def load_state_dict(self, state_dict, resume=False):
    """Copy every entry whose translated name and shape match; warn about the rest
    (unexpected, mis-shaped, missing) instead of raising (base_model.py:26-52)."""
    target = self.state_dict()
    visited = set()
    for key, value in state_dict.items():
      key = self.name_mapping(key, resume)
      visited.add(key)
      dst = target.get(key)
      if dst is None:
        warnings.warn('unexpected key "{}" in state_dict'.format(key))
      elif dst.shape != value.shape:
        warnings.warn('While copying the parameter named {}, whose dimensions in the models are'
                      ' {} and whose dimensions in the checkpoint are {}, ...'.format(
                          key, dst.size(), value.size()))
      else:
        dst.copy_(getattr(value, 'data', value))
    absent = set(target) - visited
    if absent:
      warnings.warn('missing keys in state_dict: "{}"'.format(absent))

  def get_params_lr(self):
    raise NotImplementedError()
