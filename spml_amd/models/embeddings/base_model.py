"""Checkpoint loading with the torchvision -> local name mapping
(`spml/models/embeddings/base_model.py`)."""
import warnings

import torch.nn as nn
from torch.nn.parameter import Parameter


class ResnetBase(nn.Module):

  def name_mapping(self, name, resume=False):
    if resume:
      return name[len('module.'):] if name.startswith('module.') else name
    if name.startswith('conv1') or name.startswith('bn1'):
      return 'resnet_backbone.conv1.' + name
    for src, dst in (('layer1', 'res2'), ('layer2', 'res3'), ('layer3', 'res4'), ('layer4', 'res5')):
      name = name.replace(src, 'resnet_backbone.%s.layers' % dst)
    return name

  def load_state_dict(self, state_dict, resume=False):
    """Copies what matches and WARNS (never raises) about unexpected, missing
    or mis-shaped entries (base_model.py:26-52)."""
    own = self.state_dict()
    seen = set()
    for name, param in state_dict.items():
      name = self.name_mapping(name, resume)
      seen.add(name)
      if name not in own:
        warnings.warn('unexpected key "{}" in state_dict'.format(name))
        continue
      if isinstance(param, Parameter):
        param = param.data
      if own[name].shape == param.shape:
        own[name].copy_(param)
      else:
        warnings.warn('While copying the parameter named {}, whose dimensions in the models are'
                      ' {} and whose dimensions in the checkpoint are {}, ...'.format(
                          name, own[name].size(), param.size()))
    missing = set(own.keys()) - seen
    if missing:
      warnings.warn('missing keys in state_dict: "{}"'.format(missing))

  def get_params_lr(self):
    raise NotImplementedError()
