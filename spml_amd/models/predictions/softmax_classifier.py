"""Stage-2 softmax classifier on frozen pixel embeddings.

Counterpart of `spml/models/predictions/softmax_classifier.py` (trained by
`pyscripts/train/train_classifier.py`): unit-normalised embedding -> 3x3 conv (2C, no
bias) -> BN -> ReLU -> dropout 0.65 -> 1x1 conv to `num_classes`; cross-entropy and pixel
accuracy at label resolution; labels >= num_classes count as ignored."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from spml_amd.nn.batchnorm import BatchNorm2d

import spml_amd.models.utils as model_utils
from spml_amd import _ffi, ops

_DROPOUT = 0.65


def _head(in_dim, num_classes):
  hidden = 2 * in_dim
  return nn.Sequential(
      nn.Conv2d(in_dim, hidden, 3, stride=1, padding=1, bias=False),
      BatchNorm2d(hidden),
      nn.ReLU(inplace=True),
      nn.Dropout(p=_DROPOUT),
      nn.Conv2d(hidden, num_classes, 1, stride=1, bias=True))


class SoftmaxClassifier(nn.Module):

  def __init__(self, config):
    super().__init__()
    self.num_classes = config.dataset.num_classes
    self.ignore_index = config.dataset.semantic_ignore_index
    self.semantic_classifier = _head(config.network.embedding_dim, self.num_classes)
    self.semantic_loss = nn.CrossEntropyLoss(ignore_index=self.ignore_index)

  def _logits(self, embedding):
    unit = embedding / embedding.norm(dim=1, keepdim=True)
    return self.semantic_classifier(unit)

  def _supervise(self, logits, labels):
    """Cross-entropy + accuracy over the valid pixels, logits upsampled to the labels."""
    logits = ops.upsample_bilinear(logits, size=labels.shape[-2:])       # (deterministic mode: fixed-order backward)
    labels = torch.where(labels >= self.num_classes,
                         torch.full_like(labels, self.ignore_index), labels)
    labels = labels.squeeze(1).long()
    prediction = logits.argmax(dim=1)
    keep = labels != self.ignore_index
    accuracy = (prediction == labels)[keep].float().mean()
    if logits.is_cuda and _ffi.deterministic():
      # the framework's 2-D NLL forward adds the pixels' terms up with atomics: the reported loss (not its gradient)
      # flickers in the last bit run to run; deterministic mode: per-pixel terms, then a plain (fixed-order) sum
      per_pixel = F.cross_entropy(logits, labels, ignore_index=self.ignore_index, reduction='none')
      loss = per_pixel.sum() / keep.sum()
    else:
      loss = self.semantic_loss(logits, labels)
    return logits, prediction, loss, accuracy

  def forward(self, datas, targets=None):
    """softmax_classifier.py:36-93: `datas['embedding']` [N,C,H,W]; optional
    `targets['semantic_label']` [N,H',W']."""
    logits = self._logits(datas['embedding'])
    labels = (targets or {}).get('semantic_label', None)
    if labels is None:
      prediction, loss, accuracy = logits.argmax(dim=1), None, None
    else:
      logits, prediction, loss, accuracy = self._supervise(logits, labels)
    return {'semantic_prediction': prediction, 'semantic_logit': logits,
            'sem_ann_loss': loss, 'accuracy': accuracy}

  def get_params_lr(self):
    """Weights at 10x, biases at 20x without weight decay (softmax_classifier.py:95-111)."""
    pick = lambda suffix: list(model_utils.get_params(self, ['semantic_classifier'], [suffix]))
    return [{'params': pick('weight'), 'lr': 10},
            {'params': pick('bias'), 'lr': 20, 'weight_decay': 0}]


def softmax_classifier(config):
  return SoftmaxClassifier(config)
