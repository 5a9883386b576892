"""SegSort + softmax head (`spml/models/predictions/segsort_softmax.py`, the
predictor `pyscripts/train/train.py:31` imports): the contrastive terms of
`Segsort` plus a cross-entropy loss of a small conv classifier trained on the
DETACHED, L2-normalised embedding map."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from spml_amd.nn.batchnorm import BatchNorm2d

import spml_amd.models.utils as model_utils
from spml_amd import ops
from spml_amd.models.predictions.segsort import Segsort


class _AddChannelBias(torch.autograd.Function):
  """y + bias[c] with the bias gradient summed in two stages over the [pixels, C] view of the output gradient.
  The framework's own bias gradient of a convolution (a reduction over N, H, W of a channels-last tensor with C = 21)
  picks a 32 x 16-thread launch: 0.70 ms per training step for 5.7 MB."""

  @staticmethod
  def forward(ctx, y, bias):
    return y + bias.view(1, -1, 1, 1)

  @staticmethod
  def backward(ctx, g):
    c = g.shape[1]
    g2 = g.permute(0, 2, 3, 1).reshape(-1, c)            # a view of a channels-last gradient
    db = None
    if ctx.needs_input_grad[1]:
      rows = g2.shape[0]
      inner = next(b for b in range(min(256, rows), 0, -1) if rows % b == 0)      # two stages: ~rows / 256 x C partial sums first
      db = g2.view(rows // inner, inner, c).sum(1).sum(0)
    return g, db


def _conv_bias(conv, x):
  """`conv(x)` (spml/models/predictions/segsort_softmax.py:41-48: the classifier's closing 1x1 convolution)."""
  if conv.bias is None or not x.is_cuda or os.environ.get('SPML_NO_BIAS_TWO_STAGE') == '1' or type(conv) is not nn.Conv2d:
    return conv(x)                         # (a re-classed module -- deterministic mode, spml_amd/nn/conv.py -- keeps its own forward)
  y = F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
  return _AddChannelBias.apply(y, conv.bias)


class SegsortSoftmax(Segsort):

  def __init__(self, config):
    super().__init__(config)
    dim = config.network.embedding_dim
    self.semantic_classifier = nn.Sequential(
        nn.Conv2d(dim, dim * 2, kernel_size=3, padding=1, stride=1, bias=False),
        BatchNorm2d(dim * 2), nn.ReLU(inplace=True), nn.Dropout(p=0.75),
        nn.Conv2d(dim * 2, config.dataset.num_classes, kernel_size=1, stride=1, bias=True))
    self.softmax_loss = nn.CrossEntropyLoss(ignore_index=config.dataset.semantic_ignore_index)

  def _logits(self, embeddings):
    embeddings = embeddings / torch.norm(embeddings, dim=1, keepdim=True)
    cls = self.semantic_classifier
    from spml_amd import mc_bottleneck
    if mc_bottleneck.conv_bn_act_available(cls[0], cls[1], embeddings):
      # wide embeddings (BASELINE config 5: 512 -> 1024, 3x3): convolution + batch norm + ReLU on the
      # matrix-core kernels (20 ms of fp32 library convolution per step there); the 64-d head of the other
      # recipes is below the kernels' channel granularity and stays on the framework ops
      return _conv_bias(cls[4], cls[3](mc_bottleneck.conv_bn_act(cls[0], cls[1], embeddings)))
    return _conv_bias(cls[4], cls[:4](embeddings))

  def predictions(self, datas, targets={}):
    logits = self._logits(datas['embedding'])          # segsort_softmax.py:89-101
    return torch.argmax(logits, dim=1), logits

  def losses(self, datas, targets={}):
    """CE of the classifier head (segsort_softmax.py:112-131) added to the
    semantic-annotation term (:196), then the contrastive terms."""
    logits = self._logits(datas['embedding'].detach())
    labels = targets.get('semantic_label', None)
    labels = labels.masked_fill(labels >= self.num_classes, self.semantic_ignore_index)
    labels = labels.squeeze(1).long() if labels.dim() == 4 else labels.long()
    if (os.environ.get('SPML_NO_FUSED_CE') != '1' and ops.upsample_cross_entropy_available(logits, labels) and
        self.softmax_loss.weight is None and self.softmax_loss.reduction == 'mean' and
        self.softmax_loss.label_smoothing == 0.0):
      # up-sampling, log-softmax and the NLL in one pass over the label map: the [N, C, H, W] logits
      # (354 MB at batch 16, 513 x 513) are never written (labels here are in [0, C) or the ignore index)
      ce = ops.upsample_cross_entropy(logits, labels, self.softmax_loss.ignore_index)
    else:
      logits = F.interpolate(logits, size=labels.shape[-2:], mode='bilinear')
      ce = self.softmax_loss(logits, labels)

    sem_ann, sem_occ, img_sim, acc = self._contrastive_losses(datas, targets)
    if self.sem_ann_loss is not None:
      # reference: sem_ann_loss = CE; sem_ann_loss += segsort; sem_ann_loss *= weight
      w = self.sem_ann_loss_weight
      sem_ann = ce * w + sem_ann
    else:
      sem_ann = ce
    return sem_ann, sem_occ, img_sim, acc

  def forward(self, datas, targets=None, with_loss=True, with_prediction=False):
    targets = targets if targets is not None else {}
    outputs = {}
    if with_prediction:
      pred, logits = self.predictions(datas, targets)
      outputs.update({'semantic_prediction': pred, 'semantic_logit': logits})
    if with_loss:
      sem_ann, sem_occ, img_sim, acc = self.losses(datas, targets)
      outputs.update({'sem_ann_loss': sem_ann, 'sem_occ_loss': sem_occ,
                      'img_sim_loss': img_sim, 'accuracy': acc})
      if self.feat_aff_set_loss is not None:
        outputs['feat_aff_loss'] = self.feature_affinity_loss(datas, targets)
    return outputs

  def get_params_lr(self):
    """classifier weights x10, biases x20 without decay (segsort_softmax.py:270-289)."""
    return [
        {'params': list(model_utils.get_params(self, ['semantic_classifier'], ['weight'])),
         'lr': 10},
        {'params': list(model_utils.get_params(self, ['semantic_classifier'], ['bias'])),
         'lr': 20, 'weight_decay': 0},
    ]


def segsort(config):
  """Parametric prototype predictor."""
  return SegsortSoftmax(config)
