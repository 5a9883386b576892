"""Stage-2 softmax classifier on frozen pixel embeddings
(`spml/models/predictions/softmax_classifier.py`; trained by
`pyscripts/train/train_classifier.py`).  SURVEY.md 8(f) row N4."""
import torch
import torch.nn as nn
import torch.nn.functional as F

import spml_amd.models.utils as model_utils


class SoftmaxClassifier(nn.Module):

  def __init__(self, config):
    super().__init__()
    dim = config.network.embedding_dim
    self.semantic_classifier = nn.Sequential(
        nn.Conv2d(dim, dim * 2, kernel_size=3, padding=1, stride=1, bias=False),
        nn.BatchNorm2d(dim * 2),
        nn.ReLU(inplace=True),
        nn.Dropout(p=0.65),
        nn.Conv2d(dim * 2, config.dataset.num_classes, kernel_size=1, stride=1, bias=True))
    self.semantic_loss = nn.CrossEntropyLoss(ignore_index=config.dataset.semantic_ignore_index)
    self.ignore_index = config.dataset.semantic_ignore_index
    self.num_classes = config.dataset.num_classes

  def forward(self, datas, targets=None):
    """`datas['embedding']` [N,C,H,W] -> logits / prediction, cross-entropy and pixel
    accuracy against `targets['semantic_label']` when given (softmax_classifier.py:36-93)."""
    targets = targets if targets is not None else {}
    emb = datas['embedding']
    emb = emb / torch.norm(emb, dim=1, keepdim=True)
    logits = self.semantic_classifier(emb)
    loss, acc = None, None
    labels = targets.get('semantic_label', None)
    if labels is not None:
      logits = F.interpolate(logits, size=labels.shape[-2:], mode='bilinear')
      pred = torch.argmax(logits, dim=1)
      labels = labels.masked_fill(labels >= self.num_classes, self.ignore_index)
      labels = labels.squeeze(1).long()
      loss = self.semantic_loss(logits, labels)
      valid = torch.ne(labels, self.ignore_index)
      acc = torch.masked_select(torch.eq(pred, labels), valid).float().mean()
    else:
      pred = torch.argmax(logits, dim=1)
    return {'semantic_prediction': pred, 'semantic_logit': logits, 'sem_ann_loss': loss,
            'accuracy': acc}

  def get_params_lr(self):
    """classifier weights x10, biases x20 without decay (softmax_classifier.py:95-111)."""
    return [
        {'params': list(model_utils.get_params(self, ['semantic_classifier'], ['weight'])), 'lr': 10},
        {'params': list(model_utils.get_params(self, ['semantic_classifier'], ['bias'])), 'lr': 20,
         'weight_decay': 0},
    ]


def softmax_classifier(config):
  return SoftmaxClassifier(config)
