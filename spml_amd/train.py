"""Training step of the pixel-to-segment contrastive stage: the MI355X
counterpart of the hot loop of `pyscripts/train/train.py:154-309`.

Same order of operations as the reference -- embeddings + per-image k-means,
prototypes, tag exchange, memory bank, three contrastive losses (+ the softmax
head), lr schedule, SGD.step(lr), memory-bank FIFO -- but one process per GPU:
`DistributedDataParallel` all-reduces the backbone gradients over RCCL (the
reference replicates 189 MB of parameters per step with nn.DataParallel),
SyncBatchNorm exchanges BN statistics, and only segment prototypes are
all-gathered (spml_amd.parallel)."""
import os

import torch
import torch.distributed as dist

import spml_amd.models.utils as model_utils
import spml_amd.utils.general.train as train_utils
from spml_amd import parallel
from spml_amd.models.embeddings.resnet_deeplab import resnet_50_deeplab, resnet_101_deeplab
from spml_amd.models.embeddings.resnet_pspnet import resnet_50_pspnet, resnet_101_pspnet
from spml_amd.models.predictions import segsort as segsort_plain
from spml_amd.models.predictions import segsort_softmax
from spml_amd.nn.optimizer import SGD

LOSS_KEYS = ('sem_ann_loss', 'sem_occ_loss', 'img_sim_loss', 'feat_aff_loss')


def build_models(config, softmax_head=True, recipe='voc'):
  """Embedding + prediction model for `config`.  `recipe='densepose'` picks the modules
  `pyscripts/train/train_densepose.py:28-29` imports (colour + location local features,
  nearest-neighbour propagated tags) instead of those of `pyscripts/train/train.py`."""
  backbone = config.network.backbone_types
  if recipe == 'densepose':
    import spml_amd.models.embeddings.resnet_pspnet_densepose as dp_emb
    import spml_amd.models.predictions.segsort_softmax_densepose as dp_pred
    makers = {'panoptic_pspnet_101': dp_emb.resnet_101_pspnet,
              'panoptic_pspnet_50': dp_emb.resnet_50_pspnet}
    if backbone not in makers:
      raise ValueError('Not support ' + str(backbone))
    if config.network.prediction_types != 'segsort':
      raise ValueError('Not support ' + str(config.network.prediction_types))
    return makers[backbone](config), dp_pred.segsort(config)
  if backbone == 'panoptic_deeplab_101':
    embedding_model = resnet_101_deeplab(config)
  elif backbone == 'panoptic_deeplab_50':
    embedding_model = resnet_50_deeplab(config)
  elif backbone == 'panoptic_pspnet_101':
    embedding_model = resnet_101_pspnet(config)
  elif backbone == 'panoptic_pspnet_50':
    embedding_model = resnet_50_pspnet(config)
  else:
    raise ValueError('Not support ' + str(backbone))
  if config.network.prediction_types == 'segsort':
    # pyscripts/train/train.py:31 binds `segsort` to the softmax variant
    prediction_model = (segsort_softmax if softmax_head else segsort_plain).segsort(config)
  else:
    raise ValueError('Not support ' + str(config.network.prediction_types))
  return embedding_model, prediction_model


class Trainer:
  """Holds models, optimizer and memory bank; `step(datas, targets)` runs one
  iteration and returns the scalar outputs."""

  def __init__(self, config, device, softmax_head=True, freeze_unused=True,
               channels_last=False, recipe='voc', models=None):
    self.config = config
    self.device = torch.device(device)
    self.distributed = parallel.is_distributed()
    self.world = dist.get_world_size() if self.distributed else 1
    # models: an already built (embedding, prediction) pair instead of build_models(config)
    emb, pred = models if models is not None else build_models(config, softmax_head, recipe)
    if freeze_unused:
      # conv1 / res2 are in no optimizer group (resnet_deeplab.py:185-220): they are
      # never updated, so their gradients need not be computed or all-reduced
      for name in ('conv1', 'res2'):
        for p in getattr(emb.resnet_backbone, name).parameters():
          p.requires_grad_(False)
    emb, pred = emb.to(self.device), pred.to(self.device)
    if channels_last:                 # NHWC: MIOpen's native layout, no transposes
      emb = emb.to(memory_format=torch.channels_last)
      pred = pred.to(memory_format=torch.channels_last)
    if config.network.use_syncbn and self.distributed:
      emb = torch.nn.SyncBatchNorm.convert_sync_batchnorm(emb)
      pred = torch.nn.SyncBatchNorm.convert_sync_batchnorm(pred)
    self.embedding_model, self.prediction_model = emb, pred
    self.optimizer = SGD(emb.get_params_lr() + pred.get_params_lr(), lr=1,
                         momentum=config.train.momentum,
                         weight_decay=config.train.weight_decay)
    self.emb_fwd, self.pred_fwd = emb, pred
    if self.distributed:
      ids = [self.device.index] if self.device.type == 'cuda' else None
      # buffers are BN statistics: SyncBatchNorm already keeps them identical on every rank,
      # so the per-forward buffer broadcast is skipped; gradients live in the buckets
      # (no extra copy); 64-MB buckets = 3 all-reduces for the 189 MB of gradients, each
      # large enough to run at xGMI ring bandwidth and still overlap with backward
      ddp = dict(device_ids=ids, broadcast_buffers=not config.network.use_syncbn,
                 gradient_as_bucket_view=True, bucket_cap_mb=64)
      self.emb_fwd = torch.nn.parallel.DistributedDataParallel(emb, **ddp)
      if any(p.requires_grad for p in pred.parameters()):
        self.pred_fwd = torch.nn.parallel.DistributedDataParallel(pred, **ddp)
    self.memory_banks = {}
    self.curr_iter = config.train.begin_iteration
    if self.device.type == 'cuda':
      from spml_amd import _ffi
      if _ffi.deterministic():
        # the convolutions that stay on the framework: weight gradients as one GEMM each (spml_amd/nn/conv.py)
        from spml_amd.nn.conv import make_deterministic
        make_deterministic(self.embedding_model)
        make_deterministic(self.prediction_model)
      if _ffi.deterministic() and os.environ.get('SPML_DETERMINISTIC_FRAMEWORK') == '1':
        # SPML_DETERMINISTIC=1 / _ffi.set_deterministic(True) make the library's sums order-independent (and the
        # up-sampling backward takes ops.upsample_bilinear's fixed-order form).  The framework's own switch is NOT set
        # by default: with `cudnn.deterministic` MIOpen's immediate mode falls back to its naive reference convolutions
        # (fp64 accumulation) for the units that stay on the library -- 1.5 s per step instead of 0.12
        # (profiles/r06_determinism.md); SPML_DETERMINISTIC_FRAMEWORK=1 asks for it anyway
        torch.backends.cudnn.deterministic = True
        torch.backends.cudnn.benchmark = False

  # ------------------------------------------------------------------
  def lr(self, it):
    t = self.config.train
    if t.lr_policy == 'step':
      return train_utils.lr_step(t.base_lr, it, t.decay_iterations, t.warmup_iteration)
    return train_utils.lr_poly(t.base_lr, it, t.max_iteration, t.warmup_iteration)

  def forward_losses(self, datas, targets):
    """Embeddings -> prototypes -> losses for this rank's images (train.py:167-220)."""
    targets = dict(targets)
    emb = self.emb_fwd(datas, targets)
    protos = model_utils.local_prototypes(
        emb['cluster_embedding'], emb['cluster_embedding_with_loc'], emb['cluster_index'],
        emb['cluster_batch_index'], emb['cluster_semantic_label'], emb['cluster_instance_label'])
    (targets['prototype'], targets['prototype_with_loc'], targets['prototype_semantic_label'],
     targets['prototype_instance_label'], targets['prototype_batch_index'],
     emb['cluster_index']) = parallel.gather_prototypes(*protos)
    targets['semantic_tag'] = parallel.gather_tags(targets['semantic_tag'])
    targets['prototype_semantic_tag'] = targets['semantic_tag'][targets['prototype_batch_index']]
    for k, bank in self.memory_banks.items():
      assert targets.get(k, None) is None
      targets[k] = list(bank)
    outputs = self.pred_fwd(emb, targets)
    losses = [outputs[k] for k in LOSS_KEYS if outputs.get(k, None) is not None]
    return sum(losses), outputs, targets

  def update_memory(self, targets):
    """FIFO of detached prototype tensors + batch-index shift (train.py:277-293)."""
    size = self.config.train.memory_bank_size
    with torch.no_grad():
      for k, v in targets.items():
        if 'prototype' in k and 'memory' not in k:
          bank = self.memory_banks.setdefault('memory_' + k, [])
          bank.append(v.clone().detach())
          if len(bank) > size:
            del bank[0]
      for mem in self.memory_banks.get('memory_prototype_batch_index', []):
        mem += self.config.train.batch_size * self.world

  def step(self, datas, targets):
    self.embedding_model.train()
    self.prediction_model.train()
    # (grads are dropped before the forward: the loop over the parameters then runs while the GPU is
    # busy with the backbone instead of right after the loss assembly's host syncs, when it is idle)
    self.optimizer.zero_grad()
    loss, outputs, targets = self.forward_losses(datas, targets)
    lr = self.lr(self.curr_iter)
    loss.backward()
    self.optimizer.step(lr)
    self.update_memory(targets)
    self.curr_iter += 1
    out = {k: outputs[k].detach() for k in LOSS_KEYS + ('accuracy',)
           if outputs.get(k, None) is not None}
    out['loss'] = loss.detach()
    out['lr'] = lr
    return out

  def load_state_dict(self, state):
    """Resume: models (through the reference's name mapping, resume=True), optimizer,
    memory bank and iteration counter (the reference's own resume path is broken,
    train.py:114 `fromat`)."""
    self.embedding_model.load_state_dict(state['embedding_model'], resume=True)
    torch.nn.Module.load_state_dict(self.prediction_model, state['prediction_model'])
    if 'optimizer' in state:
      self.optimizer.load_state_dict(state['optimizer'])
    self.memory_banks = {k: [t.to(self.device) for t in v]
                         for k, v in state.get('memory_banks', {}).items()}
    self.curr_iter = state.get('iteration', self.curr_iter)

  def state_dict(self):
    """Same file layout as train.py:296-304 (+ memory bank, iteration)."""
    return {
        'embedding_model': self.embedding_model.state_dict(),
        'prediction_model': self.prediction_model.state_dict(),
        'optimizer': self.optimizer.state_dict(),
        'memory_banks': self.memory_banks,
        'iteration': self.curr_iter,
    }



class ClassifierTrainer:
  """Stage 2: the softmax classifier trained on a FROZEN embedding network -- the hot loop of
  `pyscripts/train/train_classifier.py:33-185`.

  Same order of operations as the reference: the embedding network in eval mode under `no_grad`
  (:110, :140-141), the classifier in train mode (:111), loss = `sem_ann_loss` only (:146-153), lr
  policy (:156-165), `zero_grad / backward / step(lr)` of ONE optimizer that holds the groups of both
  models (:88-93) -- the embedding parameters never receive a gradient, and `lib.nn.optimizer.SGD`
  skips parameters without one, so they stay as loaded (weight decay included).  One process per
  GPU: the classifier is wrapped in DistributedDataParallel (gradient all-reduce over RCCL) and its
  batch norm synchronised as `use_syncbn` says.  The reference forwards the whole embedding model
  (k-means included, :141) and then reads only `embedding` (softmax_classifier.py:52); here only
  `generate_embeddings` runs -- the clustering has no consumer in this stage."""

  def __init__(self, config, device, channels_last=False, models=None):
    from spml_amd.models.predictions.softmax_classifier import softmax_classifier
    self.config = config
    self.device = torch.device(device)
    self.distributed = parallel.is_distributed()
    if models is not None:
      emb, pred = models
    else:
      if config.network.prediction_types != 'softmax_classifier':
        raise ValueError('Not support ' + str(config.network.prediction_types))     # train_classifier.py:86
      makers = {'panoptic_pspnet_101': resnet_101_pspnet, 'panoptic_deeplab_101': resnet_101_deeplab}
      if config.network.backbone_types not in makers:
        raise ValueError('Not support ' + str(config.network.backbone_types))       # :81
      emb, pred = makers[config.network.backbone_types](config), softmax_classifier(config)
    emb, pred = emb.to(self.device), pred.to(self.device)
    if channels_last:
      emb = emb.to(memory_format=torch.channels_last)
      pred = pred.to(memory_format=torch.channels_last)
    if config.network.use_syncbn and self.distributed:
      pred = torch.nn.SyncBatchNorm.convert_sync_batchnorm(pred)
    self.embedding_model, self.prediction_model = emb, pred
    self.optimizer = SGD(emb.get_params_lr() + pred.get_params_lr(), lr=1,
                         momentum=config.train.momentum, weight_decay=config.train.weight_decay)
    if self.device.type == 'cuda':
      from spml_amd import _ffi
      if _ffi.deterministic():
        from spml_amd.nn.conv import make_deterministic
        make_deterministic(pred)
        make_deterministic(emb)        # (frozen here: its 1x1 products and the ASPP forward leave the library)
    self.pred_fwd = pred
    if self.distributed:
      ids = [self.device.index] if self.device.type == 'cuda' else None
      self.pred_fwd = torch.nn.parallel.DistributedDataParallel(
          pred, device_ids=ids, broadcast_buffers=not config.network.use_syncbn)
    self.curr_iter = config.train.begin_iteration

  lr = Trainer.lr

  def load_pretrained(self, path_or_state):
    """`train_classifier.py:97-104`: stage 1's snapshot, embedding model only; required."""
    state = path_or_state
    if isinstance(state, (str, bytes)) or hasattr(state, '__fspath__'):
      state = torch.load(state, map_location=self.device, weights_only=True)
    if 'embedding_model' not in state:
      raise ValueError('Pre-trained model is required.')
    self.embedding_model.load_state_dict(state['embedding_model'], resume=True)

  def step(self, datas, targets):
    self.embedding_model.eval()
    self.prediction_model.train()
    with torch.no_grad():
      embeddings = self.embedding_model.generate_embeddings(datas, targets)
    outputs = self.pred_fwd({'embedding': embeddings['embedding']}, targets)
    loss = outputs['sem_ann_loss']
    lr = self.lr(self.curr_iter)
    self.optimizer.zero_grad()
    loss.backward()
    self.optimizer.step(lr)
    self.curr_iter += 1
    return {'sem_ann_loss': loss.detach(), 'loss': loss.detach(), 'accuracy': outputs['accuracy'].detach(),
            'lr': lr}

  def state_dict(self):
    """The snapshot files of train_classifier.py:172-180 (+ the iteration counter)."""
    return {'embedding_model': self.embedding_model.state_dict(),
            'prediction_model': self.prediction_model.state_dict(),
            'optimizer': self.optimizer.state_dict(), 'iteration': self.curr_iter}

  def load_state_dict(self, state):
    self.embedding_model.load_state_dict(state['embedding_model'], resume=True)
    torch.nn.Module.load_state_dict(self.prediction_model, state['prediction_model'])
    if 'optimizer' in state:
      self.optimizer.load_state_dict(state['optimizer'])
    self.curr_iter = state.get('iteration', self.curr_iter)


def voc12_scribble_config(batch_size=16, crop=513, embedding_dim=64, kmeans=6, num_classes=21,
                          memory_bank_size=2, max_iteration=30000, use_syncbn=True):
  """The recipe of bashscripts/voc12/train_spml_scribble.sh:14-44."""
  from spml_amd.config.default import make_config
  return make_config(
      network=dict(embedding_dim=embedding_dim, label_divisor=2048, use_syncbn=use_syncbn,
                   kmeans_iterations=10, kmeans_num_clusters=[kmeans, kmeans],
                   backbone_types='panoptic_deeplab_101', prediction_types='segsort'),
      dataset=dict(num_classes=num_classes, semantic_ignore_index=255),
      train=dict(lr_policy='poly', max_iteration=max_iteration, warmup_iteration=100,
                 base_lr=3e-3, weight_decay=5e-4, momentum=0.9, batch_size=batch_size,
                 crop_size=[crop, crop], memory_bank_size=memory_bank_size,
                 sem_ann_loss_types='segsort', sem_occ_loss_types='segsort',
                 img_sim_loss_types='segsort', feat_aff_loss_types='none',
                 sem_ann_concentration=6, sem_occ_concentration=12, img_sim_concentration=16,
                 feat_aff_concentration=0, sem_ann_loss_weight=1.0, sem_occ_loss_weight=0.5,
                 img_sim_loss_weight=0.1, feat_aff_loss_weight=0.0))


def voc12_tag_config(batch_size=16, crop=513, **kw):
  """The image-tag recipe of bashscripts/voc12/train_spml_tag.sh:14-44 (BASELINE config 3):
  the scribble recipe with sem_occ concentration 8 and loss weights 0.3 / 0.3 / 0.1; its
  supervision is CAM-like blobs (`synth.make_batch(supervision='tag')`)."""
  cfg = voc12_scribble_config(batch_size=batch_size, crop=crop, **kw)
  cfg.train.sem_occ_concentration = 8
  cfg.train.sem_ann_loss_weight = 0.3
  cfg.train.sem_occ_loss_weight = 0.3
  return cfg


def stress_config(batch_size=2, crop=1025, embedding_dim=512, kmeans=32, **kw):
  """BASELINE config 5 (stress / roofline run): the scribble recipe on a 1025 crop (258x258
  embedding map) with a 512-d embedding and 32x32 = 1024 k-means centroids per image --
  the many-cluster k-means kernels (kmeans_big.hip) and the wide NLL kernels (D = 512 / 514)."""
  return voc12_scribble_config(batch_size=batch_size, crop=crop, embedding_dim=embedding_dim,
                               kmeans=kmeans, **kw)


def densepose_point_config(batch_size=8, crop=769, embedding_dim=32, kmeans=12, num_classes=15,
                           memory_bank_size=0, max_iteration=45000, use_syncbn=True):
  """The recipe of bashscripts/densepose/train_spml_point.sh:14-44 (BASELINE config 4):
  PSPNet-101, 32-d embedding, 12x12 clusters, sem_occ off, feat_aff on, no memory bank.
  Use with `Trainer(..., recipe='densepose')`."""
  from spml_amd.config.default import make_config
  return make_config(
      network=dict(embedding_dim=embedding_dim, label_divisor=2048, use_syncbn=use_syncbn,
                   kmeans_iterations=10, kmeans_num_clusters=[kmeans, kmeans],
                   backbone_types='panoptic_pspnet_101', prediction_types='segsort'),
      dataset=dict(num_classes=num_classes, semantic_ignore_index=255),
      train=dict(lr_policy='poly', max_iteration=max_iteration, warmup_iteration=100,
                 base_lr=3e-3, weight_decay=5e-4, momentum=0.9, batch_size=batch_size,
                 crop_size=[crop, crop], memory_bank_size=memory_bank_size,
                 sem_ann_loss_types='segsort', sem_occ_loss_types='none',
                 img_sim_loss_types='segsort', feat_aff_loss_types='segsort',
                 sem_ann_concentration=6, sem_occ_concentration=0, img_sim_concentration=16,
                 feat_aff_concentration=12, sem_ann_loss_weight=1.0, sem_occ_loss_weight=0.0,
                 img_sim_loss_weight=0.1, feat_aff_loss_weight=0.5))
