"""CPU restatement of one training step of the reference
(`pyscripts/train/train.py:154-309`) on top of oracle/spml_oracle.py.

TEST INFRASTRUCTURE ONLY (parity checker and the `cpu_baseline` leg of
bench.py).  The convolutional modules are plain torch modules shared with the
product (they are device-agnostic); everything the product does with HIP kernels
-- clustering, prototypes, contrastive losses -- is done here by the oracle."""
import torch
import torch.nn.functional as F

from oracle import spml_oracle as O

LOSS_KEYS = ('sem_ann_loss', 'sem_occ_loss', 'img_sim_loss')


class CpuStep:

  def __init__(self, embedding_model, prediction_model, config, optimizer=None, softmax_head=False,
               recipe='voc'):
    """recipe 'densepose': the modules pyscripts/train/train_densepose.py:28-29 binds (k-means on
    embedding + 5 local channels, tags propagated from the nearest labelled segment)."""
    self.emb, self.pred, self.cfg = embedding_model, prediction_model, config
    self.optimizer = optimizer
    self.softmax_head = softmax_head
    self.recipe = recipe
    self.memory = {}
    # what the last forward_losses / step saw: the segment id of every kept pixel (as
    # segment_by_kmeans returns it) and, after step(), d loss / d embedding map -- the parity
    # tests inject the former into the GPU step and compare the latter
    self.last = {}
    # parity tests: segment ids to use INSTEAD of this run's own k-means result (same role as the
    # injection into the GPU step; lets an fp64 run of the same step share the fp32 run's clustering)
    self.given_cluster_index = None

  def _head_ce(self, emb, targets):
    cfg = self.cfg
    x = emb.detach()
    x = x / torch.norm(x, dim=1, keepdim=True)
    logits = self.pred.semantic_classifier(x)
    slab = targets['semantic_label']
    logits = F.interpolate(logits, size=slab.shape[-2:], mode='bilinear')
    slab = slab.masked_fill(slab >= cfg.dataset.num_classes, cfg.dataset.semantic_ignore_index)
    return F.cross_entropy(logits, slab, ignore_index=cfg.dataset.semantic_ignore_index)

  def _forward_losses_densepose(self, datas, targets):
    cfg, t = self.cfg, self.cfg.train
    out = self.emb.generate_embeddings(datas)
    emb = out['embedding']
    emb.retain_grad()
    size = emb.shape[-2:]
    sem = O.resize_labels(targets['semantic_label'], size)
    ins = O.resize_labels(targets['instance_label'], size)
    div = cfg.network.label_divisor
    c = O.densepose_generate_clusters(emb, sem, ins, out['local_feature'], cfg.network.kmeans_num_clusters,
                                      div, cfg.dataset.semantic_ignore_index, cfg.network.kmeans_iterations)
    if self.given_cluster_index is not None:
      c['cluster_index'] = self.given_cluster_index.clone()
    self.last = {'cluster_index': c['cluster_index'].clone(), 'embedding': emb}
    protos, protos_loc, p_sem, p_ins, p_bat, new_clu = [
        x[0] for x in O.gather_clustering_and_update_prototypes(
            [c['cluster_embedding']], [c['cluster_embedding_with_loc']], [c['cluster_index']],
            [c['cluster_batch_index']], [c['cluster_semantic_label']], [c['cluster_instance_label']])]
    tag = targets['semantic_tag']
    tgt = {'prototype': protos, 'prototype_with_loc': protos_loc, 'prototype_semantic_label': p_sem,
           'prototype_instance_label': p_ins, 'prototype_batch_index': p_bat, 'semantic_tag': tag,
           'prototype_semantic_tag': tag[p_bat]}
    full = dict(tgt)
    for k, v in self.memory.items():
      full[k] = list(v)
    c['cluster_index'] = new_clu
    occ = (t.sem_occ_concentration, t.sem_occ_loss_weight) if t.sem_occ_loss_types != 'none' else None
    la, lo, li, acc = O.densepose_losses(
        c, full, cfg.dataset.num_classes, div, (t.sem_ann_concentration, t.sem_ann_loss_weight), occ,
        (t.img_sim_concentration, t.img_sim_loss_weight), self._head_ce(emb, targets))
    outputs = {'sem_ann_loss': la, 'sem_occ_loss': lo, 'img_sim_loss': li, 'accuracy': acc}
    return sum(x for x in (la, lo, li) if x is not None), outputs, tgt

  def forward_losses(self, datas, targets):
    if self.recipe == 'densepose':
      return self._forward_losses_densepose(datas, targets)
    cfg = self.cfg
    out = self.emb.generate_embeddings(datas)
    emb = out['embedding']
    emb.retain_grad()
    size = emb.shape[-2:]
    sem = O.resize_labels(targets['semantic_label'], size)
    ins = O.resize_labels(targets['instance_label'], size)
    div = cfg.network.label_divisor
    labels = sem * div + ins
    ignore = int(labels.max()) + 1
    labels = labels.masked_fill(sem == cfg.dataset.semantic_ignore_index, ignore)
    e, el, lab, clu, bat = O.segment_by_kmeans(
        emb, labels, cfg.network.kmeans_num_clusters, local_features=out['local_feature'],
        ignore_index=ignore, iterations=cfg.network.kmeans_iterations)
    c_sem, c_ins = lab // div, lab % div
    if self.given_cluster_index is not None:
      clu = self.given_cluster_index.clone()
    self.last = {'cluster_index': clu.clone(), 'embedding': emb}
    protos, protos_loc, p_sem, p_ins, p_bat, new_clu = [
        x[0] for x in O.gather_clustering_and_update_prototypes([e], [el], [clu], [bat],
                                                                [c_sem], [c_ins])]
    tag = targets['semantic_tag']
    tgt = {'prototype': protos, 'prototype_with_loc': protos_loc,
           'prototype_semantic_label': p_sem, 'prototype_instance_label': p_ins,
           'prototype_batch_index': p_bat, 'semantic_tag': tag,
           'prototype_semantic_tag': tag[p_bat]}
    full = dict(tgt)
    for k, v in self.memory.items():
      full[k] = list(v)
    datas_c = {'cluster_index': new_clu, 'cluster_embedding': e,
               'cluster_embedding_with_loc': el, 'cluster_semantic_label': c_sem,
               'cluster_instance_label': c_ins, 'cluster_batch_index': bat}
    t = cfg.train
    la, lo, li, acc = O.segsort_losses(
        datas_c, full, cfg.dataset.num_classes,
        (t.sem_ann_concentration, t.sem_ann_loss_weight),
        (t.sem_occ_concentration, t.sem_occ_loss_weight),
        (t.img_sim_concentration, t.img_sim_loss_weight))
    if self.softmax_head:
      la = self._head_ce(emb, targets) * t.sem_ann_loss_weight + la
    outputs = {'sem_ann_loss': la, 'sem_occ_loss': lo, 'img_sim_loss': li, 'accuracy': acc}
    return la + lo + li, outputs, tgt

  def update_memory(self, tgt):
    size = self.cfg.train.memory_bank_size
    with torch.no_grad():
      for k, v in tgt.items():
        if 'prototype' in k:
          bank = self.memory.setdefault('memory_' + k, [])
          bank.append(v.clone().detach())
          if len(bank) > size:
            del bank[0]
      for mem in self.memory.get('memory_prototype_batch_index', []):
        mem += self.cfg.train.batch_size

  def step(self, datas, targets, lr):
    loss, outputs, tgt = self.forward_losses(datas, targets)
    if self.optimizer is not None:
      self.optimizer.zero_grad()
      loss.backward()
      self.last['d_embedding'] = self.last.pop('embedding').grad
      self.optimizer.step(lr)
    self.update_memory(tgt)
    outputs['loss'] = loss.detach()
    return outputs


class CpuClassifierStep:
  """Stage 2 on CPU, plain torch: one iteration of `pyscripts/train/train_classifier.py:139-169` --
  frozen embedding network (eval, no_grad), the softmax classifier's cross-entropy at label
  resolution and pixel accuracy (`spml/models/predictions/softmax_classifier.py:36-93`, restated
  here op for op on the classifier's own layers), zero_grad / backward / SGD.step(lr).  TEST
  INFRASTRUCTURE like the rest of oracle/: pinned to tests/golden/h02_classifier_step.npz."""

  def __init__(self, embedding_model, classifier, config, optimizer):
    self.emb, self.cls, self.cfg, self.optimizer = embedding_model, classifier, config, optimizer

  def forward(self, datas, targets):
    cfg = self.cfg
    with torch.no_grad():
      emb = self.emb.generate_embeddings(datas)['embedding']
    unit = emb / torch.norm(emb, dim=1, keepdim=True)                      # softmax_classifier.py:52-54
    logits = self.cls.semantic_classifier(unit)
    labels = targets['semantic_label']
    logits = F.interpolate(logits, size=labels.shape[-2:], mode='bilinear')   # :61-65
    pred = torch.argmax(logits, dim=1)
    labels = labels.masked_fill(labels >= cfg.dataset.num_classes, cfg.dataset.semantic_ignore_index).long()
    loss = F.cross_entropy(logits, labels, ignore_index=cfg.dataset.semantic_ignore_index)   # :78
    valid = labels != cfg.dataset.semantic_ignore_index
    acc = torch.masked_select(pred == labels, valid).float().mean()        # :79-82
    return loss, acc

  def step(self, datas, targets, lr):
    self.emb.eval()
    self.cls.train()
    loss, acc = self.forward(datas, targets)
    self.optimizer.zero_grad()
    loss.backward()
    self.optimizer.step(lr)
    return {'loss': loss.detach(), 'sem_ann_loss': loss.detach(), 'accuracy': acc.detach()}
