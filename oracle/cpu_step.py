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

  def __init__(self, embedding_model, prediction_model, config, optimizer=None, softmax_head=False):
    self.emb, self.pred, self.cfg = embedding_model, prediction_model, config
    self.optimizer = optimizer
    self.softmax_head = softmax_head
    self.memory = {}

  def forward_losses(self, datas, targets):
    cfg = self.cfg
    out = self.emb.generate_embeddings(datas)
    emb = out['embedding']
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
      x = emb.detach()
      x = x / torch.norm(x, dim=1, keepdim=True)
      logits = self.pred.semantic_classifier(x)
      slab = targets['semantic_label']
      logits = F.interpolate(logits, size=slab.shape[-2:], mode='bilinear')
      slab = slab.masked_fill(slab >= cfg.dataset.num_classes, cfg.dataset.semantic_ignore_index)
      ce = F.cross_entropy(logits, slab, ignore_index=cfg.dataset.semantic_ignore_index)
      la = ce * t.sem_ann_loss_weight + la
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
      self.optimizer.step(lr)
    self.update_memory(tgt)
    outputs['loss'] = loss.detach()
    return outputs
