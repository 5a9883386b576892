"""Non-parametric SegSort predictor (`spml/models/predictions/segsort.py`):
assembles the pixel-to-segment contrastive losses (semantic annotation,
semantic co-occurrence, low-level image similarity) on the gfx950 NLL kernels
and predicts by nearest-neighbour retrieval."""
import os

import torch
import torch.nn as nn

import spml_amd.utils.segsort.common as segsort_common
import spml_amd.utils.segsort.eval as segsort_eval
from spml_amd import parallel
import spml_amd.utils.segsort.loss as segsort_loss


_streams = {}


def _side_streams(device, n):
  """`n` side streams of `device`, created once (n <= 1: none, everything stays on the current stream)."""
  if n <= 1:
    return []
  key = (torch.device(device).index, n)
  if key not in _streams:
    _streams[key] = [torch.cuda.Stream(device=device) for _ in range(n)]
  return _streams[key]


def _nonzero_pair(mask_a, mask_b, extra=None):
  """`mask_a.nonzero().view(-1), mask_b.nonzero().view(-1)` with one host synchronisation for the two
  sizes instead of one each (compaction through an exclusive scan + scatter).  `extra`: a 1-D integer
  tensor that needs to reach the host anyway rides along with the same read (returned as a list)."""
  head = torch.stack([mask_a.sum(), mask_b.sum()])
  vals = (torch.cat([head, extra.reshape(-1).to(head.dtype)]) if extra is not None else head).tolist()
  counts, tail = vals[:2], vals[2:]
  out = []
  for mask, n in zip((mask_a, mask_b), counts):
    mask = mask.reshape(-1)
    dst = torch.where(mask, torch.cumsum(mask, 0) - 1, torch.full_like(mask, n, dtype=torch.long))
    src = torch.arange(mask.shape[0], device=mask.device)
    out.append(src.new_empty((n + 1,)).scatter_(0, dst, src)[:n])
  return (out[0], out[1], tail) if extra is not None else out


class Segsort(nn.Module):

  def __init__(self, config):
    super().__init__()
    t = config.train
    self.sem_ann_loss = self._construct_loss(t.sem_ann_loss_types,
                                             concentration=t.sem_ann_concentration)
    self.sem_ann_loss_weight = t.sem_ann_loss_weight
    occ_type = 'set_segsort' if t.sem_occ_loss_types == 'segsort' else 'none'
    self.sem_occ_loss = self._construct_loss(occ_type, concentration=t.sem_occ_concentration)
    self.sem_occ_loss_weight = t.sem_occ_loss_weight
    self.img_sim_loss = self._construct_loss(t.img_sim_loss_types,
                                             concentration=t.img_sim_concentration)
    self.img_sim_loss_weight = t.img_sim_loss_weight
    # The reference constructs this loss but never calls it (segsort.py:41-47); the
    # feature-affinity relationship is realised in segsort_softmax_densepose.py:174-222
    # as a Set-SegSort loss over tags propagated from the nearest labelled segment of
    # the same image.  That term is available here behind an explicit opt-in,
    # `config.train.evaluate_feat_aff = True`; without it the `feat_aff_*` keys
    # are parsed and ignored exactly as in the reference, so that the shipped recipes
    # (e.g. DensePose: feat_aff segsort / 0.5) optimise the reference's objective.
    self.feat_aff_loss = self._construct_loss(t.feat_aff_loss_types,
                                              concentration=t.feat_aff_concentration)
    enabled = bool(t.get('evaluate_feat_aff', False)) and t.feat_aff_loss_types == 'segsort'
    self.feat_aff_set_loss = self._construct_loss(
        'set_segsort' if enabled else 'none', concentration=t.feat_aff_concentration)
    self.feat_aff_loss_weight = t.feat_aff_loss_weight
    self.semantic_ignore_index = config.dataset.semantic_ignore_index
    self.num_classes = config.dataset.num_classes
    self.label_divisor = config.network.label_divisor

  def _construct_loss(self, loss_types, **kwargs):
    if loss_types == 'segsort':
      return segsort_loss.SegSortLoss(kwargs['concentration'], group_mode='segsort+',
                                      reduction='mean')
    if loss_types == 'set_segsort':
      return segsort_loss.SetSegSortLoss(kwargs['concentration'], group_mode='segsort+',
                                         reduction='mean')
    if loss_types == 'none':
      return None
    raise KeyError('Unsupported loss types: {:s}'.format(loss_types))

  # ------------------------------------------------------------------ predict
  def predictions(self, datas, targets={}):
    """k-NN retrieval of segment prototypes against a prototype memory
    (segsort.py:68-125)."""
    memory = targets.get('semantic_memory_prototype', None)
    memory_labels = targets.get('semantic_memory_prototype_label', None)
    emb = datas.get('cluster_embedding', None)
    clu = datas.get('cluster_index', None)
    if memory is None or memory_labels is None or emb is None or clu is None:
      return None, None
    _, clu = torch.unique(clu, return_inverse=True)
    m = int(clu.max()) + 1
    protos = segsort_common.calculate_prototypes_from_labels(emb, clu, m)
    dummy = torch.zeros(m, dtype=torch.long, device=protos.device)
    _, topk = segsort_eval.top_k_ranking(protos, dummy, memory, memory_labels, 20)
    pred = segsort_eval.majority_label_from_topk(topk)
    return pred[clu], topk[clu]

  # ------------------------------------------------------------------- losses
  # which embedding the per-image term uses (the DensePose predictor drops the location)
  img_sim_embedding_key = 'cluster_embedding_with_loc'

  def _memory_bank_ready(self, targets):
    return all(targets.get(k, []) for k in (
        'memory_prototype', 'memory_prototype_semantic_label',
        'memory_prototype_semantic_tag', 'memory_prototype_batch_index'))

  def _occurrence_sets(self, targets, use_memory, clu, bat, p_sem, p_bat):
    """Tag sets of the semantic co-occurrence term, one packed 64-bit set per pixel and per
    prototype: the image-level tags without the background column (segsort.py:147-151
    keeps them as [., T] multi-hot)."""
    nc = self.num_classes
    p_tags = targets['prototype_semantic_tag'][:, 1:nc]
    if use_memory:
      p_tags = torch.cat([p_tags] + [t[:, 1:nc] for t in targets['memory_prototype_semantic_tag']])
    # (any number of classes: only those present on both sides are packed, loss.pack_tag_set_pair)
    img_sets, p_sets = segsort_loss.pack_tag_set_pair(targets['semantic_tag'][:, 1:nc], p_tags)
    return img_sets[bat], p_sets

  def _contrastive_losses(self, datas, targets):
    """The three contrastive terms + retrieval accuracy (segsort.py:127-243)."""
    sem_ann = sem_occ = img_sim = acc = None
    nc = self.num_classes

    # per-image term, device part first: its one host read (every image's last pair id) travels
    # with the sizes the semantic terms need -- 3 host synchronisations per training step in all
    sim = None
    if self.img_sim_loss is not None:
      clu = datas['cluster_index']
      ins = datas['cluster_instance_label']
      bat = datas['cluster_batch_index']
      # pixels are image-major: every image is one contiguous slice (sizes known on the host from the
      # clustering, `cluster_image_sizes`).  The reference re-indexes (over-segmentation id, cluster)
      # pairs per image (prepare_prototype_labels, segsort.py:228-240); here ONE dense re-indexing over
      # (image, cluster, id) does it for all images, and one host read returns every image's last id
      sizes = datas.get('cluster_image_sizes', None)
      if sizes is None:
        sizes = torch.unique_consecutive(bat, return_counts=True)[1].tolist()
      sizes = [int(v) for v in sizes if int(v) > 0]
      n_img, total = len(sizes), int(clu.shape[0])
      img_of_px = torch.zeros_like(bat)                    # index of the pixel's image among the present ones
      if total > 1:
        img_of_px[1:] = torch.cumsum(bat[1:] != bat[:-1], 0)
      off = ins.max() + 1
      stride = (clu.max() + 1) * off                       # keys of image b lie in [b * stride, (b + 1) * stride)
      uniq, pair = segsort_common._unique_inverse((img_of_px * (clu.max() + 1) + clu) * off + ins,
                                                  with_uniq=False, padded=True)
      # distinct pairs of the images up to and including b = position of (b + 1) * stride among the sorted keys
      bounds = torch.arange(1, n_img + 1, device=uniq.device) * stride
      sim = (sizes, off, uniq, pair, torch.searchsorted(uniq, bounds) - 1)
    last = None

    if self.sem_ann_loss is not None or self.sem_occ_loss is not None:
      clu = datas['cluster_index']
      emb = datas['cluster_embedding']
      sem = datas['cluster_semantic_label']
      bat = datas['cluster_batch_index']
      protos = targets['prototype']
      live = protos.shape[0]          # prototypes that carry a gradient (memory bank is detached)
      p_sem = targets['prototype_semantic_label']
      p_bat = targets['prototype_batch_index']

      mem_p = targets.get('memory_prototype', [])
      mem_sem = targets.get('memory_prototype_semantic_label', [])
      mem_bat = targets.get('memory_prototype_batch_index', [])
      use_memory = self._memory_bank_ready(targets)        # memory bank (segsort.py:162-183)
      if use_memory:
        protos = torch.cat([protos] + list(mem_p), dim=0)
        p_sem = torch.cat([p_sem] + list(mem_sem), dim=0)
        p_bat = torch.cat([p_bat] + list(mem_bat), dim=0)
      px_sets, p_sets = self._occurrence_sets(targets, use_memory, clu, bat, p_sem, p_bat)

      # labelled pixels / prototypes and the index remap (segsort.py:185-195):
      # the i-th labelled prototype gets id i
      labelled = p_sem < nc
      if sim is not None:
        px, pr, last = _nonzero_pair(sem < nc, labelled, extra=sim[4])   # (one host sync for all of it)
      else:
        px, pr = _nonzero_pair(sem < nc, labelled)         # (one host sync for both sizes)
      remap = torch.cumsum(labelled, 0) - 1
      remap = torch.where(labelled, remap, torch.full_like(remap, pr.shape[0]))
      new_clu = remap[clu]

      if self.sem_ann_loss is not None:
        sem_ann = self.sem_ann_loss(emb[px], sem[px], new_clu[px], protos[pr], p_sem[pr],
                                    codes32=True)          # class ids
        sem_ann = sem_ann * self.sem_ann_loss_weight
      if self.sem_occ_loss is not None:
        sem_occ = self.sem_occ_loss(emb, px_sets, clu, protos, p_sets,
                                    prototype_grad_rows=live, codes32=self.num_classes <= 32)
        sem_occ = sem_occ * self.sem_occ_loss_weight
      acc = parallel.sharded_retrieval_accuracy(segsort_eval.top_k_ranking, protos, p_sem, 5)

    if sim is not None:
      sizes, off, uniq, pair, last_dev = sim
      emb = datas[self.img_sim_embedding_key]
      ins = datas['cluster_instance_label']
      if last is None:
        last = last_dev.tolist()                           # (no semantic term to share the read with)
      last = [int(v) for v in last]
      pair_lab = uniq[:last[-1] + 1] % off                 # over-segmentation id of every pair
      terms, lo, first = [], 0, 0
      # the prototypes of ALL images from one segment sum over the dense pair ids (a pair belongs to one image; a
      # prototype is normalised on its own), split per image: 1 launch forward + 1 backward instead of 16 + 16
      # (SPML_IMG_SIM_BATCHED_PROTOS=0: one call per image); the pixel rows are split the same way (one concatenation
      # in the backward pass instead of 16 zero-filled [P, D] gradients that are then added up)
      pr_all = e_all = None
      if os.environ.get('SPML_IMG_SIM_BATCHED_PROTOS') != '0' and sum(sizes) == int(emb.shape[0]):
        counts = [end + 1 - (last[i - 1] + 1 if i else 0) for i, end in enumerate(last)]
        pr_all = torch.split(segsort_common.calculate_prototypes_from_labels(emb, pair, last[-1] + 1), counts)
        e_all = torch.split(emb, sizes)
      # the images' terms are independent and each is a handful of launch-latency-bound kernels (16 900 pixels x ~150
      # prototypes: 340 us of GPU time per image, forward + backward, on a chip it cannot fill): they are spread over
      # SPML_IMG_SIM_STREAMS side streams (default 4; the backward of an op runs on its forward's stream)
      streams = _side_streams(emb.device, int(os.environ.get('SPML_IMG_SIM_STREAMS', '4'))) if emb.is_cuda else []
      cur = torch.cuda.current_stream(emb.device) if streams else None
      for st in streams:
        st.wait_stream(cur)
      for i, (n_px, end) in enumerate(zip(sizes, last)):
        lab, c_abs, first_i = ins[lo:lo + n_px], pair[lo:lo + n_px], first        # (views: no launch)
        e = e_all[i] if e_all is not None else emb[lo:lo + n_px]
        p_lab = pair_lab[first:end + 1]
        lo, first = lo + n_px, end + 1
        if streams:
          side = streams[i % len(streams)]
          # (allocated on the current stream, consumed on the side stream -- forward and, through the saved tensors,
          # backward: tell the caching allocator, so that a block freed host-side while a side-stream kernel still
          # reads it is not handed out again early)
          if os.environ.get('SPML_IMG_SIM_RECORD_STREAM') != '0':
            for t_ in (e, lab, c_abs, p_lab) + ((pr_all[i],) if pr_all is not None else ()):
              t_.record_stream(side)
          with torch.cuda.stream(side):
            c = c_abs - first_i                                # (on the side stream: everything it reads was
            #                                                    produced before the streams were forked)
            pr_img = pr_all[i] if pr_all is not None else \
                segsort_common.calculate_prototypes_from_labels(e, c, p_lab.shape[0])
            term = self.img_sim_loss(e, lab, c, pr_img, p_lab, codes32=True)   # over-segmentation ids
            term.record_stream(cur)
          terms.append(term)
          continue
        c = c_abs - first_i
        pr_img = pr_all[i] if pr_all is not None else \
            segsort_common.calculate_prototypes_from_labels(e, c, p_lab.shape[0])
        terms.append(self.img_sim_loss(e, lab, c, pr_img, p_lab, codes32=True))   # over-segmentation ids
      for st in streams:
        cur.wait_stream(st)
      img_sim = sum(terms) / len(terms) * self.img_sim_loss_weight

    return sem_ann, sem_occ, img_sim, acc

  def feature_affinity_loss(self, datas, targets):
    """Set-SegSort over nearest-neighbour propagated tags
    (segsort_softmax_densepose.py:174-222): every segment takes the class of its
    most similar labelled segment of the same image (cos >= 0.95, on the
    prototypes with location), untagged segments match everything."""
    import spml_amd.models.utils as model_utils
    protos = targets['prototype']
    protos_loc = targets['prototype_with_loc']
    p_sem = targets['prototype_semantic_label']
    p_bat = targets['prototype_batch_index']
    live = protos.shape[0]
    mem_p = targets.get('memory_prototype', [])
    mem_pl = targets.get('memory_prototype_with_loc', [])
    mem_sem = targets.get('memory_prototype_semantic_label', [])
    mem_bat = targets.get('memory_prototype_batch_index', [])
    if mem_p and mem_sem and mem_bat:
      protos = torch.cat([protos] + list(mem_p), dim=0)
      protos_loc = torch.cat([protos_loc] + list(mem_pl), dim=0)
      p_sem = torch.cat([p_sem] + list(mem_sem), dim=0)
      p_bat = torch.cat([p_bat] + list(mem_bat), dim=0)
    tags = model_utils.gather_multiset_labels_per_batch_by_nearest_neighbor(
        protos_loc, protos_loc, p_sem, p_bat, p_bat, num_classes=self.num_classes, top_k=1,
        threshold=0.95, label_divisor=self.label_divisor)
    untagged = tags.max(dim=1, keepdim=True)[0] == 0
    tags = tags.masked_fill(untagged.expand(-1, self.num_classes), 1)
    clu = datas['cluster_index']
    loss = self.feat_aff_set_loss(datas['cluster_embedding'], tags[clu], clu, protos, tags,
                                  prototype_grad_rows=live, codes32=self.num_classes <= 32)
    return loss * self.feat_aff_loss_weight

  def losses(self, datas, targets={}):
    return self._contrastive_losses(datas, targets)

  def forward(self, datas, targets=None, with_loss=True, with_prediction=False):
    targets = targets if targets is not None else {}
    outputs = {}
    if with_prediction:
      pred, score = self.predictions(datas, targets)
      outputs.update({'semantic_prediction': pred, 'semantic_score': score})
    if with_loss:
      sem_ann, sem_occ, img_sim, acc = self.losses(datas, targets)
      outputs.update({'sem_ann_loss': sem_ann, 'sem_occ_loss': sem_occ,
                      'img_sim_loss': img_sim, 'accuracy': acc})
      if self.feat_aff_set_loss is not None:
        outputs['feat_aff_loss'] = self.feature_affinity_loss(datas, targets)
    return outputs

  def get_params_lr(self):
    return []


def segsort(config):
  """Non-parametric prototype predictor."""
  return Segsort(config)
