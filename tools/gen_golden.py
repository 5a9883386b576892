#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Runs only in the build container, where the reference tree is mounted
read-only at /root/reference.  It imports the reference's own leaf functions
(nothing is copied), feeds them seeded synthetic inputs on CPU / fp32 and stores
inputs + outputs as small .npz files.  The oracle (oracle/spml_oracle.py) and,
on the GPU box, the HIP path are then checked against these files.

Two in-memory shims are needed to run the reference on CPU:
  * spml/utils/segsort/common.py:376 reads ``tensor.device.index`` which is
    None on CPU -> the function source is exec'd with ``(… .index or 0)``;
  * torch.nn.parallel.scatter_gather.gather asserts on CPU tensors ->
    replaced by torch.cat while B1 goldens are generated.

Usage:  python tools/gen_golden.py  [--ref /root/reference] [--out tests/golden]
"""

import argparse
import inspect
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True


def _np(x):
  if isinstance(x, torch.Tensor):
    return x.detach().cpu().numpy()
  return np.asarray(x)


ONLY = None      # --only a,b: write just these fixtures (the others stay as committed)


def save(out_dir, name, **arrays):
  if ONLY is not None and name not in ONLY:
    return
  path = os.path.join(out_dir, name + '.npz')
  np.savez_compressed(path, **{k: _np(v) for k, v in arrays.items()})
  print('wrote %-40s %7.1f KB' % (path, os.path.getsize(path) / 1024.0))


def coherent_embedding(gen, n, c, h, w, blobs=5, noise=0.35):
  """Spatially coherent embedding map: a few random directions blended by
  smooth spatial weights plus noise (so k-means has structure to find)."""
  dirs = torch.randn(blobs, c, generator=gen)
  cy = torch.rand(n, blobs, generator=gen)
  cx = torch.rand(n, blobs, generator=gen)
  yy = torch.linspace(0, 1, h).view(1, 1, h, 1)
  xx = torch.linspace(0, 1, w).view(1, 1, 1, w)
  wgt = torch.exp(-((yy - cy.view(n, blobs, 1, 1)) ** 2 +
                    (xx - cx.view(n, blobs, 1, 1)) ** 2) / 0.05)
  emb = torch.einsum('nbhw,bc->nchw', wgt, dirs)
  emb = emb + noise * torch.randn(n, c, h, w, generator=gen)
  return emb.float()


def blocky_labels(gen, n, h, w, cells, low, high):
  grid = torch.randint(low, high, (n, cells, cells), generator=gen)
  iy = (torch.arange(h) * cells // h).clamp(max=cells - 1)
  ix = (torch.arange(w) * cells // w).clamp(max=cells - 1)
  return grid[:, iy][:, :, ix].long()


class AttrDict(dict):
  __getattr__ = dict.__getitem__


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--ref', default='/root/reference')
  ap.add_argument('--out', default=os.path.join(
      os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden'))
  ap.add_argument('--only', default=None, help='comma separated fixture names')
  args = ap.parse_args()
  global ONLY
  ONLY = set(args.only.split(',')) if args.only else None
  out = os.path.abspath(args.out)
  os.makedirs(out, exist_ok=True)

  sys.path.insert(0, args.ref)
  torch.set_num_threads(1)       # bit-stable fp32 sums

  import spml.utils.general.common as g_common
  import spml.utils.segsort.common as s_common
  import spml.utils.segsort.loss as s_loss
  import spml.utils.segsort.eval as s_eval
  import spml.models.utils as m_utils
  import spml.utils.general.train as g_train
  import spml.models.predictions.segsort as p_segsort

  # --- shim 1: device.index is None on CPU (common.py:376) ------------------
  src = inspect.getsource(s_common.segment_by_kmeans)
  assert 'cur_cluster_indices.device.index' in src
  src = src.replace('cur_cluster_indices.device.index',
                    '(cur_cluster_indices.device.index or GPU_ID)')
  ns = dict(s_common.__dict__)
  ns['GPU_ID'] = 0
  exec(compile(src, '<segment_by_kmeans+cpu-shim>', 'exec'), ns)
  ref_segment_by_kmeans = ns['segment_by_kmeans']

  def segment_by_kmeans_rank(gpu_id, *a, **kw):
    ns['GPU_ID'] = gpu_id
    try:
      return ref_segment_by_kmeans(*a, **kw)
    finally:
      ns['GPU_ID'] = 0

  # --- shim 2: scatter_gather.gather asserts on CPU --------------------------
  m_utils.scatter_gather = types.SimpleNamespace(
      gather=lambda xs, dev: torch.cat(list(xs), 0))

  # ======================= A1 / A2 / A3 / A13 ================================
  g = torch.Generator().manual_seed(235)
  x = torch.randn(7, 5, 19, generator=g)
  x[0, 0] = 0.0                     # zero row -> eps branch
  x[1, 2] *= 1e-14                  # tiny norm < eps
  save(out, 'a01_normalize', x=x, y=g_common.normalize_embedding(x))

  grids = {}
  for tag, (k, hw) in {'k3_17': ((3, 3), (17, 17)), 'k6_130': ((6, 6), (130, 130)),
                       'k6_128': ((6, 6), (128, 128)), 'k12_194': ((12, 12), (194, 194)),
                       'k32_258': ((32, 32), (258, 258)), 'k6_513': ((6, 6), (513, 513)),
                       'k4x5_33x29': ((4, 5), (33, 29)), 'k12_512': ((12, 12), (512, 512)),
                       'k2_3': ((2, 2), (3, 3))}.items():
    grids['init_' + tag] = s_common.initialize_cluster_labels(k, hw, 'cpu')
    grids['argk_' + tag] = np.array(k)
  save(out, 'a03_init_grid', **grids)

  locs = {}
  for tag, hw in {'17x17': (17, 17), '33x29': (33, 29), '130x130': (130, 130)}.items():
    locs['float_' + tag] = s_common.generate_location_features(hw, 'cpu', 'float')
    locs['int_' + tag] = s_common.generate_location_features(hw, 'cpu', 'int')
  save(out, 'a02_location', **locs)

  lab = torch.randint(0, 9, (2, 11, 13), generator=g)
  save(out, 'a13_onehot_resize', lab=lab, onehot=g_common.one_hot(lab),
       onehot12=g_common.one_hot(lab, 12),
       big=blocky_labels(g, 2, 65, 65, 5, 0, 255),
       **{'resized_%d' % s: g_common.resize_labels(
           blocky_labels(torch.Generator().manual_seed(7), 2, 65, 65, 5, 0, 255), (s, s))
          for s in (17, 33, 130)})
  save(out, 'a13_resize_src',
       src=blocky_labels(torch.Generator().manual_seed(7), 2, 65, 65, 5, 0, 255))

  # ======================= A4 / A5 / A6 ======================================
  for tag, (p, d, k, it) in {'tiny': (289, 10, 9, 10), 'small': (1089, 66, 36, 10),
                             'k144': (2000, 34, 144, 6)}.items():
    g = torch.Generator().manual_seed(1000 + p)
    side = int(round(p ** 0.5))
    if side * side == p:
      emb = coherent_embedding(g, 1, d, side, side)[0].permute(1, 2, 0).reshape(p, d)
    else:
      emb = torch.randn(p, d, generator=g)
    emb = g_common.normalize_embedding(emb)
    init = torch.randint(0, k, (p,), generator=g)
    init[:k] = torch.arange(k)
    if tag == 'tiny':
      init[init == 4] = 3           # cluster 4 starts empty -> zero prototype
    protos0 = s_common.calculate_prototypes_from_labels(emb, init, k)
    near0 = s_common.find_nearest_prototypes(emb, protos0)
    labels = init
    per_iter, per_proto, per_margin = [], [], []
    for _ in range(it):
      pr = s_common.calculate_prototypes_from_labels(emb, labels, k)
      sims = torch.mm(emb, pr.t())
      labels = s_common.find_nearest_prototypes(emb, pr)
      t2 = torch.topk(sims, 2, dim=1).values
      per_iter.append(labels)
      per_proto.append(pr)
      per_margin.append(t2[:, 0] - t2[:, 1])
    final = s_common.kmeans_with_initial_labels(emb, init, k, it)
    assert torch.equal(final, labels)
    save(out, 'a06_kmeans_' + tag, emb=emb, init=init, k=np.array(k),
         iterations=np.array(it), protos0=protos0, nearest0=near0,
         labels_per_iter=torch.stack(per_iter), protos_per_iter=torch.stack(per_proto),
         margin_per_iter=torch.stack(per_margin), final=final)

  # ======================= A7 / A14 ==========================================
  sem = torch.tensor([3, 3, 5, 5, 3, 7])
  ins = torch.tensor([0, 0, 0, 1, 1, 1])
  pl, inv = s_common.prepare_prototype_labels(sem, ins, 8)
  g = torch.Generator().manual_seed(77)
  sem2 = torch.randint(0, 21, (500,), generator=g)
  ins2 = torch.randint(0, 40, (500,), generator=g)
  pl2, inv2 = s_common.prepare_prototype_labels(sem2, ins2, 21)
  sel, major = s_common.find_majority_label_index(sem2, ins2)
  save(out, 'a07_labels', sem=sem, ins=ins, off=np.array(8), plab=pl, inv=inv,
       sem2=sem2, ins2=ins2, off2=np.array(21), plab2=pl2, inv2=inv2,
       major_sel=sel, major_lab=major)

  # ======================= A8 segment_by_kmeans ==============================
  for tag, (n, c, h, w, k, div, gpu) in {
      'tiny': (2, 8, 17, 17, (3, 3), 256, 0),
      'small': (2, 32, 29, 29, (6, 6), 2048, 0),
      'rank1': (2, 16, 21, 25, (4, 3), 2048, 1)}.items():
    g = torch.Generator().manual_seed(4000 + c)
    emb = coherent_embedding(g, n, c, h, w)
    sem = blocky_labels(g, n, h, w, 3, 0, 21)
    # unlabelled (254) outside a few blobs, ignore strip (255) bottom/right
    keep = blocky_labels(g, n, h, w, 6, 0, 4) == 0
    sem = torch.where(keep, sem, torch.full_like(sem, 254))
    sem[:, -2:, :] = 255
    sem[:, :, -3:] = 255
    ins = blocky_labels(g, n, h, w, 4, 0, 200)
    labels = sem * div + ins
    ignore = int(labels.max()) + 1
    labels = labels.masked_fill(sem == 255, ignore)
    loc = (s_common.generate_location_features((h, w), 'cpu', 'float') - 0.5
           ).unsqueeze(0).expand(n, h, w, 2)
    o = segment_by_kmeans_rank(gpu, emb, labels, list(k), local_features=loc,
                               ignore_index=ignore, iterations=10)
    o2 = segment_by_kmeans_rank(gpu, emb, None, list(k), iterations=3)
    save(out, 'a08_segment_' + tag, emb=emb, labels=labels, sem=sem, ins=ins,
         k=np.array(k), ignore=np.array(ignore), gpu=np.array(gpu), loc=loc,
         div=np.array(div),
         o_emb=o[0], o_embloc=o[1], o_lab=o[2], o_clu=o[3], o_bat=o[4],
         d_emb=o2[0], d_embloc=o2[1], d_lab=o2[2], d_clu=o2[3], d_bat=o2[4])

  # ======================= A9 / A10 losses (fwd + grads) =====================
  for tag, (p, m, d, ncls, kappa) in {'tiny': (200, 23, 10, 5, 6.0),
                                      'small': (700, 150, 64, 21, 12.0),
                                      'loc': (400, 60, 66, 21, 16.0)}.items():
    g = torch.Generator().manual_seed(9000 + p)
    protos = g_common.normalize_embedding(torch.randn(m, d, generator=g))
    own = torch.randint(0, m, (p,), generator=g)
    emb = g_common.normalize_embedding(
        protos[own] + 0.7 * torch.randn(p, d, generator=g))
    p_sem = torch.randint(0, ncls, (m,), generator=g)
    sem = p_sem[own].clone()
    flip = torch.rand(p, generator=g) < 0.1
    sem[flip] = torch.randint(0, ncls, (int(flip.sum()),), generator=g)
    # make one class have a single prototype -> 'pos <= 0' fallback branch
    p_sem[0] = ncls + 3
    sem[own == 0] = ncls + 3
    emb_r = emb.clone().requires_grad_(True)
    pro_r = protos.clone().requires_grad_(True)
    nll = s_loss._calculate_log_likelihood(emb_r, sem, own, pro_r, p_sem, kappa,
                                           'segsort+')
    loss = s_loss.SegSortLoss(kappa, 'segsort+', reduction='mean')(
        emb_r, sem, own, pro_r, p_sem)
    loss.backward()
    # tag sets
    p_tags = (torch.rand(m, ncls - 1, generator=g) < 0.15).long()
    p_tags[torch.arange(m), torch.randint(0, ncls - 1, (m,), generator=g)] = 1
    p_tags[1] = 0                                  # a prototype with no tag
    tags = p_tags[own].clone()
    tflip = torch.rand(p, generator=g) < 0.1
    tags[tflip] = (torch.rand(int(tflip.sum()), ncls - 1, generator=g) < 0.1).long()
    emb_s = emb.clone().requires_grad_(True)
    pro_s = protos.clone().requires_grad_(True)
    snll = s_loss._one_hot_calculate_log_likelihood(emb_s, tags, own, pro_s, p_tags,
                                                    kappa, 'segsort+')
    sloss = s_loss.SetSegSortLoss(kappa, 'segsort+', reduction='mean')(
        emb_s, tags, own, pro_s, p_tags)
    sloss.backward()
    save(out, 'a09_loss_' + tag, emb=emb, protos=protos, own=own, sem=sem,
         p_sem=p_sem, kappa=np.array(kappa), nll=nll, loss=loss,
         d_emb=emb_r.grad, d_protos=pro_r.grad,
         tags=tags, p_tags=p_tags, set_nll=snll, set_loss=sloss,
         set_d_emb=emb_s.grad, set_d_protos=pro_s.grad)

  # ======================= A11 / A12 =========================================
  g = torch.Generator().manual_seed(1111)
  q = g_common.normalize_embedding(torch.randn(150, 32, generator=g))
  pr = g_common.normalize_embedding(torch.randn(400, 32, generator=g))
  ql = torch.randint(0, 21, (150,), generator=g)
  prl = torch.randint(0, 21, (400,), generator=g)
  acc5, top5 = s_eval.top_k_ranking(q, ql, pr, prl, 5)
  acc20, top20 = s_eval.top_k_ranking(q, ql, pr, prl, 20)
  accs, tops = s_eval.top_k_ranking(pr, prl, pr, prl, 5)
  save(out, 'a11_topk', q=q, ql=ql, pr=pr, prl=prl, acc5=acc5, top5=top5,
       acc20=acc20, top20=top20, acc_self=accs, top_self=tops,
       major20=s_eval.majority_label_from_topk(top20),
       major20_21=s_eval.majority_label_from_topk(top20, 21))

  # ======================= B1 (2 shards) + B3 ================================
  shards = []
  for gpu in (0, 1):
    g = torch.Generator().manual_seed(500 + gpu)
    n, c, h, w, k, div = 2, 16, 19, 23, (3, 3), 2048
    emb = coherent_embedding(g, n, c, h, w)
    sem = blocky_labels(g, n, h, w, 3, 0, 21)
    keep = blocky_labels(g, n, h, w, 5, 0, 3) == 0
    sem = torch.where(keep, sem, torch.full_like(sem, 254))
    sem[:, -2:, :] = 255
    ins = blocky_labels(g, n, h, w, 4, 0, 50)
    labels = (sem * div + ins)
    ignore = int(labels.max()) + 1
    labels = labels.masked_fill(sem == 255, ignore)
    o = segment_by_kmeans_rank(gpu, emb, labels, list(k), ignore_index=ignore,
                               iterations=5)
    shards.append(dict(emb=o[0], embloc=o[1], sem=o[2] // div, ins=o[2] % div,
                       clu=o[3], bat=o[4], raw_emb=emb, raw_labels=labels,
                       ignore=ignore))
  embs = [s['emb'].clone().requires_grad_(True) for s in shards]
  emls = [s['embloc'].clone().requires_grad_(True) for s in shards]
  res = m_utils.gather_clustering_and_update_prototypes(
      embs, emls, [s['clu'] for s in shards], [s['bat'] for s in shards],
      [s['sem'] for s in shards], [s['ins'] for s in shards], 'cpu')
  protos, protos_loc, p_sem, p_ins, p_bat, new_clu = res
  wgt = torch.randn(protos[0].shape, generator=g)
  wgt2 = torch.randn(protos_loc[0].shape, generator=g)
  ((protos[0] * wgt).sum() + (protos_loc[0] * wgt2).sum()).backward()
  b1 = {}
  for i, s in enumerate(shards):
    for kname in ('emb', 'embloc', 'sem', 'ins', 'clu', 'bat', 'raw_emb', 'raw_labels'):
      b1['s%d_%s' % (i, kname)] = s[kname]
    b1['s%d_ignore' % i] = np.array(s['ignore'])
    b1['s%d_new_clu' % i] = new_clu[i]
    b1['s%d_d_emb' % i] = embs[i].grad
    b1['s%d_d_embloc' % i] = emls[i].grad
  save(out, 'b01_gather', protos=protos[0], protos_loc=protos_loc[0], p_sem=p_sem[0],
       p_ins=p_ins[0], p_bat=p_bat[0], wgt=wgt, wgt2=wgt2, **b1)

  all_emb = torch.cat([s['emb'] for s in shards])
  all_bat = torch.cat([s['bat'] for s in shards])
  ms = m_utils.gather_multiset_labels_per_batch_by_nearest_neighbor(
      all_emb, protos[0].detach(), p_sem[0], all_bat, p_bat[0],
      num_classes=21, top_k=3, threshold=0.6)
  save(out, 'b03_multiset', emb=all_emb, bat=all_bat, protos=protos[0], p_sem=p_sem[0],
       p_bat=p_bat[0], out=ms, threshold=np.array(0.6))

  # ======================= F1-F3: Segsort.losses, with memory bank ===========
  cfg = AttrDict(
      train=AttrDict(sem_ann_loss_types='segsort', sem_occ_loss_types='segsort',
                     img_sim_loss_types='segsort', feat_aff_loss_types='none',
                     sem_ann_concentration=6.0, sem_occ_concentration=12.0,
                     img_sim_concentration=16.0, feat_aff_concentration=0.0,
                     sem_ann_loss_weight=1.0, sem_occ_loss_weight=0.5,
                     img_sim_loss_weight=0.1, feat_aff_loss_weight=0.0),
      dataset=AttrDict(semantic_ignore_index=255, num_classes=21),
      network=AttrDict(label_divisor=2048))
  model = p_segsort.Segsort(cfg)
  s = shards[0]
  one = m_utils.gather_clustering_and_update_prototypes(
      [s['emb']], [s['embloc']], [s['clu']], [s['bat']], [s['sem']], [s['ins']], 'cpu')
  sem_tag = torch.zeros(2, 256, dtype=torch.long)
  for b in range(2):
    present = torch.unique(s['sem'][s['bat'] == b])
    sem_tag[b, present] = 1
  emb_r = s['emb'].clone().requires_grad_(True)
  eml_r = s['embloc'].clone().requires_grad_(True)
  datas = {'cluster_index': one[5][0], 'cluster_embedding': emb_r,
           'cluster_embedding_with_loc': eml_r,
           'cluster_semantic_label': s['sem'], 'cluster_instance_label': s['ins'],
           'cluster_batch_index': s['bat']}
  # memory bank made from shard 1's prototypes (detached), batch index shifted
  s1 = shards[1]
  mem = m_utils.gather_clustering_and_update_prototypes(
      [s1['emb']], [s1['embloc']], [s1['clu']], [s1['bat']], [s1['sem']], [s1['ins']], 'cpu')
  mem_tag = torch.zeros(4, 256, dtype=torch.long)
  for b in (2, 3):
    present = torch.unique(s1['sem'][s1['bat'] == b])
    mem_tag[b, present] = 1
  targets = {'prototype': one[0][0].detach(), 'prototype_semantic_label': one[2][0],
             'prototype_batch_index': one[4][0], 'semantic_tag': sem_tag,
             'prototype_semantic_tag': sem_tag[one[4][0]],
             'memory_prototype': [mem[0][0].detach()],
             'memory_prototype_semantic_label': [mem[2][0]],
             'memory_prototype_batch_index': [mem[4][0]],
             'memory_prototype_semantic_tag': [mem_tag[mem[4][0]]]}
  l_ann, l_occ, l_img, acc = model.losses(datas, targets)
  (l_ann + l_occ + l_img).backward()
  targets_nomem = {k: v for k, v in targets.items() if not k.startswith('memory')}
  n_ann, n_occ, n_img, n_acc = model.losses(
      {k: (v.detach() if v.is_floating_point() else v) for k, v in datas.items()},
      targets_nomem)
  save(out, 'f01_segsort_losses',
       clu=one[5][0], emb=s['emb'], embloc=s['embloc'], sem=s['sem'], ins=s['ins'],
       bat=s['bat'], protos=one[0][0], p_sem=one[2][0], p_bat=one[4][0],
       sem_tag=sem_tag, mem_protos=mem[0][0], mem_p_sem=mem[2][0], mem_p_bat=mem[4][0],
       mem_tag=mem_tag[mem[4][0]],
       l_ann=l_ann, l_occ=l_occ, l_img=l_img, acc=acc,
       d_emb=emb_r.grad, d_embloc=eml_r.grad,
       n_ann=n_ann, n_occ=n_occ, n_img=n_img, n_acc=n_acc)

  # ======================= N1: kNN predictions + memory-bank files ============
  # Segsort.predictions (segsort.py:68-125): prototypes of the (gappy) cluster ids,
  # 20-NN retrieval in the memory bank in 10 groups, majority vote, scatter to pixels.
  import spml.utils.segsort.others as s_others
  gappy = one[5][0] * 3 + 1                       # ids with holes: exercises the unique()
  bank = torch.cat([mem[0][0], one[0][0]], 0).detach()
  bank_lab = torch.cat([mem[2][0], one[2][0]], 0)
  assert bank.shape[0] >= 20
  pred, topk = model.predictions(
      {'cluster_embedding': s['emb'], 'cluster_index': gappy},
      {'semantic_memory_prototype': bank, 'semantic_memory_prototype_label': bank_lab})
  # the on-disk memory bank of prototype.py:207-211 read back by others.py:11-41
  bank_dir = os.path.join(out, 'n1_memory_bank')
  os.makedirs(bank_dir, exist_ok=True)
  half = bank.shape[0] // 2
  if ONLY is None or 'n1_predictions' in ONLY or not os.path.exists(os.path.join(bank_dir, '2007_000032.npy')):
    np.save(os.path.join(bank_dir, '2007_000032.npy'),
            {'prototype': bank[:half].numpy(), 'prototype_label': bank_lab[:half].numpy()})
    np.save(os.path.join(bank_dir, '2007_000039.npy'),
            {'prototype': bank[half:].numpy(), 'prototype_label': bank_lab[half:].numpy()})
  loaded, loaded_lab = s_others.load_memory_banks(bank_dir)
  save(out, 'n1_predictions', emb=s['emb'], clu=gappy, bank=bank, bank_lab=bank_lab,
       pred=pred, topk=topk, loaded=loaded, loaded_lab=loaded_lab)

  # ======================= N4: DensePose embedding variant + predictor ========
  # resnet_pspnet_densepose.py: 5-channel local features (location + smoothed, normalised
  # colour), k-means on C+5 channels, embedding-with-local rebuilt from 0.1 * embedding;
  # segsort_softmax_densepose.py: tags propagated from the nearest labelled segment.
  import spml.models.embeddings.resnet_pspnet_densepose as e_dp
  import spml.models.predictions.segsort_softmax_densepose as p_dp
  cfg_dp = AttrDict(
      train=AttrDict(sem_ann_loss_types='segsort', sem_occ_loss_types='segsort',
                     img_sim_loss_types='segsort', feat_aff_loss_types='none',
                     sem_ann_concentration=6.0, sem_occ_concentration=12.0,
                     img_sim_concentration=16.0, feat_aff_concentration=0.0,
                     sem_ann_loss_weight=1.0, sem_occ_loss_weight=0.5,
                     img_sim_loss_weight=0.1, feat_aff_loss_weight=0.0),
      dataset=AttrDict(semantic_ignore_index=255, num_classes=15),
      network=AttrDict(label_divisor=2048, embedding_dim=16, kmeans_num_clusters=[3, 3],
                       kmeans_iterations=5, use_syncbn=False, backbone_types='panoptic_pspnet_101'))
  g = torch.Generator().manual_seed(77)
  torch.manual_seed(77)
  net = e_dp.ResnetPspnet([1, 1, 1, 1], [1, 2, 1, 1], [1, 1, 2, 4], cfg_dp).eval()
  n, c, h, w = 2, 16, 22, 18
  image = torch.nn.functional.interpolate(torch.randn(n, 3, 12, 10, generator=g), size=(88, 72),
                                          mode='bilinear', align_corners=False)
  image = image + 0.1 * torch.randn(n, 3, 88, 72, generator=g)
  dp_emb = coherent_embedding(g, n, c, h, w)
  dp_sem = blocky_labels(g, n, h, w, 3, 0, 15)
  dp_sem = torch.where(blocky_labels(g, n, h, w, 5, 0, 3) == 0, dp_sem, torch.full_like(dp_sem, 254))
  dp_sem[:, :, -3:] = 255
  dp_ins = blocky_labels(g, n, h, w, 4, 0, 40)
  with torch.no_grad():
    dp_local = net.lfn(image, size=(h, w))
  assert dp_local.shape[-1] == 5
  orig_sbk = e_dp.segsort_common.segment_by_kmeans
  e_dp.segsort_common.segment_by_kmeans = ref_segment_by_kmeans
  try:
    dp_out = net.generate_clusters(dp_emb, dp_sem, dp_ins, dp_local)
  finally:
    e_dp.segsort_common.segment_by_kmeans = orig_sbk

  pred_dp = p_dp.SegsortSoftmax(cfg_dp).eval()
  s0 = shards[0]
  dp_nc = 15
  sem15 = torch.where(s0['sem'] < 21, s0['sem'] % dp_nc, s0['sem'])     # classes of this recipe
  p_sem15 = torch.where(one[2][0] < 21, one[2][0] % dp_nc, one[2][0])
  m_sem15 = torch.where(mem[2][0] < 21, mem[2][0] % dp_nc, mem[2][0])
  emb_r = s0['emb'].clone().requires_grad_(True)
  fmap = torch.randn(2, 16, 11, 13, generator=g)
  flab = blocky_labels(g, 2, 40, 44, 3, 0, 17)
  flab[:, :4] = 255
  datas_dp = {'cluster_index': one[5][0], 'cluster_embedding': emb_r,
              'cluster_embedding_with_loc': s0['embloc'],
              'cluster_semantic_label': sem15, 'cluster_instance_label': s0['ins'],
              'cluster_batch_index': s0['bat'], 'embedding': fmap}
  targets_dp = {'prototype': one[0][0].detach(), 'prototype_with_loc': one[1][0].detach(),
                'prototype_semantic_label': p_sem15, 'prototype_batch_index': one[4][0],
                'semantic_label': flab.clone(),
                'memory_prototype': [mem[0][0].detach()],
                'memory_prototype_with_loc': [mem[1][0].detach()],
                'memory_prototype_semantic_label': [m_sem15],
                'memory_prototype_batch_index': [mem[4][0]]}
  dl_ann, dl_occ, dl_img, dl_acc = pred_dp.losses(datas_dp, targets_dp)
  (dl_ann + dl_occ + dl_img).backward()
  all_loc = torch.cat([one[1][0], mem[1][0]], 0).detach()
  all_sem = torch.cat([p_sem15, m_sem15], 0)
  all_bat = torch.cat([one[4][0], mem[4][0]], 0)
  prop_tags = m_utils.gather_multiset_labels_per_batch_by_nearest_neighbor(
      all_loc, all_loc, all_sem, all_bat, all_bat, num_classes=dp_nc, top_k=1, threshold=0.95,
      label_divisor=2048)
  save(out, 'n4_densepose',
       image=image, emb_map=dp_emb, sem_map=dp_sem, ins_map=dp_ins, local=dp_local,
       o_emb=dp_out['cluster_embedding'], o_embloc=dp_out['cluster_embedding_with_loc'],
       o_sem=dp_out['cluster_semantic_label'], o_ins=dp_out['cluster_instance_label'],
       o_clu=dp_out['cluster_index'], o_bat=dp_out['cluster_batch_index'],
       cls_w0=pred_dp.semantic_classifier[0].weight, cls_bn_w=pred_dp.semantic_classifier[1].weight,
       cls_bn_b=pred_dp.semantic_classifier[1].bias, cls_w4=pred_dp.semantic_classifier[4].weight,
       cls_b4=pred_dp.semantic_classifier[4].bias,
       clu=one[5][0], emb=s0['emb'], embloc=s0['embloc'], sem=sem15, ins=s0['ins'], bat=s0['bat'],
       fmap=fmap, flab=flab, protos=one[0][0], protos_loc=one[1][0], p_sem=p_sem15, p_bat=one[4][0],
       mem_protos=mem[0][0], mem_protos_loc=mem[1][0], mem_p_sem=m_sem15, mem_p_bat=mem[4][0],
       l_ann=dl_ann, l_occ=dl_occ, l_img=dl_img, acc=dl_acc, d_emb=emb_r.grad, prop_tags=prop_tags)

  # ======================= H1: two training steps of the reference ===========
  # pyscripts/train/train.py:154-309 on ONE device with the reference's own model classes
  # (ResnetDeeplab + SegsortSoftmax -- train.py:31 binds `segsort` to the softmax variant)
  # and lib.nn.optimizer.SGD: embeddings + k-means, prototypes, tags, memory bank, losses,
  # poly lr, SGD.step(lr), memory-bank FIFO with the batch-index shift.  Step 1 runs with
  # the memory bank filled by step 0.  Weights: tests/tools_synth.reinit_parameters (the
  # same function re-creates them in the tests), inputs: spml_amd.synth.make_batch.
  import spml.models.embeddings.resnet_deeplab as e_dl
  import spml.models.predictions.segsort_softmax as p_soft
  import lib.nn.optimizer as ref_opt
  sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
  sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
  from tools_synth import reinit_parameters, parameter_checksums
  from spml_amd import synth
  cfg_h1 = AttrDict(
      train=AttrDict(sem_ann_loss_types='segsort', sem_occ_loss_types='segsort',
                     img_sim_loss_types='segsort', feat_aff_loss_types='none',
                     sem_ann_concentration=6.0, sem_occ_concentration=12.0,
                     img_sim_concentration=16.0, feat_aff_concentration=0.0,
                     sem_ann_loss_weight=1.0, sem_occ_loss_weight=0.5,
                     img_sim_loss_weight=0.1, feat_aff_loss_weight=0.0,
                     base_lr=3e-3, max_iteration=30000, warmup_iteration=100, momentum=0.9,
                     weight_decay=5e-4, batch_size=2, memory_bank_size=2),
      dataset=AttrDict(semantic_ignore_index=255, num_classes=21),
      network=AttrDict(label_divisor=2048, embedding_dim=16, kmeans_num_clusters=[4, 4],
                       kmeans_iterations=5, use_syncbn=False,
                       backbone_types='panoptic_deeplab_101'))
  def run_h1(dropout):
    """dropout=True: the head's nn.Dropout(0.75) draws from torch's global CPU generator,
    re-seeded at the start of every step (the oracle test does the same); False: p = 0 on the
    reference module instance -- the variant the GPU path is compared with (its dropout
    mask comes from another generator)."""
    h_emb = reinit_parameters(e_dl.ResnetDeeplab([1, 1, 1, 1], [1, 2, 1, 1], [1, 1, 2, 4], cfg_h1), 31)
    h_pred = reinit_parameters(p_soft.SegsortSoftmax(cfg_h1), 32)
    h_emb.train(); h_pred.train()
    if not dropout:
      h_pred.semantic_classifier[3].p = 0.0
    h_opt = ref_opt.SGD(h_emb.get_params_lr() + h_pred.get_params_lr(), lr=1,
                        momentum=cfg_h1.train.momentum, weight_decay=cfg_h1.train.weight_decay)
    h_opt.zero_grad()
    orig_sbk = e_dl.segsort_common.segment_by_kmeans
    e_dl.segsort_common.segment_by_kmeans = ref_segment_by_kmeans
    h_store = {}
    memory_banks = {}
    num_gpus, h_iter0 = 1, 57          # inside the warm-up ramp of lr_poly
    try:
      for step in range(2):
        torch.manual_seed(4000 + step)
        datas, targets = synth.make_batch(2, 161, seed=900 + step)
        image_batch, label_batch = [datas], [dict(targets)]
        embeddings = [h_emb(image_batch[0], label_batch[0])]
        embeddings[0]['embedding'].retain_grad()
        seg_ids = embeddings[0]['cluster_index'].clone()     # as segment_by_kmeans returned them
        (prototypes, prototypes_with_loc, prototype_semantic_labels, prototype_instance_labels,
         prototype_batch_indices, cluster_indices) = m_utils.gather_clustering_and_update_prototypes(
             [e['cluster_embedding'] for e in embeddings],
             [e['cluster_embedding_with_loc'] for e in embeddings],
             [e['cluster_index'] for e in embeddings], [e['cluster_batch_index'] for e in embeddings],
             [e['cluster_semantic_label'] for e in embeddings],
             [e['cluster_instance_label'] for e in embeddings], 'cpu')
        label_batch[0]['prototype'] = prototypes[0]
        label_batch[0]['prototype_with_loc'] = prototypes_with_loc[0]
        label_batch[0]['prototype_semantic_label'] = prototype_semantic_labels[0]
        label_batch[0]['prototype_instance_label'] = prototype_instance_labels[0]
        label_batch[0]['prototype_batch_index'] = prototype_batch_indices[0]
        embeddings[0]['cluster_index'] = cluster_indices[0]
        semantic_tags = m_utils.gather_and_update_datas([label_batch[0]['semantic_tag']], 'cpu')
        label_batch[0]['semantic_tag'] = semantic_tags[0]
        label_batch[0]['prototype_semantic_tag'] = torch.index_select(
            semantic_tags[0], 0, label_batch[0]['prototype_batch_index'])
        for k in memory_banks.keys():
          assert label_batch[0].get(k, None) is None
          label_batch[0][k] = list(memory_banks[k])
        outputs = h_pred(embeddings[0], label_batch[0])
        losses = []
        for k in ['sem_ann_loss', 'sem_occ_loss', 'img_sim_loss', 'feat_aff_loss']:
          if outputs.get(k, None) is not None:
            outputs[k] = outputs[k].mean()
            losses.append(outputs[k])
        loss = sum(losses)
        lr = g_train.lr_poly(cfg_h1.train.base_lr, h_iter0 + step, cfg_h1.train.max_iteration,
                             cfg_h1.train.warmup_iteration)
        h_opt.zero_grad()
        loss.backward()
        h_opt.step(lr)
        with torch.no_grad():
          for k in list(label_batch[0].keys()):
            if 'prototype' in k and 'memory' not in k:
              memory_banks.setdefault('memory_' + k, []).append(label_batch[0][k].clone().detach())
              if len(memory_banks['memory_' + k]) > cfg_h1.train.memory_bank_size:
                memory_banks['memory_' + k] = memory_banks['memory_' + k][1:]
          for mem_lab in memory_banks.get('memory_prototype_batch_index', []):
            mem_lab += cfg_h1.train.batch_size * num_gpus
        names_e, sums_e = parameter_checksums(h_emb)
        names_p, sums_p = parameter_checksums(h_pred)
        t = 's%d_' % step
        h_store.update({
            t + 'image_seed': np.array(900 + step), t + 'image_head': datas['image'].reshape(-1)[:64],
            t + 'image_sums': np.array([datas['image'].double().sum().item(), datas['image'].double().abs().sum().item()]),
            t + 'semantic_label': targets['semantic_label'].to(torch.int16),
            t + 'instance_label': targets['instance_label'].to(torch.int16),
            t + 'semantic_tag': targets['semantic_tag'].to(torch.int16),
            t + 'sem_ann_loss': outputs['sem_ann_loss'], t + 'sem_occ_loss': outputs['sem_occ_loss'],
            t + 'img_sim_loss': outputs['img_sim_loss'], t + 'accuracy': outputs['accuracy'].mean(),
            t + 'loss': loss, t + 'lr': np.array(lr), t + 'n_prototypes': np.array(prototypes[0].shape[0]),
            t + 'emb_param_sums': sums_e, t + 'pred_param_sums': sums_p,
            t + 'aspp_w_head': dict(h_emb.named_parameters())['aspp.aspp_1.0.weight'].detach().reshape(-1)[:256].clone(),
            t + 'cls_w_head': dict(h_pred.named_parameters())['semantic_classifier.4.weight'].detach().reshape(-1)[:256].clone(),
        })
        if not dropout:
          # for the GPU step test's "given clustering" mode: the reference's own segment ids (the
          # chaotic k-means outcome, injected on the GPU) and d loss / d embedding map (every
          # 7th element + the two sums) -- everything downstream of the clustering is smooth
          d_emb = embeddings[0]['embedding'].grad.reshape(-1)
          h_store.update({
              t + 'cluster_index': seg_ids.to(torch.int16),
              t + 'd_embedding_strided': d_emb[::7].clone(),
              t + 'd_embedding_sums': np.array([d_emb.double().sum().item(), d_emb.double().abs().sum().item()]),
          })
    finally:
      e_dl.segsort_common.segment_by_kmeans = orig_sbk
    h_store['iter0'] = np.array(h_iter0)
    h_store['emb_param_names'] = np.array(names_e)
    h_store['pred_param_names'] = np.array(names_p)
    save(out, 'h01_step' if dropout else 'h01_step_nodrop', **h_store)

  run_h1(True)
  run_h1(False)

  # SGD alone: the reference class on a few tensors, three steps with changing lr / groups
  g = torch.Generator().manual_seed(5)
  w0 = [torch.randn(7, 5, generator=g), torch.randn(11, generator=g), torch.randn(3, 2, 2, generator=g)]
  grads = [[torch.randn(w.shape, generator=g) for w in w0] for _ in range(3)]
  ps = [torch.nn.Parameter(w.clone()) for w in w0]
  groups = [{'params': [ps[0]], 'lr': 1.0}, {'params': [ps[1]], 'lr': 2.0, 'weight_decay': 0.0},
            {'params': [ps[2]], 'lr': 10.0}]
  ref_sgd = ref_opt.SGD(groups, lr=1, momentum=0.9, weight_decay=5e-4)
  sgd_lrs = [3e-3, 1.7e-3, 2.9e-3]
  sgd_out = {}
  for i in range(3):
    for pth, gr in zip(ps, grads[i]):
      pth.grad = gr.clone()
    ref_sgd.step(sgd_lrs[i])
    for j, pth in enumerate(ps):
      sgd_out['w%d_after%d' % (j, i)] = pth.detach().clone()
  save(out, 'h01_sgd', lrs=np.array(sgd_lrs), **{'w%d' % j: w for j, w in enumerate(w0)},
       **{'g%d_%d' % (i, j): gr for i in range(3) for j, gr in enumerate(grads[i])}, **sgd_out)

  # ======================= N2 / N3: arithmetic blocks of the inference scripts ============
  # The scripts themselves cannot be imported (cv2, tensorboardX, hard-coded .cuda()), but the
  # blocks below are pure torch: their SOURCE LINES are read from the reference tree at run
  # time, dedented, stripped of the device moves and exec'd on seeded CPU inputs (nothing is
  # copied into the repository; only inputs and outputs are stored).
  import linecache
  import math
  import textwrap

  def ref_lines(path, first, last):
    txt = ''.join(linecache.getline(path, i) for i in range(first, last + 1))
    assert txt.strip(), path
    return textwrap.dedent(txt).replace('.to("cuda:0")', '').replace('.cuda()', '')

  # ---- N2: pyscripts/inference/prototype.py:134-205 (window ends, per-crop normalise +
  # overlap accumulation, division by the counts, full-image k-means, prototypes, majority
  # labels).  embedding_model: a seeded 5x5 conv as generate_embeddings, the reference's own
  # ResnetDeeplab.generate_clusters (on the CPU-shimmed segment_by_kmeans).
  import spml.models.embeddings.resnet_deeplab as e_dl2
  proto_py = os.path.join(args.ref, 'pyscripts', 'inference', 'prototype.py')
  src_n2 = ref_lines(proto_py, 134, 205)
  assert 'patch_ind_h' in src_n2 and 'find_majority_label_index' in src_n2

  class StubEmbedder:
    label_divisor = 2048
    semantic_ignore_index = 255
    kmeans_iterations = 10

    def __init__(self, conv, clusters):
      self.conv, self.kmeans_num_clusters = conv, clusters

    def generate_embeddings(self, datas, targets=None, resize_as_input=False):
      return {'embedding': self.conv(datas['image']), 'local_feature': None}

    generate_clusters = e_dl2.ResnetDeeplab.generate_clusters

  n2_store = {}
  orig_sbk2 = e_dl2.segsort_common.segment_by_kmeans
  e_dl2.segsort_common.segment_by_kmeans = ref_segment_by_kmeans
  try:
    for ci, (c, pad, valid, crop, stride, k) in enumerate([
        (16, (70, 90), (60, 83), (48, 48), (32, 32), (3, 3)),
        (8, (50, 50), (41, 50), (50, 50), (33, 33), (2, 2))]):
      gen = torch.Generator().manual_seed(1300 + ci)
      torch.manual_seed(1300 + ci)
      conv = torch.nn.Conv2d(3, c, 5, padding=2)
      base = torch.randn(1, 3, pad[0] // 8 + 2, pad[1] // 8 + 2, generator=gen)
      image = torch.nn.functional.interpolate(base, size=pad, mode='bilinear', align_corners=False)
      image = image + 0.05 * torch.randn(1, 3, pad[0], pad[1], generator=gen)
      sem = torch.randint(0, 5, (valid[0] // 10 + 1, valid[1] // 10 + 1), generator=gen)
      sem = sem.repeat_interleave(10, 0).repeat_interleave(10, 1)[:valid[0], :valid[1]].contiguous()
      fake = torch.full((1, pad[0], pad[1]), 255, dtype=torch.long)
      fake[:, :valid[0], :valid[1]] = 0                      # prototype.py:117-131
      env = {
          'config': AttrDict(test=AttrDict(stride=list(stride), crop_size=list(crop)),
                             network=AttrDict(label_divisor=2048)),
          'pad_image_h': pad[0], 'pad_image_w': pad[1], 'image_batch': {'image': image},
          'embedding_model': StubEmbedder(conv, list(k)), 'common_utils': g_common,
          'segsort_common': s_common, 'fake_label_batch': {'semantic_label': fake, 'instance_label': fake.clone()},
          'label_batch': {'semantic_label': sem.unsqueeze(0)}, 'math': math, 'np': np, 'torch': torch,
          'os': os, 'prototype_dir': '/nonexistent', 'base_name': 'x.png'}
      exec(compile(src_n2, proto_py + ':134-205', 'exec'), env)
      t = 'c%d_' % ci
      n2_store.update({
          t + 'image': image, t + 'sem': sem, t + 'conv_w': conv.weight, t + 'conv_b': conv.bias,
          t + 'cfg': np.array([c, pad[0], pad[1], valid[0], valid[1], crop[0], crop[1], stride[0],
                               stride[1], k[0], k[1]]),
          t + 'ends_h': env['patch_ind_h'], t + 'ends_w': env['patch_ind_w'],
          t + 'embedding': env['embeddings']['embedding'], t + 'counts': env['counts'],
          t + 'cluster_index': env['embeddings']['cluster_index'],
          t + 'prototypes': env['prototypes'], t + 'prototype_labels': env['prototype_labels']})
  finally:
    e_dl2.segsort_common.segment_by_kmeans = orig_sbk2
  save(out, 'n2_window', **n2_store)

  # ---- N3: pyscripts/inference/pseudo_camrw_crf.py:139-148 (per view: crop to the image,
  # un-flip, 1/8 bilinear, normalise, exp(5 cos - 5)) and :150-164 (mean over the views, CAM to
  # 1/8, 20th power, column normalisation, T <- T.T x WALK_STEPS, cam . T).
  rw_py = os.path.join(args.ref, 'pyscripts', 'inference', 'pseudo_camrw_crf.py')
  src_view = ref_lines(rw_py, 139, 148)
  src_walk = ref_lines(rw_py, 150, 164)
  assert 'exp_()' in src_view and 'WALK_STEPS' in src_walk
  walk_steps = None
  for i in range(20, 40):
    ln = linecache.getline(rw_py, i).strip()
    if ln.startswith('WALK_STEPS'):
      walk_steps = int(ln.split('=')[1])
  assert walk_steps == 6
  n3_store = {}
  for ci, (c, image_hw, pad_hw) in enumerate([(12, (56, 72), (64, 72)), (8, (32, 32), (32, 32))]):
    gen = torch.Generator().manual_seed(1400 + ci)
    image_h, image_w = image_hw
    views = []
    base = torch.randn(1, c, pad_hw[0] // 16 + 2, pad_hw[1] // 16 + 2, generator=gen)
    for flip in (False, True):
      e = torch.nn.functional.interpolate(base, size=pad_hw, mode='bilinear', align_corners=False)
      e = e + 0.2 * torch.randn(1, c, pad_hw[0], pad_hw[1], generator=gen)
      views.append((torch.flip(e, dims=[3]) if flip else e, flip))
    cam = torch.rand(21, image_h, image_w, generator=gen)
    env = {'torch': torch, 'F': torch.nn.functional, 'affs': [], 'image_h': image_h, 'image_w': image_w,
           'resize_image_h': image_h, 'resize_image_w': image_w, 'image_batch': None, 'label_batch': None,
           'WALK_STEPS': walk_steps, 'cam_full_arr': cam.clone()}
    for vi, (e, flip) in enumerate(views):
      env['embedding_model'] = lambda a, b, resize_as_input=True, _e=e: {'embedding': _e}
      env['data_info'] = {'is_flip': flip}
      exec(compile(src_view, rw_py + ':139-148', 'exec'), env)
      n3_store['c%d_view%d' % (ci, vi)] = e
      n3_store['c%d_flip%d' % (ci, vi)] = np.array(int(flip))
      n3_store['c%d_embs8_%d' % (ci, vi)] = env['embs']        # the 1/8-resolution unit embedding
    exec(compile(src_walk, rw_py + ':150-164', 'exec'), env)
    n3_store.update({'c%d_cam' % ci: cam, 'c%d_cam8' % ci: env['cam_full_arr'],
                     'c%d_trans' % ci: env['aff_mat'] / torch.sum(env['aff_mat'], dim=0, keepdim=True),
                     'c%d_cam_rw' % ci: env['cam_rw'], 'c%d_hw' % ci: np.array([image_h, image_w])})
  n3_store['walk_steps'] = np.array(walk_steps)
  save(out, 'n3_randomwalk', **n3_store)

  # ======================= N5 = N1 o N2: full-resolution kNN label inference ==============
  # pyscripts/inference/inference.py:162-227: window ends, per-crop normalise + overlap average,
  # k-means over the whole (padded) image with the padding ignored, Segsort.predictions against a
  # memory bank -> one label per un-padded pixel.  Same exec-the-lines arrangement as N2: a seeded
  # 5x5 conv as generate_embeddings, the reference's own generate_clusters and Segsort.
  inf_py = os.path.join(args.ref, 'pyscripts', 'inference', 'inference.py')
  src_n5 = ref_lines(inf_py, 162, 227)
  assert 'patch_ind_h' in src_n5 and 'with_prediction=True' in src_n5
  n5_store = {}
  e_dl2.segsort_common.segment_by_kmeans = ref_segment_by_kmeans
  try:
    for ci, (c, pad, valid, crop, stride, k, n_bank, n_cls) in enumerate([
        (16, (70, 90), (60, 83), (48, 48), (32, 32), (5, 5), 60, 5),
        (8, (50, 50), (41, 50), (50, 50), (33, 33), (6, 4), 33, 4)]):
      gen = torch.Generator().manual_seed(1500 + ci)
      torch.manual_seed(1500 + ci)
      conv = torch.nn.Conv2d(3, c, 5, padding=2)
      base = torch.randn(1, 3, pad[0] // 8 + 2, pad[1] // 8 + 2, generator=gen)
      image = torch.nn.functional.interpolate(base, size=pad, mode='bilinear', align_corners=False)
      image = image + 0.05 * torch.randn(1, 3, pad[0], pad[1], generator=gen)
      fake = torch.full((1, pad[0], pad[1]), 255, dtype=torch.long)
      fake[:, :valid[0], :valid[1]] = 0                      # inference.py:145-156
      # a memory bank that looks like the image's own segments: normalised embeddings of random pixels
      with torch.no_grad():
        full = g_common.normalize_embedding(conv(image).permute(0, 2, 3, 1).reshape(-1, c))
      pick = torch.randint(0, full.shape[0], (n_bank,), generator=gen)
      bank = g_common.normalize_embedding(full[pick] + 0.1 * torch.randn(n_bank, c, generator=gen))
      bank_lab = torch.randint(0, n_cls, (n_bank,), generator=gen)
      env = {
          'config': AttrDict(test=AttrDict(stride=list(stride), crop_size=list(crop)),
                             network=AttrDict(label_divisor=2048)),
          'pad_image_h': pad[0], 'pad_image_w': pad[1], 'image_batch': {'image': image},
          'embedding_model': StubEmbedder(conv, list(k)), 'common_utils': g_common,
          'fake_label_batch': {'semantic_label': fake, 'instance_label': fake.clone()},
          'prediction_model': model, 'semantic_memory_prototypes': bank,
          'semantic_memory_prototype_labels': bank_lab, 'math': math, 'np': np, 'torch': torch}
      exec(compile(src_n5, inf_py + ':162-227', 'exec'), env)
      t = 'c%d_' % ci
      pred = env['outputs']['semantic_prediction']
      assert pred.numel() == valid[0] * valid[1]             # inference.py:233 views it as the un-padded image
      n5_store.update({
          t + 'image': image, t + 'conv_w': conv.weight, t + 'conv_b': conv.bias,
          t + 'cfg': np.array([c, pad[0], pad[1], valid[0], valid[1], crop[0], crop[1], stride[0],
                               stride[1], k[0], k[1]]),
          t + 'bank': bank, t + 'bank_lab': bank_lab,
          t + 'cluster_index': env['embeddings']['cluster_index'].to(torch.int16),
          t + 'semantic_prediction': pred.view(valid[0], valid[1]).to(torch.uint8),
          t + 'semantic_topk': env['outputs']['semantic_score'].to(torch.uint8)[::7].clone()})
  finally:
    e_dl2.segsort_common.segment_by_kmeans = orig_sbk2
  save(out, 'n5_inference', **n5_store)

  # ======================= H2: two steps of the stage-2 classifier training ===============
  # pyscripts/train/train_classifier.py:139-169, the loop body exec'd as it stands on ONE device:
  # the reference's ResnetDeeplab in eval mode under no_grad, its SoftmaxClassifier (dropout
  # p = 0: the GPU draws its mask from another generator) in train mode, ONE lib.nn.optimizer.SGD
  # over the groups of both models, poly lr.  DataParallel's calling convention (`model(*zip(...))`
  # -> list of per-device outputs, scatter_gather.gather) is stood in for by two lambdas.
  import spml.models.predictions.softmax_classifier as p_cls
  cls_py = os.path.join(args.ref, 'pyscripts', 'train', 'train_classifier.py')
  src_h2 = ref_lines(cls_py, 139, 169)
  assert 'prediction_model(*zip(embeddings, label_batch))' in src_h2 and 'optimizer.step(lr)' in src_h2
  cfg_h2 = AttrDict(
      train=AttrDict(base_lr=3e-3, max_iteration=4000, warmup_iteration=100, momentum=0.9,
                     weight_decay=5e-4, batch_size=2, lr_policy='poly'),
      dataset=AttrDict(semantic_ignore_index=255, num_classes=21),
      network=AttrDict(label_divisor=2048, embedding_dim=16, kmeans_num_clusters=[1, 1],
                       kmeans_iterations=0, use_syncbn=False, backbone_types='panoptic_deeplab_101'))
  c_emb = reinit_parameters(e_dl.ResnetDeeplab([1, 1, 1, 1], [1, 2, 1, 1], [1, 1, 2, 4], cfg_h2), 31)
  c_pred = reinit_parameters(p_cls.SoftmaxClassifier(cfg_h2), 33)
  c_pred.semantic_classifier[3].p = 0.0
  c_opt = ref_opt.SGD(c_emb.get_params_lr() + c_pred.get_params_lr(), lr=1,
                      momentum=cfg_h2.train.momentum, weight_decay=cfg_h2.train.weight_decay)
  c_opt.zero_grad()
  c_emb.eval()                                               # train_classifier.py:110-111
  c_pred.train()
  emb_sums_before = parameter_checksums(c_emb)[1]
  h2_store = {}
  orig_sbk3 = e_dl.segsort_common.segment_by_kmeans
  e_dl.segsort_common.segment_by_kmeans = ref_segment_by_kmeans
  try:
    for step in range(2):
      datas, targets = synth.make_batch(2, 161, seed=950 + step)
      env = {
          'torch': torch, 'config': cfg_h2, 'train_utils': g_train, 'optimizer': c_opt, 'curr_iter': 40 + step,
          'gpu_ids': ['cpu'], 'image_batch': [datas], 'label_batch': [dict(targets)],
          'embedding_model': lambda *pairs: [c_emb(*pr) for pr in pairs],
          'prediction_model': lambda *pairs: [c_pred(*pr) for pr in pairs],
          'scatter_gather': types.SimpleNamespace(
              gather=lambda outs, dev: {k: torch.stack([o[k] for o in outs]) if outs[0][k].dim() == 0
                                        else torch.cat([o[k] for o in outs], 0) for k in outs[0]})}
      exec(compile(src_h2, cls_py + ':139-169', 'exec'), env)
      names_p, sums_p = parameter_checksums(c_pred)
      t = 's%d_' % step
      h2_store.update({
          t + 'image_seed': np.array(950 + step), t + 'image_head': datas['image'].reshape(-1)[:64],
          t + 'image_sums': np.array([datas['image'].double().sum().item(), datas['image'].double().abs().sum().item()]),
          t + 'semantic_label': targets['semantic_label'].to(torch.int16),
          t + 'loss': env['loss'].detach(), t + 'accuracy': env['acc'].detach(), t + 'lr': np.array(env['lr']),
          t + 'pred_param_sums': sums_p,
          t + 'cls_w_head': dict(c_pred.named_parameters())['semantic_classifier.4.weight'].detach().reshape(-1)[:256].clone(),
          t + 'conv_w_head': dict(c_pred.named_parameters())['semantic_classifier.0.weight'].detach().reshape(-1)[:256].clone(),
          t + 'bn_running_mean': c_pred.semantic_classifier[1].running_mean.clone(),
          t + 'bn_running_var': c_pred.semantic_classifier[1].running_var.clone()})
  finally:
    e_dl.segsort_common.segment_by_kmeans = orig_sbk3
  # (the frozen network: not one parameter of it moved)
  assert torch.equal(parameter_checksums(c_emb)[1], emb_sums_before)
  h2_store['iter0'] = np.array(40)
  h2_store['pred_param_names'] = np.array(names_p)
  save(out, 'h02_classifier_step', **h2_store)

  # ======================= LR schedules ======================================
  its = np.arange(0, 30000, 37)
  save(out, 'h01_lr', its=its,
       poly=np.array([g_train.lr_poly(3e-3, int(i), 30000, 100) for i in its]),
       step=np.array([g_train.lr_step(3e-3, int(i), [20000, 25000], 100) for i in its]))


if __name__ == '__main__':
  main()
