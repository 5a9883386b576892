"""The Python mirror of the reference API (spml_amd.utils / spml_amd.models) on
the GPU against the golden vectors of the reference -- these read like the
reference's own call sites."""
import pytest
import torch

from conftest import load_golden
from oracle import spml_oracle as O
import spml_amd.models.utils as model_utils
import spml_amd.utils.general.common as gc
import spml_amd.utils.segsort.common as sc
import spml_amd.utils.segsort.eval as se
import spml_amd.utils.segsort.loss as sl
from spml_amd.config.default import make_config
from spml_amd.models.predictions.segsort import Segsort

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def g2d(g, *names):
  return [g[n].to(DEV) for n in names]


@pytest.mark.parametrize('tag', ['tiny', 'small', 'rank1'])
def test_segment_by_kmeans_matches_reference(tag):
  g = load_golden('a08_segment_' + tag)
  emb = g.emb.to(DEV).requires_grad_(True)
  o = sc.segment_by_kmeans(emb, g.labels.to(DEV), g.k.tolist(), local_features=g.loc.to(DEV),
                           ignore_index=g.ignore, iterations=10, shard_id=g.gpu)
  torch.testing.assert_close(o[0].detach().cpu(), g.o_emb, rtol=0, atol=1e-6)
  torch.testing.assert_close(o[1].detach().cpu(), g.o_embloc, rtol=0, atol=1e-6)
  assert torch.equal(o[2].cpu(), g.o_lab) and torch.equal(o[4].cpu(), g.o_bat)
  mism = (o[3].cpu() != g.o_clu).float().mean().item()
  assert mism < 0.01, mism
  # no labels, default location features and grid: 3 iterations
  d = sc.segment_by_kmeans(g.emb.to(DEV), None, g.k.tolist(), iterations=3, shard_id=g.gpu)
  torch.testing.assert_close(d[1].cpu(), g.d_embloc, rtol=0, atol=1e-6)
  assert torch.equal(d[4].cpu(), g.d_bat)
  assert (d[3].cpu() != g.d_clu).float().mean().item() < 0.01
  # gradient reaches the NCHW embedding through K1
  (o[0].sum() + o[1].sum()).backward()
  assert torch.isfinite(emb.grad).all() and emb.grad.abs().sum() > 0


@pytest.mark.parametrize('tag', ['tiny', 'small'])
def test_kmeans_prototype_and_nearest_functions(tag):
  g = load_golden('a06_kmeans_' + tag)
  emb, init = g2d(g, 'emb', 'init')
  p0 = sc.calculate_prototypes_from_labels(emb, init, g.k)
  torch.testing.assert_close(p0.cpu(), g.protos0, rtol=0, atol=2e-6)
  near = sc.find_nearest_prototypes(emb, g.protos0.to(DEV))
  sims = g.emb @ g.protos0.t()
  t2 = sims.topk(2, dim=1).values
  bad = (near.cpu() != g.nearest0)
  assert (t2[bad][:, 0] - t2[bad][:, 1]).numel() == 0 or (t2[bad][:, 0] - t2[bad][:, 1]).max() < 1e-5
  final = sc.kmeans_with_initial_labels(emb, init, g.k, g.iterations)
  assert (final.cpu() != g.final).float().mean().item() < 5e-3
  torch.testing.assert_close(gc.normalize_embedding(emb).cpu(), O.normalize_embedding(g.emb),
                             rtol=0, atol=1e-6)


def test_gather_clustering_and_update_prototypes_two_shards():
  g = load_golden('b01_gather')
  embs = [g['s%d_emb' % i].to(DEV).requires_grad_(True) for i in (0, 1)]
  emls = [g['s%d_embloc' % i].to(DEV).requires_grad_(True) for i in (0, 1)]
  r = model_utils.gather_clustering_and_update_prototypes(
      embs, emls, g2d(g, 's0_clu', 's1_clu'), g2d(g, 's0_bat', 's1_bat'),
      g2d(g, 's0_sem', 's1_sem'), g2d(g, 's0_ins', 's1_ins'))
  torch.testing.assert_close(r[0][0].detach().cpu(), g.protos, rtol=0, atol=2e-6)
  torch.testing.assert_close(r[1][0].detach().cpu(), g.protos_loc, rtol=0, atol=2e-6)
  assert torch.equal(r[2][0].cpu(), g.p_sem) and torch.equal(r[3][0].cpu(), g.p_ins)
  assert torch.equal(r[4][0].cpu(), g.p_bat)
  assert torch.equal(r[5][0].cpu(), g.s0_new_clu) and torch.equal(r[5][1].cpu(), g.s1_new_clu)
  ((r[0][0] * g.wgt.to(DEV)).sum() + (r[1][0] * g.wgt2.to(DEV)).sum()).backward()
  for i in (0, 1):
    torch.testing.assert_close(embs[i].grad.cpu(), g['s%d_d_emb' % i], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(emls[i].grad.cpu(), g['s%d_d_embloc' % i], rtol=1e-4, atol=1e-6)
  tags = model_utils.gather_and_update_datas([torch.ones(2, 4), torch.zeros(3, 4)])
  assert tags[0].shape == (5, 4) and tags[1] is tags[0]


def test_multiset_labels_by_nearest_neighbor():
  g = load_golden('b03_multiset')
  out = model_utils.gather_multiset_labels_per_batch_by_nearest_neighbor(
      *g2d(g, 'emb', 'protos', 'p_sem', 'bat', 'p_bat'), num_classes=21, top_k=3,
      threshold=g.threshold)
  assert (out.cpu() != g.out).float().mean().item() < 2e-3


def test_top_k_ranking_and_majority():
  g = load_golden('a11_topk')
  acc, top = se.top_k_ranking(*g2d(g, 'q', 'ql', 'pr', 'prl'), 5)
  assert torch.equal(top.cpu(), g.top5) and abs(acc.item() - g.acc5) < 1e-6
  acc, top = se.top_k_ranking(*g2d(g, 'pr', 'prl', 'pr', 'prl'), 5)
  assert torch.equal(top.cpu(), g.top_self) and abs(acc.item() - g.acc_self) < 1e-6
  _, top20 = se.top_k_ranking(*g2d(g, 'q', 'ql', 'pr', 'prl'), 20)
  assert torch.equal(se.majority_label_from_topk(top20).cpu(), g.major20)


def test_label_algebra_on_gpu_tensors():
  """A7 / A13 / A14 on GPU tensors against the reference goldens: prepare_prototype_labels,
  find_majority_label_index, one_hot, resize_labels -- exact."""
  import spml_amd.utils.general.common as gc
  import spml_amd.utils.segsort.common as sc
  g = load_golden('a07_labels')
  pl, inv = sc.prepare_prototype_labels(g.sem2.to(DEV), g.ins2.to(DEV), g.off2)
  assert pl.is_cuda and torch.equal(pl.cpu(), g.plab2) and torch.equal(inv.cpu(), g.inv2)
  sel, major = sc.find_majority_label_index(g.sem2.to(DEV), g.ins2.to(DEV))
  assert sel.is_cuda and torch.equal(sel.cpu(), g.major_sel) and torch.equal(major.cpu(), g.major_lab)
  g = load_golden('a13_onehot_resize')
  assert torch.equal(gc.one_hot(g.lab.to(DEV)).cpu(), g.onehot)
  assert torch.equal(gc.one_hot(g.lab.to(DEV), 12).cpu(), g.onehot12)
  src = load_golden('a13_resize_src').src.to(DEV)
  for s in (17, 33, 130):
    out = gc.resize_labels(src, (s, s))
    assert out.is_cuda and torch.equal(out.cpu(), g['resized_%d' % s])


def test_segsort_predictions_nearest_neighbour_retrieval():
  """N1: `Segsort.predictions` (segsort.py:68-125) with a prototype memory bank."""
  g = load_golden('n1_predictions')
  model = Segsort(_cfg())
  pred, topk = model.predictions(
      {'cluster_embedding': g.emb.to(DEV), 'cluster_index': g.clu.to(DEV)},
      {'semantic_memory_prototype': g.bank.to(DEV),
       'semantic_memory_prototype_label': g.bank_lab.to(DEV)})
  assert torch.equal(pred.cpu(), g.pred) and torch.equal(topk.cpu(), g.topk)
  assert model.predictions({'cluster_embedding': g.emb.to(DEV)}, {}) == (None, None)


def _cfg(**train):
  base = dict(sem_ann_loss_types='segsort', sem_occ_loss_types='segsort',
              img_sim_loss_types='segsort', feat_aff_loss_types='none',
              sem_ann_concentration=6.0, sem_occ_concentration=12.0, img_sim_concentration=16.0,
              feat_aff_concentration=0.0, sem_ann_loss_weight=1.0, sem_occ_loss_weight=0.5,
              img_sim_loss_weight=0.1, feat_aff_loss_weight=0.0)
  base.update(train)
  return make_config(train=base, dataset=dict(num_classes=21, semantic_ignore_index=255),
                     network=dict(label_divisor=2048))


def _f01_inputs(g, with_memory=True):
  e = g.emb.to(DEV).requires_grad_(True)
  el = g.embloc.to(DEV).requires_grad_(True)
  datas = {'cluster_index': g.clu.to(DEV), 'cluster_embedding': e,
           'cluster_embedding_with_loc': el, 'cluster_semantic_label': g.sem.to(DEV),
           'cluster_instance_label': g.ins.to(DEV), 'cluster_batch_index': g.bat.to(DEV)}
  targets = {'prototype': g.protos.to(DEV), 'prototype_semantic_label': g.p_sem.to(DEV),
             'prototype_batch_index': g.p_bat.to(DEV), 'semantic_tag': g.sem_tag.to(DEV),
             'prototype_semantic_tag': g.sem_tag[g.p_bat].to(DEV)}
  if with_memory:
    targets.update({'memory_prototype': [g.mem_protos.to(DEV)],
                    'memory_prototype_semantic_label': [g.mem_p_sem.to(DEV)],
                    'memory_prototype_batch_index': [g.mem_p_bat.to(DEV)],
                    'memory_prototype_semantic_tag': [g.mem_tag.to(DEV)]})
  return e, el, datas, targets


def test_segsort_predictor_losses_match_reference():
  """Segsort.losses on the golden inputs of the reference's own Segsort.losses
  (three contrastive terms + accuracy, with and without the memory bank, grads)."""
  g = load_golden('f01_segsort_losses')
  model = Segsort(_cfg()).to(DEV)
  e, el, datas, targets = _f01_inputs(g)
  la, lo, li, acc = model.losses(datas, targets)
  for got, want in ((la, g.l_ann), (lo, g.l_occ), (li, g.l_img), (acc, g.acc)):
    assert abs(got.item() - want) <= 1e-4 * max(1.0, abs(want)), (got.item(), want)
  (la + lo + li).backward()
  for got, want in ((e.grad.cpu(), g.d_emb), (el.grad.cpu(), g.d_embloc)):
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() <= 1e-3 * scale
    assert (got - want).abs().mean().item() <= 2e-5 * scale
  e, el, datas, targets = _f01_inputs(g, with_memory=False)
  la, lo, li, acc = model.losses(datas, targets)
  for got, want in ((la, g.n_ann), (lo, g.n_occ), (li, g.n_img), (acc, g.n_acc)):
    assert abs(got.item() - want) <= 1e-4 * max(1.0, abs(want)), (got.item(), want)
  with pytest.raises(KeyError):
    Segsort(_cfg(sem_ann_loss_types='bogus'))


def test_feature_affinity_term_is_set_segsort_over_propagated_tags():
  """feat_aff = SetSegSortLoss o gather_multiset_labels (densepose predictor)."""
  g = load_golden('f01_segsort_losses')
  b = load_golden('b01_gather')
  cfg = _cfg(feat_aff_loss_types='segsort', feat_aff_concentration=12.0, feat_aff_loss_weight=0.5)
  # as in the reference, the keys alone do not add a term ...
  assert Segsort(cfg).feat_aff_set_loss is None
  cfg.train.evaluate_feat_aff = True        # ... the explicit opt-in does
  model = Segsort(cfg).to(DEV)
  e, el, datas, targets = _f01_inputs(g, with_memory=False)
  # prototypes with location for shard 0 (recomputed with the oracle)
  r = O.gather_clustering_and_update_prototypes([b.s0_emb], [b.s0_embloc], [b.s0_clu],
                                                [b.s0_bat], [b.s0_sem], [b.s0_ins])
  targets['prototype_with_loc'] = r[1][0].to(DEV)
  out = model(datas, targets)
  tags = O.gather_multiset_labels_per_batch_by_nearest_neighbor(
      r[1][0], r[1][0], g.p_sem, g.p_bat, g.p_bat, num_classes=21, top_k=1, threshold=0.95)
  tags = tags.masked_fill((tags.max(1, keepdim=True)[0] == 0).expand(-1, 21), 1)
  want = O.set_segsort_loss(g.emb, tags[g.clu], g.clu, g.protos, tags, 12.0) * 0.5
  assert abs(out['feat_aff_loss'].item() - want.item()) <= 1e-4 * max(1.0, abs(want.item()))


def test_set_segsort_loss_with_150_classes():
  """SetSegSortLoss with multi-hot tags over 150 classes (a 64-bit word holds 63): the classes present
  on both sides are remapped to bit positions per call; per-pixel NLL and gradients vs the oracle."""
  gen = torch.Generator().manual_seed(11)
  p, m, d, t = 2500, 400, 32, 150
  protos = O.normalize_embedding(torch.randn(m, d, generator=gen))
  own = torch.randint(0, m, (p,), generator=gen)
  emb = O.normalize_embedding(protos[own] + 0.7 * torch.randn(p, d, generator=gen))
  present = torch.randperm(t, generator=gen)[:48]
  p_tags = torch.zeros(m, t, dtype=torch.long)
  p_tags[torch.arange(m).repeat_interleave(2), present[torch.randint(0, 48, (2 * m,), generator=gen)]] = 1
  tags = p_tags[own]
  e, pr = emb.clone().requires_grad_(True), protos.clone().requires_grad_(True)
  want = O.set_segsort_nll(e, tags, own, pr, p_tags, 10.0).view(-1)
  want.mean().backward()
  eg, pg = emb.to(DEV).requires_grad_(True), protos.to(DEV).requires_grad_(True)
  got = sl.SetSegSortLoss(10.0, reduction='none')(eg, tags.to(DEV), own.to(DEV), pg, p_tags.to(DEV)).view(-1)
  got.mean().backward()
  rel = (got.detach().cpu() - want.detach()).abs() / want.detach().abs().clamp(min=1.0)
  assert (rel > 2e-5).float().mean().item() <= 5e-3 and rel.max().item() < 5e-3
  for g_, w_ in ((eg.grad.cpu(), e.grad), (pg.grad.cpu(), pr.grad)):
    assert (g_ - w_).abs().max().item() <= 1e-3 * w_.abs().max().item()
    assert (g_ - w_).abs().mean().item() <= 2e-5 * w_.abs().max().item()


def test_loss_modules_reductions_and_modes():
  g = load_golden('a09_loss_tiny')
  emb, sem, own, protos, p_sem = g2d(g, 'emb', 'sem', 'own', 'protos', 'p_sem')
  none = sl.SegSortLoss(g.kappa, reduction='none')(emb, sem, own, protos, p_sem)
  assert none.shape == (emb.shape[0], 1)
  mean = sl.SegSortLoss(g.kappa)(emb, sem, own, protos, p_sem)
  tot = sl.SegSortLoss(g.kappa, reduction='sum')(emb, sem, own, protos, p_sem)
  assert abs(mean.item() - g.loss) < 1e-5 and abs(tot.item() - g.loss * emb.shape[0]) < 1e-2
  # multi-hot tags as in the reference, or pre-packed sets
  tags, p_tags = g2d(g, 'tags', 'p_tags')
  a = sl.SetSegSortLoss(g.kappa)(emb, tags, own, protos, p_tags)
  b2 = sl.SetSegSortLoss(g.kappa)(emb, sl.pack_tag_sets(tags), own, protos, sl.pack_tag_sets(p_tags))
  assert abs(a.item() - g.set_loss) < 1e-5 and a.item() == b2.item()
  # group_mode other than 'segsort+': numerator is the own-segment similarity (loss.py:71-72)
  plain = sl.SegSortLoss(g.kappa, group_mode='segsort')(emb, sem, own, protos, p_sem)
  sim = ((g.emb @ g.protos.t()) * g.kappa).exp()
  num = sim.gather(1, g.own.view(-1, 1))
  den = (sim * (g.sem.view(-1, 1) != g.p_sem.view(1, -1)).float()).sum(1, keepdim=True) + num
  assert abs(plain.item() - (-(num / den).log()).mean().item()) < 1e-5
  assert 'SegSortLoss(concentration=' in repr(sl.SegSortLoss(6))


def _densepose_cfg():
  return make_config(
      train=dict(sem_ann_loss_types='segsort', sem_occ_loss_types='segsort',
                 img_sim_loss_types='segsort', feat_aff_loss_types='none',
                 sem_ann_concentration=6.0, sem_occ_concentration=12.0, img_sim_concentration=16.0,
                 feat_aff_concentration=0.0, sem_ann_loss_weight=1.0, sem_occ_loss_weight=0.5,
                 img_sim_loss_weight=0.1, feat_aff_loss_weight=0.0),
      dataset=dict(num_classes=15, semantic_ignore_index=255),
      network=dict(label_divisor=2048, embedding_dim=16, kmeans_num_clusters=[3, 3],
                   kmeans_iterations=5, use_syncbn=False, backbone_types='panoptic_pspnet_101'))


def test_densepose_embedding_variant_matches_reference():
  """N4: 5-channel local features (location + smoothed colour) through K1 and the k-means
  (C+5 = 21 channels -> generic path; the MFMA path takes D = 32q + 5), then the
  embedding-with-local rebuilt from 0.1 x embedding (resnet_pspnet_densepose.py:90-160)."""
  from spml_amd.models.embeddings.local_model import LocationColorNetwork
  from spml_amd.models.embeddings.resnet_pspnet_densepose import ResnetPspnetDensepose
  g = load_golden('n4_densepose')
  lfn = LocationColorNetwork(use_color=True, use_location=True, norm_color=True, smooth_ksize=5).to(DEV)
  local = lfn(g.image.to(DEV), size=tuple(g.emb_map.shape[-2:]))
  torch.testing.assert_close(local.cpu(), g.local, rtol=1e-5, atol=1e-6)

  class Net(ResnetPspnetDensepose):
    def __init__(self):            # no backbone needed for the clustering half
      torch.nn.Module.__init__(self)
      self.label_divisor, self.semantic_ignore_index = 2048, 255
      self.kmeans_num_clusters, self.kmeans_iterations = [3, 3], 5
  out = Net().generate_clusters(g.emb_map.to(DEV), g.sem_map.to(DEV), g.ins_map.to(DEV),
                                g.local.to(DEV))
  assert out['cluster_embedding_with_loc'].shape[1] == 16 + 5
  assert torch.equal(out['cluster_semantic_label'].cpu(), g.o_sem)
  assert torch.equal(out['cluster_instance_label'].cpu(), g.o_ins)
  assert torch.equal(out['cluster_batch_index'].cpu(), g.o_bat)
  torch.testing.assert_close(out['cluster_embedding'].cpu(), g.o_emb, rtol=0, atol=2e-6)
  assert (out['cluster_index'].cpu() != g.o_clu).float().mean().item() < 2e-3
  torch.testing.assert_close(out['cluster_embedding_with_loc'].cpu(), g.o_embloc, rtol=0, atol=2e-6)


def test_densepose_predictor_losses_match_reference():
  """N4: `segsort_softmax_densepose.SegsortSoftmax.losses` -- CE head + SegSort with
  nearest-neighbour propagated tags in the co-occurrence slot, memory bank without tags,
  per-image term without location."""
  from spml_amd.models.predictions.segsort_softmax_densepose import SegsortSoftmaxDensepose
  g = load_golden('n4_densepose')
  model = SegsortSoftmaxDensepose(_densepose_cfg()).to(DEV).eval()
  with torch.no_grad():
    head = model.semantic_classifier
    head[0].weight.copy_(g.cls_w0); head[1].weight.copy_(g.cls_bn_w); head[1].bias.copy_(g.cls_bn_b)
    head[4].weight.copy_(g.cls_w4); head[4].bias.copy_(g.cls_b4)
  emb = g.emb.to(DEV).requires_grad_(True)
  datas = {'cluster_index': g.clu.to(DEV), 'cluster_embedding': emb,
           'cluster_embedding_with_loc': g.embloc.to(DEV), 'cluster_semantic_label': g.sem.to(DEV),
           'cluster_instance_label': g.ins.to(DEV), 'cluster_batch_index': g.bat.to(DEV),
           'embedding': g.fmap.to(DEV)}
  targets = {'prototype': g.protos.to(DEV), 'prototype_with_loc': g.protos_loc.to(DEV),
             'prototype_semantic_label': g.p_sem.to(DEV), 'prototype_batch_index': g.p_bat.to(DEV),
             'semantic_label': g.flab.to(DEV),
             'memory_prototype': [g.mem_protos.to(DEV)],
             'memory_prototype_with_loc': [g.mem_protos_loc.to(DEV)],
             'memory_prototype_semantic_label': [g.mem_p_sem.to(DEV)],
             'memory_prototype_batch_index': [g.mem_p_bat.to(DEV)]}
  l_ann, l_occ, l_img, acc = model.losses(datas, targets)
  for got, want in ((l_ann, g.l_ann), (l_occ, g.l_occ), (l_img, g.l_img), (acc, g.acc)):
    assert abs(float(got) - float(want)) <= 1e-4 * max(1.0, abs(float(want))), (float(got), float(want))
  (l_ann + l_occ + l_img).backward()
  scale = g.d_emb.abs().max().item()
  # (a handful of pixels sit on the reference's `sum - own` cancellation)
  assert (emb.grad.cpu() - g.d_emb).abs().max().item() <= 3e-3 * scale
  assert (emb.grad.cpu() - g.d_emb).abs().mean().item() <= 2e-5 * scale
