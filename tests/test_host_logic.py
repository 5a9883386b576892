"""Host-side logic that needs no GPU: config, LR schedules, optimizer, label
algebra, checkpoint name mapping, parameter groups, synthetic data."""
import os
import sys
import warnings

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import spml_oracle as O
import spml_amd
import spml_amd.utils.general.common as gc
import spml_amd.utils.general.train as gt
import spml_amd.utils.segsort.common as sc
import spml_amd.utils.segsort.loss as sl
import spml_amd.utils.segsort.eval as se
from spml_amd.config import default as cfg_mod
from spml_amd.nn.optimizer import SGD
from spml_amd.train import Trainer, voc12_scribble_config, build_models
from spml_amd import synth


def test_lr_schedules_match_reference():
  g = load_golden('h01_lr')
  poly = [gt.lr_poly(3e-3, int(i), 30000, 100) for i in g.its.tolist()]
  step = [gt.lr_step(3e-3, int(i), [20000, 25000], 100) for i in g.its.tolist()]
  np.testing.assert_allclose(poly, g.poly.numpy(), rtol=1e-12)
  np.testing.assert_allclose(step, g.step.numpy(), rtol=1e-12)


def test_label_algebra_on_cpu_tensors():
  g = load_golden('a07_labels')
  pl, inv = sc.prepare_prototype_labels(g.sem2, g.ins2, g.off2)
  assert torch.equal(pl, g.plab2) and torch.equal(inv, g.inv2)
  sel, major = sc.find_majority_label_index(g.sem2, g.ins2)
  assert torch.equal(sel, g.major_sel) and torch.equal(major, g.major_lab)
  g = load_golden('a13_onehot_resize')
  assert torch.equal(gc.one_hot(g.lab), g.onehot)
  assert torch.equal(gc.one_hot(g.lab, 12), g.onehot12)
  src = load_golden('a13_resize_src').src
  for s in (17, 33, 130):
    assert torch.equal(gc.resize_labels(src, (s, s)), g['resized_%d' % s])
  g = load_golden('a03_init_grid')
  assert torch.equal(sc.initialize_cluster_labels([6, 6], (130, 130), 'cpu'), g.init_k6_130)
  g = load_golden('a02_location')
  assert torch.equal(sc.generate_location_features((33, 29), 'cpu', 'float'), g['float_33x29'])
  with pytest.raises(ValueError):
    sc.generate_location_features((3, 3), 'cpu', 'bogus')
  g = load_golden('a11_topk')
  assert torch.equal(se.majority_label_from_topk(g.top20, 21), g.major20_21)


def test_pack_tag_sets():
  tags = torch.tensor([[1, 0, 0, 1], [0, 0, 0, 0], [0, 1, 1, 0]])
  assert sl.pack_tag_sets(tags).tolist() == [9, 0, 6]
  a = (torch.rand(50, 20) < 0.2).long()
  b = (torch.rand(70, 20) < 0.2).long()
  want = (a.float() @ b.t().float()) > 0
  pa, pb = sl.pack_tag_sets(a), sl.pack_tag_sets(b)
  assert torch.equal((pa.view(-1, 1) & pb.view(1, -1)) != 0, want)
  with pytest.raises(ValueError):
    sl.pack_tag_sets(torch.zeros(2, 64, dtype=torch.long))


def test_pack_tag_set_pair_any_width():
  """Tag sets wider than 63 classes (segsort/loss.py:95-130 has no limit): only the classes present
  on both sides are packed; the predicate equals the reference's `mm(tags, proto_tags.t()) > 0`."""
  from spml_amd.utils.segsort.loss import pack_tag_set_pair
  gen = torch.Generator().manual_seed(3)
  p, m, t = 300, 120, 150
  present = torch.zeros(t, dtype=torch.long)
  present[torch.randperm(t, generator=gen)[:55]] = 1
  px = (torch.rand(p, t, generator=gen) < 0.03).long() * present
  pr = (torch.rand(m, t, generator=gen) < 0.03).long() * present
  a, b = pack_tag_set_pair(px, pr)
  want = (px.float() @ pr.t().float()) > 0
  assert torch.equal((a.view(-1, 1) & b.view(1, -1)) != 0, want)
  assert int(a.max()) < 2 ** 62 and want.any() and not want.all()
  narrow_a, narrow_b = pack_tag_set_pair(px[:, :40], pr[:, :40])        # <= 63 columns: plain packing
  assert torch.equal((narrow_a.view(-1, 1) & narrow_b.view(1, -1)) != 0, (px[:, :40].float() @ pr[:, :40].t().float()) > 0)
  dense = torch.ones(4, 70, dtype=torch.long)
  with pytest.raises(ValueError):                                        # > 63 classes shared in one call
    pack_tag_set_pair(dense, dense)


def test_config_defaults_update_and_cli(tmp_path):
  c = cfg_mod.make_config(train={'base_lr': '3e-3', 'weight_decay': '5e-4'},
                          network={'embedding_dim': 64})
  assert c.train.base_lr == 3e-3 and c.train.weight_decay == 5e-4
  assert c.network.embedding_dim == 64 and c.network.label_divisor == 255
  assert c.train.momentum == 0.9 and c.dataset.semantic_ignore_index == 255
  y = tmp_path / 'cfg.yaml'
  y.write_text('gpus: "0,1"\nnetwork:\n  embedding_dim: 32\ntrain:\n  base_lr: 1e-2\nextra: 5\n')
  from spml_amd.config import parse_args as pa
  args = pa.parse_args('t', ['--snapshot_dir', 's', '--cfg_path', str(y), '--label_divisor', '2048'])
  assert args.label_divisor == 2048 and cfg_mod.config.network.embedding_dim == 32
  assert cfg_mod.config.train.base_lr == 1e-2 and cfg_mod.config.extra == 5
  assert cfg_mod.config.gpus == '0,1'
  with pytest.raises(SystemExit):
    pa.parse_args('t', ['--cfg_path', str(y)])        # --snapshot_dir is required


def test_sgd_step_matches_reference_update_rule():
  torch.manual_seed(0)
  w = torch.randn(7, 5, requires_grad=True)
  b = torch.randn(5, requires_grad=True)
  opt = SGD([{'params': [w], 'lr': 10}, {'params': [b], 'lr': 20, 'weight_decay': 0}], lr=1,
            momentum=0.9, weight_decay=5e-4)
  w0, b0 = w.detach().clone(), b.detach().clone()
  bufw, bufb = torch.zeros_like(w0), torch.zeros_like(b0)
  for lr in (3e-4, 1e-3):
    loss = (w.sum() ** 2 + (b * b).sum())
    opt.zero_grad()
    loss.backward()
    gw, gb = w.grad.clone(), b.grad.clone()
    opt.step(lr)
    # lib/nn/optimizer.py:80-101
    bufw = 0.9 * bufw + 10 * lr * (gw + 5e-4 * w0)
    bufb = 0.9 * bufb + 20 * lr * gb
    w0, b0 = w0 - bufw, b0 - bufb
    torch.testing.assert_close(w.detach(), w0, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(b.detach(), b0, rtol=1e-6, atol=1e-7)


def test_model_names_groups_and_checkpoint_mapping():
  cfg = voc12_scribble_config(batch_size=2)
  emb, pred = build_models(cfg)
  assert abs(sum(p.numel() for p in emb.parameters()) - 47342784) == 0
  groups = emb.get_params_lr()
  assert [g['lr'] for g in groups] == [1, 2, 10, 20]
  assert groups[1]['weight_decay'] == 0 and groups[3]['weight_decay'] == 0
  in_groups = {id(p) for g in groups for p in g['params']}
  frozen = [n for n, p in emb.named_parameters() if id(p) not in in_groups]
  assert frozen and all(n.startswith(('resnet_backbone.conv1', 'resnet_backbone.res2')) for n in frozen)
  assert [g['lr'] for g in pred.get_params_lr()] == [10, 20]
  assert emb.name_mapping('layer3.4.conv2.weight') == 'resnet_backbone.res4.4.conv2.weight'
  assert emb.name_mapping('bn1.running_mean') == 'resnet_backbone.conv1.bn1.running_mean'
  assert emb.name_mapping('module.aspp.aspp_1.0.bias', resume=True) == 'aspp.aspp_1.0.bias'
  sd = {'module.' + k: v.clone() for k, v in emb.state_dict().items()}
  sd['module.not_there'] = torch.zeros(1)
  with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    emb.load_state_dict(sd, resume=True)
  assert any('unexpected key' in str(x.message) for x in w)
  with pytest.raises(KeyError):
    cfg.train.sem_ann_loss_types = 'bogus'
    build_models(cfg)


def test_install_as_spml_alias():
  spml_amd.install_as_spml()
  import spml.utils.segsort.common as ref_style
  import spml.utils.general.train as t
  assert ref_style.segment_by_kmeans is sc.segment_by_kmeans
  assert t.lr_poly is gt.lr_poly


def test_synthetic_batch_contract():
  d, t = synth.make_batch(2, 65, seed=3)
  assert d['image'].shape == (2, 3, 65, 65) and d['image'].dtype == torch.float32
  assert t['semantic_label'].dtype == torch.int64 and t['semantic_tag'].shape == (2, 256)
  sem = t['semantic_label']
  assert (sem[:, -24:, :] == 255).all() and ((sem < 21) | (sem == 254) | (sem == 255)).all()
  d2, t2 = synth.make_batch(2, 65, seed=3)
  assert torch.equal(d['image'], d2['image']) and torch.equal(t['instance_label'], t2['instance_label'])


def test_synthetic_batch_with_per_image_palettes():
  """palette = (1, 3): background + 1..3 object classes per image, tags = the classes that occur (VOC-like tag sets:
  list_tag_dataset.py:75-78) -- most image pairs share no object class, so the co-occurrence term has negatives;
  the default generator (all 21 classes everywhere, what the golden fixtures were made with) is unchanged."""
  d, t = synth.make_batch(16, 513, seed=235, palette=(1, 3))
  tag = t['semantic_tag'][:, :21]
  n_obj = tag[:, 1:].sum(1)
  assert (n_obj >= 1).all() and (n_obj <= 3).all()
  sem = t['semantic_label']
  assert ((sem < 21) | (sem == 254) | (sem == 255)).all()
  for b in range(16):                                   # labelled pixels only use the image's own palette
    present = sem[b][sem[b] < 21].unique()
    assert tag[b][present].all()
  share = (tag[:, 1:].float() @ tag[:, 1:].float().t() > 0).float()
  off = (share.sum() - 16) / (16 * 15)
  assert off < 0.5, float(off)                          # (the dense generator: every pair)
  d0, t0 = synth.make_batch(16, 513, seed=235)
  dense = t0['semantic_tag'][:, 1:21].float()
  assert ((dense @ dense.t()) > 0).float().mean() > 0.95
  d2, t2 = synth.make_batch(16, 513, seed=235, palette=(1, 3))
  assert torch.equal(t['semantic_label'], t2['semantic_label']) and torch.equal(d['image'], d2['image'])


@pytest.mark.skipif(not os.path.isdir('/root/reference/spml'), reason='reference tree not mounted')
def test_embedding_network_matches_reference_modules():
  """Same weights -> same embedding map as the reference's ResnetDeeplab (CPU)."""
  sys.dont_write_bytecode = True
  mods = {k: v for k, v in sys.modules.items() if k == 'spml' or k.startswith('spml.')}
  for k in mods:
    del sys.modules[k]
  sys.path.insert(0, '/root/reference')
  try:
    from spml.models.embeddings.resnet_deeplab import ResnetDeeplab as RefNet
    cfg = voc12_scribble_config(batch_size=1, embedding_dim=16)
    torch.manual_seed(1)
    ref = RefNet([1, 1, 1, 1], [1, 2, 1, 1], [1, 1, 2, 4], cfg).eval()
  finally:
    sys.path.remove('/root/reference')
    for k in [k for k in sys.modules if k == 'spml' or k.startswith('spml.')]:
      del sys.modules[k]
    sys.modules.update(mods)
  from spml_amd.models.embeddings.resnet_deeplab import ResnetDeeplab
  mine = ResnetDeeplab([1, 1, 1, 1], [1, 2, 1, 1], [1, 1, 2, 4], cfg).eval()
  assert list(mine.state_dict().keys()) == list(ref.state_dict().keys())
  torch.nn.Module.load_state_dict(mine, ref.state_dict())
  x = {'image': torch.randn(1, 3, 65, 65)}
  with torch.no_grad():
    a = ref.generate_embeddings(x)
    b = mine.generate_embeddings(x)
  torch.testing.assert_close(b['embedding'], a['embedding'], rtol=1e-5, atol=1e-6)
  torch.testing.assert_close(b['local_feature'], a['local_feature'], rtol=0, atol=0)
  assert b['embedding'].shape[-2:] == (18, 18)


def test_memory_bank_files_round_trip(tmp_path):
  """others.load_memory_banks reads the reference's fixture files; save_memory_bank
  writes the same format (prototype.py:207-211)."""
  import spml_amd.utils.segsort.others as so
  g = load_golden('n1_predictions')
  bank_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'n1_memory_bank')
  protos, labels = so.load_memory_banks(bank_dir)
  assert protos.dtype == torch.float32 and labels.dtype == torch.int64
  assert torch.equal(protos, g.loaded) and torch.equal(labels, g.loaded_lab)
  half = protos.shape[0] // 2
  so.save_memory_bank(str(tmp_path / 'b.npy'), protos[half:], labels[half:])
  so.save_memory_bank(str(tmp_path / 'a.npy'), protos[:half], labels[:half])
  p2, l2 = so.load_memory_banks(str(tmp_path))
  assert torch.equal(p2, protos) and torch.equal(l2, labels)
  d = np.load(str(tmp_path / 'a.npy'), allow_pickle=True).item()
  assert sorted(d.keys()) == ['prototype', 'prototype_label']
  with pytest.raises(AssertionError):
    so.load_memory_banks(str(tmp_path / 'missing'))


def test_sliding_window_ends_follow_the_reference_arithmetic():
  """prototype.py:134-142: ceil((pad - crop) / stride) + 1 windows, ends = linspace(crop,
  pad, n) truncated to int32; the oracle and the product agree and cover the image."""
  from spml_amd import inference
  for pad, crop, stride in [(513, 513, 342), (770, 513, 342), (1025, 513, 342), (700, 512, 340),
                            (90, 48, 32), (70, 48, 32), (50, 50, 33)]:
    a = inference.sliding_window_ends(pad, crop, stride)
    b = O.sliding_window_ends(pad, crop, stride)
    assert a.dtype == np.int32 and np.array_equal(a, b)
    assert a[0] == crop and a[-1] == pad and len(a) == -(-(pad - crop) // stride) + 1
    assert np.all(np.diff(a) <= stride) or len(a) == 1
  assert list(inference.sliding_window_ends(770, 513, 342)) == [513, 770]
  assert list(inference.sliding_window_ends(1025, 513, 342)) == [513, 769, 1025]


def _with_reference_modules(fn):
  """Run fn() with /root/reference's `spml` importable, then restore sys.modules."""
  sys.dont_write_bytecode = True
  mods = {k: v for k, v in sys.modules.items() if k == 'spml' or k.startswith('spml.')}
  for k in mods:
    del sys.modules[k]
  sys.path.insert(0, '/root/reference')
  try:
    return fn()
  finally:
    sys.path.remove('/root/reference')
    for k in [k for k in sys.modules if k == 'spml' or k.startswith('spml.')]:
      del sys.modules[k]
    sys.modules.update(mods)


@pytest.mark.skipif(not os.path.isdir('/root/reference/spml'), reason='reference tree not mounted')
def test_pspnet_and_softmax_classifier_match_reference_modules():
  """N4: same weights -> same outputs as the reference's ResnetPspnet (head,
  parameter names, LR groups) and SoftmaxClassifier (logits, loss, accuracy), on CPU."""
  cfg = voc12_scribble_config(batch_size=1, embedding_dim=16)

  def build():
    from spml.models.embeddings.resnet_pspnet import ResnetPspnet as RefNet
    from spml.models.predictions.softmax_classifier import SoftmaxClassifier as RefCls
    torch.manual_seed(2)
    return RefNet([1, 1, 1, 1], [1, 2, 1, 1], [1, 1, 2, 4], cfg).eval(), RefCls(cfg)
  ref, ref_cls = _with_reference_modules(build)
  from spml_amd.models.embeddings.resnet_pspnet import ResnetPspnet
  from spml_amd.models.predictions.softmax_classifier import SoftmaxClassifier
  mine = ResnetPspnet([1, 1, 1, 1], [1, 2, 1, 1], [1, 1, 2, 4], cfg).eval()
  assert list(mine.state_dict().keys()) == list(ref.state_dict().keys())
  torch.nn.Module.load_state_dict(mine, ref.state_dict())
  x = {'image': torch.randn(1, 3, 65, 65)}
  with torch.no_grad():
    a, b = ref.generate_embeddings(x, resize_as_input=True), mine.generate_embeddings(x, resize_as_input=True)
  torch.testing.assert_close(b['embedding'], a['embedding'], rtol=1e-5, atol=1e-6)
  torch.testing.assert_close(b['local_feature'], a['local_feature'], rtol=0, atol=0)
  names = {id(p): n for n, p in mine.named_parameters()}
  ref_names = {id(p): n for n, p in ref.named_parameters()}
  for g_mine, g_ref in zip(mine.get_params_lr(), ref.get_params_lr()):
    assert g_mine['lr'] == g_ref['lr'] and g_mine.get('weight_decay') == g_ref.get('weight_decay')
    assert [names[id(p)] for p in g_mine['params']] == [ref_names[id(p)] for p in g_ref['params']]
  assert mine.name_mapping('layer3.0.conv1.weight') == ref.name_mapping('layer3.0.conv1.weight')
  assert mine.name_mapping('module.pspp.1.bias', resume=True) == 'pspp.1.bias'

  cls = SoftmaxClassifier(cfg)
  assert list(cls.state_dict().keys()) == list(ref_cls.state_dict().keys())
  cls.load_state_dict(ref_cls.state_dict())
  cls.eval(); ref_cls.eval()
  emb = torch.randn(2, 16, 17, 17)
  lab = torch.randint(0, 23, (2, 33, 33))
  lab[0, :5] = 255
  a = ref_cls({'embedding': emb}, {'semantic_label': lab.clone()})
  b = cls({'embedding': emb}, {'semantic_label': lab.clone()})
  torch.testing.assert_close(b['semantic_logit'], a['semantic_logit'], rtol=1e-6, atol=1e-6)
  torch.testing.assert_close(b['sem_ann_loss'], a['sem_ann_loss'], rtol=1e-6, atol=1e-6)
  torch.testing.assert_close(b['accuracy'], a['accuracy'], rtol=0, atol=0)
  assert torch.equal(b['semantic_prediction'], a['semantic_prediction'])
  nolab = cls({'embedding': emb})
  assert nolab['sem_ann_loss'] is None and nolab['semantic_prediction'].shape == (2, 17, 17)
  for g_mine, g_ref in zip(cls.get_params_lr(), ref_cls.get_params_lr()):
    assert g_mine['lr'] == g_ref['lr'] and len(g_mine['params']) == len(g_ref['params'])


def test_build_models_knows_the_pspnet_backbones():
  cfg = voc12_scribble_config(batch_size=1, embedding_dim=8)
  cfg.network.backbone_types = 'panoptic_pspnet_50'
  emb, _ = build_models(cfg, softmax_head=False)
  assert type(emb).__name__ == 'ResnetPspnet' and hasattr(emb, 'pspp')
  emb, pred = build_models(cfg, recipe='densepose')
  assert type(emb).__name__ == 'ResnetPspnetDensepose' and type(pred).__name__ == 'SegsortSoftmaxDensepose'
  assert 'lfn.smooth_kernel.weight' in emb.state_dict()          # the colour blur, as in the reference
  cfg.network.backbone_types = 'nope'
  with pytest.raises(ValueError):
    build_models(cfg)
  with pytest.raises(ValueError):
    build_models(cfg, recipe='densepose')


def test_fixed_order_convolution_gradients_match_the_framework_in_fp64():
  """spml_amd/nn/conv.py (deterministic mode's stand-in for the framework convolutions that remain): the autograd
  function -- forward, data gradient, weight gradient as chunked GEMMs, bias gradient -- against nn.Conv2d in fp64 on
  CPU tensors: strided 1x1, dilated 3x3, strided 3x3, the pointwise product in both memory formats; the layout of the
  input is kept."""
  import torch.nn as nn
  from spml_amd.nn.conv import DetConv2d, _DetConv2dFn, _fixed_order_matmul_t, make_deterministic
  torch.manual_seed(0)
  a, b = torch.randn(7, 5000, dtype=torch.double), torch.randn(9, 5000, dtype=torch.double)
  torch.testing.assert_close(_fixed_order_matmul_t(a, b), a @ b.t(), rtol=1e-12, atol=1e-11)
  for cin, cout, k, s, p, d in [(8, 12, 1, 2, 0, 1), (8, 12, 3, 1, 2, 2), (6, 5, 3, 2, 1, 1), (4, 7, 1, 1, 0, 1)]:
    for channels_last in (False, True):
      conv = nn.Conv2d(cin, cout, k, s, p, d, bias=True).double()
      x = torch.randn(3, cin, 31, 33, dtype=torch.double)
      if channels_last:
        x = x.contiguous(memory_format=torch.channels_last)
      x.requires_grad_(True)
      y = conv(x)
      up = torch.randn_like(y)
      (y * up).sum().backward()
      want = (y.detach(), x.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone())
      x.grad = None
      conv.zero_grad()
      y2 = _DetConv2dFn.apply(x, conv.weight, conv.bias, conv.stride, conv.padding, conv.dilation)
      (y2 * up).sum().backward()
      for got, ref in zip((y2.detach(), x.grad, conv.weight.grad, conv.bias.grad), want):
        torch.testing.assert_close(got, ref, rtol=1e-11, atol=1e-11)
      if k == 1 and s == 1:
        assert y2.is_contiguous(memory_format=torch.channels_last) == channels_last or y2.is_contiguous() != channels_last
  net = nn.Sequential(nn.Conv2d(4, 4, 3), nn.BatchNorm2d(4), nn.Conv2d(4, 2, 1))
  net[2].weight.requires_grad_(False)                       # (frozen: left alone)
  assert make_deterministic(net) == 1 and type(net[0]) is DetConv2d and type(net[2]) is nn.Conv2d
  assert list(net.state_dict()) == list(nn.Sequential(nn.Conv2d(4, 4, 3), nn.BatchNorm2d(4), nn.Conv2d(4, 2, 1)).state_dict())
  torch.testing.assert_close(net[0](torch.ones(1, 4, 5, 5)), nn.Conv2d.forward(net[0], torch.ones(1, 4, 5, 5)))   # CPU: plain path
