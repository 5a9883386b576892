"""Seeded synthetic inputs shared by tests and bench."""
import torch


def coherent_rows(gen, n, d, side, noise=0.3, blobs=7):
  """n images of side*side unit-norm rows with spatially coherent structure."""
  dirs = torch.randn(blobs, d, generator=gen)
  cy = torch.rand(n, blobs, generator=gen)
  cx = torch.rand(n, blobs, generator=gen)
  yy = torch.linspace(0, 1, side).view(1, 1, side, 1)
  xx = torch.linspace(0, 1, side).view(1, 1, 1, side)
  wgt = torch.exp(-((yy - cy.view(n, blobs, 1, 1)) ** 2 + (xx - cx.view(n, blobs, 1, 1)) ** 2) / 0.03)
  e = torch.einsum('nbhw,bd->nhwd', wgt, dirs) + noise * torch.randn(n, side, side, d, generator=gen)
  e = e.reshape(n, side * side, d)
  return e / e.norm(dim=-1, keepdim=True).clamp(min=1e-12)


def reinit_parameters(module, seed):
  """Deterministic weights that do not depend on the order in which a model class creates
  its sub-modules: every tensor of the state dict is refilled from one seeded generator in
  sorted key order (He-scaled conv weights, BN weight 1 +- 0.1, small biases, unit running
  statistics).  Used on the REFERENCE model classes by tools/gen_golden.py (h01_step) and
  on this repository's classes by the tests, so that both start from identical weights."""
  gen = torch.Generator().manual_seed(seed)
  sd = module.state_dict()
  with torch.no_grad():
    for k in sorted(sd.keys()):
      v = sd[k]
      if not v.is_floating_point():
        continue
      if k.endswith('running_mean'):
        v.zero_()
      elif k.endswith('running_var'):
        v.fill_(1.0)
      elif v.dim() > 1:
        fan_in = v[0].numel()
        v.copy_(torch.randn(v.shape, generator=gen) * (2.0 / fan_in) ** 0.5)
      elif k.endswith('weight'):
        v.copy_(1.0 + 0.1 * torch.randn(v.shape, generator=gen))
      else:
        v.copy_(0.05 * torch.randn(v.shape, generator=gen))
  return module


def parameter_checksums(module):
  """(names, [sum, abs-sum, first element] per parameter) in sorted name order."""
  names = sorted(k for k, _ in module.named_parameters())
  pd = dict(module.named_parameters())
  vals = torch.stack([torch.stack([pd[k].detach().double().sum(), pd[k].detach().double().abs().sum(),
                                   pd[k].detach().double().reshape(-1)[0]]) for k in names])
  return names, vals


# ---- H1 fixtures (tests/golden/h01_step*.npz): config, models, batches ----
def h01_config():
  from spml_amd.train import voc12_scribble_config
  cfg = voc12_scribble_config(batch_size=2, crop=161, embedding_dim=16, kmeans=4,
                              memory_bank_size=2, max_iteration=30000, use_syncbn=False)
  cfg.network.kmeans_iterations = 5
  return cfg


def h01_models(cfg):
  from spml_amd.models.embeddings.resnet_deeplab import ResnetDeeplab
  from spml_amd.models.predictions.segsort_softmax import segsort
  from tools_synth import reinit_parameters
  emb = reinit_parameters(ResnetDeeplab([1, 1, 1, 1], [1, 2, 1, 1], [1, 1, 2, 4], cfg), 31)
  pred = reinit_parameters(segsort(cfg), 32)
  return emb, pred


def h01_batch(g, step):
  from spml_amd import synth
  datas, targets = synth.make_batch(2, 161, seed=int(g['s%d_image_seed' % step]))
  t = 's%d_' % step
  # the inputs are regenerated from the seed; the fixture pins them
  assert torch.equal(datas['image'].reshape(-1)[:64], g[t + 'image_head'])
  assert abs(datas['image'].double().sum().item() - g[t + 'image_sums'][0].item()) < 1e-6
  for k in ('semantic_label', 'instance_label', 'semantic_tag'):
    assert torch.equal(targets[k], g[t + k].long()), k
  return datas, targets




# ---- H2 fixture (tests/golden/h02_classifier_step.npz): stage-2 config, models, batches ----
def h02_config():
  from spml_amd.train import voc12_scribble_config
  cfg = voc12_scribble_config(batch_size=2, crop=161, embedding_dim=16, kmeans=1, max_iteration=4000,
                              use_syncbn=False)
  cfg.network.kmeans_iterations = 0
  cfg.network.prediction_types = 'softmax_classifier'
  return cfg


def h02_models(cfg):
  from spml_amd.models.embeddings.resnet_deeplab import ResnetDeeplab
  from spml_amd.models.predictions.softmax_classifier import softmax_classifier
  emb = reinit_parameters(ResnetDeeplab([1, 1, 1, 1], [1, 2, 1, 1], [1, 1, 2, 4], cfg), 31)
  pred = reinit_parameters(softmax_classifier(cfg), 33)
  pred.semantic_classifier[3].p = 0.0            # (the fixture was captured with dropout p = 0)
  return emb, pred


def h02_batch(g, step):
  from spml_amd import synth
  datas, targets = synth.make_batch(2, 161, seed=int(g['s%d_image_seed' % step]))
  t = 's%d_' % step
  assert torch.equal(datas['image'].reshape(-1)[:64], g[t + 'image_head'])
  assert abs(datas['image'].double().sum().item() - g[t + 'image_sums'][0].item()) < 1e-6
  assert torch.equal(targets['semantic_label'], g[t + 'semantic_label'].long())
  return datas, targets


def check_h02_step(g, it, out, pred, tol):
  """Loss, accuracy, classifier parameters and BN running statistics after step `it` against the fixture."""
  t = 's%d_' % it
  for k in ('loss', 'accuracy'):
    want = float(g[t + k])
    assert abs(float(out[k]) - want) <= tol * max(1.0, abs(want)), (it, k, float(out[k]), want)
  names, sums = parameter_checksums(pred)
  assert names == g.pred_param_names
  want = g[t + 'pred_param_sums']
  sums = sums.cpu()
  assert ((sums[:, 0] - want[:, 0]).abs() <= tol * want[:, 1] + tol).all()
  torch.testing.assert_close(sums[:, 1], want[:, 1], rtol=tol, atol=tol)
  pd = dict(pred.named_parameters())
  for key, name in (('cls_w_head', 'semantic_classifier.4.weight'), ('conv_w_head', 'semantic_classifier.0.weight')):
    got = pd[name].detach().reshape(-1)[:256].cpu()
    torch.testing.assert_close(got, g[t + key], rtol=0, atol=tol * float(g[t + key].abs().max()))
  bn = pred.semantic_classifier[1]
  torch.testing.assert_close(bn.running_mean.cpu(), g[t + 'bn_running_mean'], rtol=0,
                             atol=tol * max(1.0, float(g[t + 'bn_running_mean'].abs().max())))
  torch.testing.assert_close(bn.running_var.cpu(), g[t + 'bn_running_var'], rtol=10 * tol, atol=tol)
