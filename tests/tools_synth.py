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


