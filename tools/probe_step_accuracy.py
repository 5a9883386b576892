#!/usr/bin/env python3
"""Where the GPU training step's floating-point error comes from (fp64 CPU run as the yardstick).

For the h01 model (tests/tools_synth.py) and one batch, with the fp32 oracle's clustering given to
every run: (1) embedding map GPU vs fp64; (2) d loss / d embedding: GPU vs the fp64 oracle evaluated
AT THE GPU'S embedding (isolates K1 / prototype / NLL / head backward kernels from the network's
forward rounding) and vs the fp64 run end to end; (3) parameter gradients vs fp64, next to the CPU
fp32 path's own error.  `python tools/probe_step_accuracy.py [--nhwc]`"""
import copy, os, sys
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle.cpu_step import CpuStep
from spml_amd.train import Trainer
from step_helpers import given_clustering, to_gpu
from tools_synth import h01_batch, h01_config, h01_models
from conftest import load_golden


def rel(a, b):
  a, b = a.double().cpu(), b.double().cpu()
  return ((a - b).norm() / b.norm().clamp(min=1e-300)).item()


def headline():
  """--config headline [--batch N]: the benchmarked configuration at full depth (tests/step_helpers.py,
  headline_depth_accuracy); prints the tables of profiles/r04_step_accuracy.md."""
  from step_helpers import headline_depth_accuracy
  batch = int(sys.argv[sys.argv.index('--batch') + 1]) if '--batch' in sys.argv else 4
  r = headline_depth_accuracy(batch=batch, verbose=lambda *a: print(*a, flush=True))
  print('\nResNet-101 DeepLab-v2, %d x 513 x 513, VOC12 scribble recipe; matrix-core units entered %d times' % (batch, r['mc_units']))
  print('\n| stage output | benchmarked path (NHWC, matrix-core units) vs fp64 | NCHW library path vs fp64 |\n|---|---|---|')
  for n, (ea, eb) in r['stages'].items():
    print('| %s | %.2e | %.2e |' % (n, ea, eb))
  print('\n| loss | GPU (benchmarked path) | fp64 oracle at the GPU embedding | difference |\n|---|---|---|---|')
  for n, (ga, gd) in r['losses'].items():
    print('| %s | %.7f | %.9f | %.1e |' % (n, ga, gd, abs(ga - gd)))
  print('\nd loss / d embedding, relative L2 against the fp64 oracle at the same embedding: benchmarked path %.2e, '
        'NCHW library path %.2e (its own embedding differs: includes the forward difference)' % (r['d_embedding'], r['d_embedding_b']))
  pg = r['param_grad']
  med = lambda v: sorted(v)[len(v) // 2]
  print('\nparameter gradients of the network (all three from the benchmarked path\'s d loss / d embedding), relative L2 vs fp64:')
  print('| | benchmarked path | NCHW library path |\n|---|---|---|')
  print('| median over %d tensors | %.2e | %.2e |' % (len(pg), med([t[1] for t in pg]), med([t[2] for t in pg])))
  print('| maximum | %.2e (%s) | %.2e |' % (pg[0][1], pg[0][0], max(t[2] for t in pg)))
  for n, ea, eb in pg[:6]:
    print('| %s | %.2e | %.2e |' % (n, ea, eb))


def main():
  if '--config' in sys.argv and sys.argv[sys.argv.index('--config') + 1] == 'headline':
    return headline()
  nhwc = '--nhwc' in sys.argv
  g = load_golden('h01_step_nodrop')
  cfg = h01_config()
  emb, pred = h01_models(cfg)
  pred.semantic_classifier[3].p = 0.0
  emb.train(); pred.train()
  e64, p64 = copy.deepcopy(emb).double(), copy.deepcopy(pred).double()
  e32, p32 = copy.deepcopy(emb), copy.deepcopy(pred)
  datas, targets = h01_batch(g, 0)
  c32 = CpuStep(e32, p32, cfg, None, softmax_head=True)
  l32, _, _ = c32.forward_losses(datas, targets); l32.backward()
  ids = c32.last['cluster_index']
  c64 = CpuStep(e64, p64, cfg, None, softmax_head=True); c64.given_cluster_index = ids
  l64, _, _ = c64.forward_losses({'image': datas['image'].double()}, targets); l64.backward()
  tr = Trainer(cfg, 'cuda:0', softmax_head=True, channels_last=nhwc, models=(emb, pred))
  tr.embedding_model.train(); tr.prediction_model.train()
  rec = {}
  seen = {}
  import spml_amd.utils.segsort.common as sc
  real = sc.segment_by_kmeans
  def spy(embeddings, *a, **k):
    seen['emb'] = embeddings.detach().clone()
    return real(embeddings, *a, **k)
  sc.segment_by_kmeans = spy
  with given_clustering([ids], rec):
    loss, out, _ = tr.forward_losses(*to_gpu(datas, targets, nhwc))
    loss.backward()
  sc.segment_by_kmeans = real
  print('layout', 'NHWC' if nhwc else 'NCHW')
  print('loss gpu %.7f cpu32 %.7f cpu64 %.9f' % (loss.item(), l32.item(), l64.item()))
  eg = seen['emb'].cpu()
  print('embedding: gpu vs fp64 %.3e | cpu32 vs fp64 %.3e' % (rel(eg, c64.last['embedding']), rel(c32.last['embedding'], c64.last['embedding'])))
  # fp64 oracle losses evaluated at the GPU's embedding
  class AtEmb(CpuStep):
    def __init__(self, *a, **k):
      super().__init__(*a, **k)
  e64b, p64b = copy.deepcopy(e64), copy.deepcopy(p64)
  given = eg.double().contiguous().requires_grad_(True)
  e64b.generate_embeddings = (lambda orig: (lambda d, *a, **k: dict(orig(d, *a, **k), embedding=given)))(e64b.generate_embeddings)
  c64b = CpuStep(e64b, p64b, cfg, None, softmax_head=True); c64b.given_cluster_index = ids
  l64b, _, _ = c64b.forward_losses({'image': datas['image'].double()}, targets); l64b.backward()
  dg = rec['d_embedding'][0].cpu()
  print('loss fp64 at the GPU embedding %.9f (gpu %.7f)' % (l64b.item(), loss.item()))
  print('dEmbedding: gpu vs fp64-at-gpu-embedding %.3e | gpu vs fp64 end-to-end %.3e | cpu32 vs fp64 %.3e' % (
      rel(dg, given.grad), rel(dg, c64.last['embedding'].grad), rel(c32.last['embedding'].grad, c64.last['embedding'].grad)))
  g64 = dict((n, p.grad) for n, p in e64.named_parameters())
  g32 = dict((n, p.grad) for n, p in e32.named_parameters())
  worst = []
  for n, p in tr.embedding_model.named_parameters():
    if p.grad is not None:
      worst.append((rel(p.grad, g64[n]), rel(g32[n], g64[n]), n))
  worst.sort(reverse=True)
  print('parameter gradients, rel L2 vs fp64: gpu | cpu32')
  for a, b, n in worst[:8]:
    print('  %.3e | %.3e  %s' % (a, b, n))
  print('  median gpu %.3e cpu32 %.3e' % (sorted(w[0] for w in worst)[len(worst) // 2], sorted(w[1] for w in worst)[len(worst) // 2]))


if __name__ == '__main__':
  main()
