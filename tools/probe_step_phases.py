"""Wall time of the phases of one training step from stream events (no profiler attached):
backbone + head forward | clustering + prototypes + losses | backward | optimizer.
Compared with the kernel time of the same windows in a rocprofv3 trace this shows how much of a
phase the GPU spends waiting for the host (data-dependent shapes -> host syncs, chains of tiny ops)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import spml_amd
from spml_amd import synth
from spml_amd.train import Trainer, voc12_scribble_config

dev = torch.device('cuda', 0)
cfg = voc12_scribble_config(batch_size=16, crop=513)
torch.manual_seed(235)
tr = Trainer(cfg, dev, softmax_head=True, channels_last=True)
bs = [synth.make_batch(16, 513, num_classes=cfg.dataset.num_classes, seed=235 + i, device=dev) for i in range(2)]
for d, _ in bs:
  d['image'] = d['image'].contiguous(memory_format=torch.channels_last)

marks = {}


def stamp(name):
  e = torch.cuda.Event(enable_timing=True)
  e.record()
  marks.setdefault(name, []).append(e)


model = tr.embedding_model.module if hasattr(tr.embedding_model, 'module') else tr.embedding_model
gen = model.generate_clusters


def generate_clusters(*a, **k):
  stamp('clusters')
  return gen(*a, **k)


model.generate_clusters = generate_clusters
fl = tr.forward_losses


def forward_losses(*a, **k):
  stamp('forward')
  out = fl(*a, **k)
  stamp('backward')
  return out


tr.forward_losses = forward_losses
opt_step = tr.optimizer.step


def step(*a, **k):
  stamp('optimizer')
  return opt_step(*a, **k)


tr.optimizer.step = step
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for i in range(3 + n):
  tr.step(*bs[i % 2])
stamp('forward')
torch.cuda.synchronize()
# host synchronisations of one step (torch's sync debug mode warns at every blocking call)
import collections, warnings
torch.cuda.set_sync_debug_mode('warn')
with warnings.catch_warnings(record=True) as caught:
  warnings.simplefilter('always')
  tr.step(*bs[0])
torch.cuda.set_sync_debug_mode('default')
sites = collections.Counter()
for wmsg in caught:
  if 'synchroniz' in str(wmsg.message).lower():
    sites['%s:%d' % (os.path.relpath(wmsg.filename, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')), wmsg.lineno)] += 1
print('host synchronisations in one step: %d' % sum(sites.values()))
for k, v in sites.most_common(12):
  print('  %3d  %s' % (v, k))
order = ['forward', 'clusters', 'backward', 'optimizer']
tot = dict((k, 0.0) for k in order)
for i in range(3, 3 + n):
  for j, k in enumerate(order):
    nxt = marks[order[j + 1]][i] if j + 1 < len(order) else marks['forward'][i + 1]
    tot[k] += marks[k][i].elapsed_time(nxt)
names = {'forward': 'backbone + head forward', 'clusters': 'clustering, prototypes, losses (forward)',
         'backward': 'backward', 'optimizer': 'optimizer step (+ zero_grad of the next step)'}
for k in order:
  print('%-48s %7.2f ms' % (names[k], tot[k] / n))
print('%-48s %7.2f ms' % ('step', sum(tot.values()) / n))
