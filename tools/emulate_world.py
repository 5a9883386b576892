#!/usr/bin/env python3
"""What ONE rank of a W-GPU job computes, measured on one GPU: the bench step with the prototype exchange replaced by
W copies of this rank's own prototypes (batch indices shifted per emulated rank, tags replicated), so that the
contrastive losses, the memory bank, the label algebra and the retrieval accuracy run at the W-rank prototype count
(M = W x local, plus W x the two-batch memory bank) -- everything a rank does except the collectives themselves
(SyncBatchNorm / DDP take the 1-rank RCCL path: SPML_FORCE_DISTRIBUTED=1).  The gathered copies carry no gradient
path to other ranks; the backward kernels still produce the gradient rows of ALL live prototypes (what a real rank
hands to the reduce-scatter).     python tools/emulate_world.py [W ...]      (default 1 2 4 8)

Measurement aid for the scaling prediction of DESIGN 7 (the 8-GPU runs are the driver's)."""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) >= 3 and sys.argv[1] == '--child':
  W = int(sys.argv[2])
  os.environ['SPML_FORCE_DISTRIBUTED'] = '1'
  sys.path.insert(0, ROOT)
  import torch
  import spml_amd.parallel as par

  def gather_prototypes(protos, protos_loc, proto_sem, proto_ins, proto_bat, cluster_indices):
    n_batch = int(os.environ.get('SPML_EMULATE_BATCH', '16'))
    rep = lambda x, grad: torch.cat([x] + [x.detach() if grad else x] * (W - 1), 0)
    bat = torch.cat([proto_bat + r * n_batch for r in range(W)], 0)
    return rep(protos, True), rep(protos_loc, True), rep(proto_sem, False), rep(proto_ins, False), bat, cluster_indices

  def gather_tags(semantic_tag):
    return torch.cat([semantic_tag] * W, 0)

  class _Dist:                                     # (only spml_amd.parallel sees W ranks; DDP / SyncBatchNorm see the real group)
    def __init__(self, real): self._real = real
    def __getattr__(self, name): return getattr(self._real, name)
    def get_world_size(self, *a, **k): return W
    def get_rank(self, *a, **k): return 0

  par.gather_prototypes = gather_prototypes
  par.gather_tags = gather_tags
  par.dist = _Dist(par.dist)
  sys.argv = ['bench.py', '--steps', '12', '--warmup', '6', '--no-cpu-baseline', '--no-kmeans']
  import bench
  bench.main()
  sys.exit(0)

worlds = [int(v) for v in sys.argv[1:]] or [1, 2, 4, 8]
rows = []
for w in worlds:
  r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', str(w)], capture_output=True, text=True, cwd=ROOT)
  line = [l for l in r.stdout.splitlines() if l.startswith('{')]
  if not line:
    print('W = %d failed:\n%s' % (w, (r.stdout + r.stderr)[-1500:]))
    continue
  d = json.loads(line[-1])
  rows.append((w, d['ms_per_step'], d['ms_each_step']))
  print('W = %d: %.2f ms per step  (each: %s)' % (w, d['ms_per_step'], d['ms_each_step']), flush=True)
if rows and rows[0][0] == 1:
  base = rows[0][1]
  for w, ms, _ in rows:
    print('W = %d: step x %.3f of W = 1 -> %.2f x before the collectives of %d ranks' % (w, ms / base, w * base / ms, w))
