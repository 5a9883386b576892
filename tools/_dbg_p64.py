import sys, os
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from spml_amd import _ffi
dev = 'cuda:0'
z = np.load(os.path.join(R, 'tests/golden/a06_kmeans_small.npz'))
print(list(z.keys()))
x = torch.from_numpy(z['emb']).contiguous().to(dev).contiguous(); init = torch.from_numpy(z['init']).contiguous().to(dev).contiguous()
n, d = x.shape; k = int(z['k'])
off = torch.tensor([0, n], dtype=torch.int64, device=dev)
for iters in (2, 3):
  a, ca = _ffi.kmeans_run(x, off, n, k, init, iters, flags=0, want_centroids=True)
  b, cb = _ffi.kmeans_run(x, off, n, k, init, iters, flags=512, want_centroids=True)
  bad = (a != b).nonzero().view(-1).cpu()
  print(iters, 'mismatches', bad.numel(), 'idx', bad.tolist()[:20], 'a', a[bad][:20].tolist(), 'b', b[bad][:20].tolist())
  print('  cent diff', (ca - cb).abs().max().item(), 'rows differing', ((ca - cb).abs().amax(-1) > 1e-6).nonzero().reshape(-1).tolist(),
        'zero rows a', (ca.abs().amax(-1) == 0).nonzero().tolist(), 'b', (cb.abs().amax(-1) == 0).nonzero().tolist())
  sim = x @ cb[0].t()
  print('  scores at bad (b cent):', [(sim[i, a[i]].item(), sim[i, b[i]].item()) for i in bad[:8].tolist()])

  dc = (ca - cb)[0]
  r = dc.abs().amax(-1).argmax().item()
  print('  worst row', r, 'channels off', (dc[r].abs() > 1e-6).nonzero().reshape(-1).tolist()[:40], dc[r][dc[r].abs() > 1e-6][:8].tolist())
  print('  counts a', torch.bincount(a, minlength=k).tolist())
print('---- single fused pass')
cent = torch.from_numpy(z['protos_per_iter'][0]).contiguous().to(dev).view(1, k, d).contiguous()
for fl in (0, 512):
  ws = _ffi.kmeans_workspace(x, off, n, k)
  _ffi.kmeans_preconvert(x, off, n, k, ws)
  lab, sums = _ffi.kmeans_fused_pass(x, off, n, cent, ws=ws, preconverted=True, flags=fl)
  torch.cuda.synchronize()
  print('flags', fl, 'lab range', lab.min().item(), lab.max().item(), 'sums finite', torch.isfinite(sums).all().item(), flush=True)
  big = (lab >= k).nonzero().reshape(-1)
  simx = x @ cent[0].t()
  print('   out-of-range labels at', big.tolist(), lab[big].tolist(), 'torch max score', simx[big].max(1).values.tolist(), 'argmax', simx[big].argmax(1).tolist(), 'idx%64', (big % 64).tolist())
  lab = lab.clamp(0, k - 1)
  sim = x @ cent[0].t()
  ref = sim.argmax(1)
  badl = (lab != ref).nonzero().reshape(-1)
  raw = torch.zeros(k, d, device=dev).index_add_(0, lab, x)
  ds = (sums[0] - raw).abs().amax(-1)
  print('flags', fl, _ffi.kmeans_last_path(), 'label mismatches vs torch', badl.numel(), badl.tolist()[:10],
        'sum rows off', (ds > 1e-4).nonzero().reshape(-1).tolist(), ds.max().item())
  rows = (ds > 1e-4).nonzero().reshape(-1).tolist()
  for r in rows[:4]:
    # which pixel explains the difference?
    diff = sums[0][r] - raw[r]
    cand = ((x - diff).abs().amax(-1) < 1e-4).nonzero().reshape(-1).tolist(), ((x + diff).abs().amax(-1) < 1e-4).nonzero().reshape(-1).tolist()
    print('   row', r, 'extra pixel', cand[0], 'missing pixel', cand[1], 'their labels', [lab[i].item() for i in cand[0] + cand[1]])
