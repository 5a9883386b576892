#!/usr/bin/env python3
"""The per-image term's NLL call (one image of the bench step: 130 x 130 pixels, D = 66, ~380 (cluster, segment)
prototypes, label predicate, 32-bit codes): forward / backward time.   python tools/bench_nll_image.py [P] [M] [D]"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from spml_amd import _ffi
dev = 'cuda:0'
P = int(sys.argv[1]) if len(sys.argv) > 1 else 16900
M = int(sys.argv[2]) if len(sys.argv) > 2 else 384
D = int(sys.argv[3]) if len(sys.argv) > 3 else 66
torch.manual_seed(0)
pr = torch.nn.functional.normalize(torch.randn(M, D, device=dev), dim=1)
own = torch.randint(0, M, (P,), device=dev)
emb = torch.nn.functional.normalize(pr[own] + 0.8 * torch.randn(P, D, device=dev), dim=1)
pc = torch.randint(0, 300, (M,), device=dev)
px = pc[own]
g = torch.full((P,), 1.0 / P, device=dev)
def t(fn, n=20):
  fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(n): out = fn()
  torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6, out
f_us, (nll, st) = t(lambda: _ffi.segsort_nll_fwd(emb, own, px, pr, pc, 16.0, 4))
b_us, _ = t(lambda: _ffi.segsort_nll_bwd(emb, own, px, pr, pc, 16.0, 4, st, g))
print('P %d M %d D %d: forward %.1f us, backward %.1f us' % (P, M, D, f_us, b_us))
