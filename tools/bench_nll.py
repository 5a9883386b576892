#!/usr/bin/env python3
"""Micro-benchmark of the NLL kernels at the bench-step size (sem_occ term)."""
import sys, time, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from spml_amd import _ffi
dev = 'cuda:0'
torch.manual_seed(0)
P, M, D = int(sys.argv[1]) if len(sys.argv) > 1 else 246016, int(sys.argv[2]) if len(sys.argv) > 2 else 17000, 64
pr = torch.nn.functional.normalize(torch.randn(M, D, device=dev), dim=1)
own = torch.randint(0, M, (P,), device=dev)
emb = torch.nn.functional.normalize(pr[own] + 0.8 * torch.randn(P, D, device=dev), dim=1)
pc = torch.randint(0, 2 ** 20, (M,), device=dev); xc = pc[own]
g = torch.full((P,), 1.0 / P, device=dev)
def t(fn, n=3):
  fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(n): out = fn()
  torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, out
ms, (nll, stats) = t(lambda: _ffi.segsort_nll_fwd(emb, own, xc, pr, pc, 12.0, 1))
print('fwd %.2f ms  (%.1f TFLOP/s useful)' % (ms, 2.0 * P * M * D / ms / 1e9))
ms, _ = t(lambda: _ffi.segsort_nll_bwd(emb, own, xc, pr, pc, 12.0, 1, stats, g))
print('bwd %.2f ms  (%.1f TFLOP/s useful)' % (ms, 6.0 * P * M * D / ms / 1e9))
