#!/usr/bin/env python3
"""One configuration of the NLL forward, a few launches (for counter passes): python tools/one_nll_fwd.py [P] [M] [random]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from spml_amd import _ffi
P = int(sys.argv[1]) if len(sys.argv) > 1 else 270400
M = int(sys.argv[2]) if len(sys.argv) > 2 else 139000
dev = 'cuda:0'
torch.manual_seed(0)
pr = torch.nn.functional.normalize(torch.randn(M, 64, device=dev), dim=1)
own = torch.randint(0, M, (P,), device=dev)
emb = torch.nn.functional.normalize(pr[own] + 0.8 * torch.randn(P, 64, device=dev), dim=1)
two = lambda n: (1 << torch.randint(0, 20, (n,), device=dev)) | (1 << torch.randint(0, 20, (n,), device=dev))
pc = two(M) if 'random' in sys.argv else two((M + 999) // 1000).repeat_interleave(1000)[:M]
order = torch.argsort(own // 1000, stable=True)
own, emb = own[order], emb[order]
px = pc[own]
for _ in range(3):
  nll, st = _ffi.segsort_nll_fwd(emb, own, px, pr, pc, 12.0, 5)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
  nll, st = _ffi.segsort_nll_fwd(emb, own, px, pr, pc, 12.0, 5)
e1.record(); torch.cuda.synchronize()
print('fwd %.3f ms  mean nll %.6f' % (e0.elapsed_time(e1) / 5, float(nll.mean())))
