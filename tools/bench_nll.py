#!/usr/bin/env python3
"""Micro-benchmark of the NLL kernels (sem_occ term: tag-set predicate) at the bench-step size and at
the prototype counts of larger jobs: M = 17 k (1 GPU incl. memory bank), 70 k (4 GPUs), 139 k (8 GPUs).
  python tools/bench_nll.py [P] [M ...]      (D = 64; --d 514 for the stress configuration)"""
import json, sys, time, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from spml_amd import _ffi
dev = 'cuda:0'
args = [a for a in sys.argv[1:] if not a.startswith('--')]
D = int(sys.argv[sys.argv.index('--d') + 1]) if '--d' in sys.argv else 64
if '--d' in sys.argv:
  args.remove(str(D))
P = int(args[0]) if args else 270400
Ms = [int(v) for v in args[1:]] or [17000, 70000, 139000]


def t(fn, n=3):
  fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(n): out = fn()
  torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, out


for M in Ms:
  torch.manual_seed(0)
  pr = torch.nn.functional.normalize(torch.randn(M, D, device=dev), dim=1)
  own = torch.randint(0, M, (P,), device=dev)
  emb = torch.nn.functional.normalize(pr[own] + 0.8 * torch.randn(P, D, device=dev), dim=1)
  # tag-set codes as the co-occurrence term sees them: prototypes are image-major and carry their image's
  # tag set (two classes out of 20: random 20-bit patterns would all intersect, i.e. no negatives and a zero
  # loss), ~1000 prototypes per image, pixels image-major as well (codes32 / codes64); codes32_random: an
  # independent code per prototype and pixel order (no tile is uniform -- the kernels' general predicate paths)
  two = lambda n: (1 << torch.randint(0, 20, (n,), device=dev)) | (1 << torch.randint(0, 20, (n,), device=dev))
  pc_img = two((M + 999) // 1000).repeat_interleave(1000)[:M]
  pc_rnd = two(M)
  order = torch.argsort(own // 1000, stable=True)
  own_img, emb_img = own[order], emb[order]
  g = torch.full((P,), 1.0 / P, device=dev)
  row = {'P': P, 'M': M, 'D': D}
  own_rnd, emb_rnd = own, emb
  for name, mode, pc in (('codes64', 1, pc_img), ('codes32', 1 | 4, pc_img), ('codes32_random', 1 | 4, pc_rnd)):
    own, emb = (own_rnd, emb_rnd) if name == 'codes32_random' else (own_img, emb_img)
    xc = pc[own]
    f_ms, (nll, stats) = t(lambda: _ffi.segsort_nll_fwd(emb, own, xc, pr, pc, 12.0, mode))
    b_ms, _ = t(lambda: _ffi.segsort_nll_bwd(emb, own, xc, pr, pc, 12.0, mode, stats, g))
    b3_ms, _ = t(lambda: _ffi.segsort_nll_bwd(emb, own, xc, pr, pc, 12.0, mode, stats, g, m_grad=M // 3))
    row[name] = {'fwd_ms': round(f_ms, 2), 'bwd_ms': round(b_ms, 2), 'bwd_live_third_ms': round(b3_ms, 2),
                 'fwd_Tpairs_per_s': round(P * M / f_ms / 1e9, 2)}
  print(json.dumps(row))
