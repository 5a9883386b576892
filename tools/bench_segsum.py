#!/usr/bin/env python3
"""Micro-benchmark of the segment-prototype kernel (scatter-sum + normalise)."""
import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from spml_amd import _ffi
torch.manual_seed(0)
for p, d, m in ((270400, 64, 5800), (16900, 66, 360), (66564, 514, 1024)):
  x = torch.randn(p, d, device='cuda')
  ids = torch.sort(torch.randint(0, m, (p,), device='cuda'))[0] if p > 20000 else torch.randint(0, m, (p // 40 + 1,), device='cuda').repeat_interleave(40)[:p].contiguous()
  _ffi.segment_sum_normalize(x, ids, m); torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(20): _ffi.segment_sum_normalize(x, ids, m)
  torch.cuda.synchronize(); print(p, d, m, '%.1f us' % ((time.perf_counter() - t0) / 20 * 1e6))
