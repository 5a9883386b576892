#!/usr/bin/env python3
"""Micro-benchmark of the top-k retrieval kernel (self-retrieval accuracy of the training step:
17 k prototypes on one GPU; 17 k queries x 139 k prototypes = one rank's share at 8 GPUs)."""
import sys, time, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from spml_amd import _ffi
torch.manual_seed(0)
for m, d, k in ((17000, 64, 5), (17000, 64, 20), (139000, 64, 5)):
  pr = torch.nn.functional.normalize(torch.randn(m, d, device='cuda'), dim=1)
  q = pr[:17000].contiguous()
  _ffi.topk_affinity(q, pr, k); torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(5): _ffi.topk_affinity(q, pr, k)
  torch.cuda.synchronize(); print(m, d, k, '%.3f ms' % ((time.perf_counter() - t0) / 5 * 1e3))
