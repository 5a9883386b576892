#!/usr/bin/env python3
"""One k-means run (for in-kernel trace builds): side d k imgs iters."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from spml_amd import _ffi
side, d, k, imgs, iters = [int(v) for v in sys.argv[1:6]]
dev = 'cuda:0'
g = torch.Generator(device=dev).manual_seed(1)
p1 = side * side
x = torch.randn(imgs * p1, d, device=dev, generator=g)
if os.environ.get('STRUCT'):      # smooth field + noise (tools/bench_kmeans.py) instead of pure noise
  yy = torch.linspace(0, 1, side, device=dev).view(-1, 1).expand(side, side).reshape(-1)
  xx = torch.linspace(0, 1, side, device=dev).view(1, -1).expand(side, side).reshape(-1)
  base = torch.randn(8, d, device=dev, generator=g)
  w = torch.stack([torch.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / 0.03)
                   for cy, cx in torch.rand(8, 2, generator=torch.Generator().manual_seed(1)).tolist()], 1)
  x = float(os.environ.get('STRUCT')) * x + (w @ base).repeat(imgs, 1)
x = torch.nn.functional.normalize(x, dim=1)
init = _ffi.kmeans_init_grid(side, side, k, k, dev).view(-1).repeat(imgs)
off = (torch.arange(imgs + 1, device=dev) * p1).to(torch.int64)
for _ in range(2):
  _ffi.kmeans_run(x, off, p1, k * k, init, iters, flags=int(os.environ.get('KM_FLAGS', '0')))
torch.cuda.synchronize()
