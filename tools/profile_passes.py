#!/usr/bin/env python3
"""Per-pass kernel durations of one k-means call (device clocks) + run time: side d k imgs iters."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from spml_amd import _ffi
side, d, k, imgs, iters = [int(v) for v in sys.argv[1:6]]
dev = 'cuda:0'
g = torch.Generator(device=dev).manual_seed(235)
p1 = side * side
x = torch.nn.functional.normalize(torch.randn(imgs * p1, d, device=dev, generator=g), dim=1)
init = _ffi.kmeans_init_grid(side, side, k, k, dev).view(-1).repeat(imgs)
off = (torch.arange(imgs + 1, device=dev) * p1).to(torch.int64)
for _ in range(3):
  _ffi.kmeans_run(x, off, p1, k * k, init, iters)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
  _ffi.kmeans_run(x, off, p1, k * k, init, iters)
e1.record(); torch.cuda.synchronize()
durs = torch.stack([_ffi.kmeans_run_profiled(x, off, p1, k * k, init, iters)[1] for _ in range(5)]).mean(0)
print(json.dumps({'lib': os.environ.get('SPML_HIP_LIB', 'default'), 'us_per_iter': round(e0.elapsed_time(e1) / 10 / iters * 1e3, 2),
                  'passes_us': [round(v, 1) for v in durs.tolist()], 'fused_mean': round(durs[1:-1].mean().item(), 2)}))
