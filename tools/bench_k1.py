#!/usr/bin/env python3
"""Micro-benchmark of K1 (normalise + NCHW->rows + location concat) forward / backward at the
training batch (16 x 64 x 130 x 130) and the stress map (2 x 512 x 258 x 258): GB/s against the
algorithmic bytes (fwd: 4C read + 4C + 4(C+2) written per pixel; bwd: 4C + 4C + 4(C+2) read, 4C written)."""
import json, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from spml_amd import _ffi
dev = 'cuda:0'


def t_us(fn, n=20):
  fn(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3


for n, c, h, w, cl in ((16, 64, 130, 130, False), (16, 64, 130, 130, True), (2, 512, 258, 258, False),
                       (2, 512, 258, 258, True)):
  g = torch.Generator(device=dev).manual_seed(1)
  emb = torch.randn(n, c, h, w, device=dev, generator=g)
  if cl:                  # channels-last storage (what the NHWC backbone produces): row-wise kernels
    emb = emb.contiguous(memory_format=torch.channels_last)
  rows = n * h * w
  f_us = t_us(lambda: _ffi.normalize_concat_loc(emb, None, None, rows))
  g1 = torch.randn(rows, c, device=dev, generator=g)
  g2 = torch.randn(rows, c + 2, device=dev, generator=g)
  b_us = t_us(lambda: _ffi.normalize_concat_loc_bwd(emb, None, None, g1, g2))
  fb = rows * 4 * (3 * c + 2)
  bb = rows * 4 * (4 * c + 2)
  print(json.dumps({'shape': [n, c, h, w], 'layout': 'channels_last' if cl else 'nchw', 'fwd_us': round(f_us, 1), 'fwd_GBps': round(fb / f_us / 1e3, 1),
                    'fwd_frac_8TB': round(fb / f_us / 1e3 / 8000, 3), 'bwd_us': round(b_us, 1),
                    'bwd_GBps': round(bb / b_us / 1e3, 1), 'bwd_frac_8TB': round(bb / b_us / 1e3 / 8000, 3)}))
