#!/usr/bin/env python3
"""Micro-benchmark of the fused spherical-k-means pass (config R of SURVEY 8d)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import torch
from spml_amd import _ffi


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--side', type=int, default=513)
  ap.add_argument('--d', type=int, default=258)
  ap.add_argument('--k', type=int, default=6)
  ap.add_argument('--imgs', type=int, default=1)
  ap.add_argument('--iters', type=int, default=10)
  ap.add_argument('--reps', type=int, default=20)
  ap.add_argument('--warm-ms', type=float, default=50.0, help='GPU time of untimed calls in front of the timed ones')
  ap.add_argument('--flags', type=int, default=0, help='SPML_KMEANS_* bits (512: E-step passes on kmeans_pass16)')
  a = ap.parse_args()
  dev = 'cuda:0'
  g = torch.Generator(device=dev).manual_seed(235)
  p1 = a.side * a.side
  x = torch.randn(a.imgs * p1, a.d, device=dev, generator=g)
  # spatially coherent structure: add a smooth field
  yy = torch.linspace(0, 1, a.side, device=dev).view(-1, 1).expand(a.side, a.side).reshape(-1)
  xx = torch.linspace(0, 1, a.side, device=dev).view(1, -1).expand(a.side, a.side).reshape(-1)
  base = torch.randn(8, a.d, device=dev, generator=g)
  w = torch.stack([torch.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / 0.03)
                   for cy, cx in torch.rand(8, 2, generator=torch.Generator().manual_seed(1)).tolist()], 1)
  x = 0.3 * x + (w @ base).repeat(a.imgs, 1)
  x = x / x.norm(dim=1, keepdim=True)
  init = _ffi.kmeans_init_grid(a.side, a.side, a.k, a.k, dev).view(-1).repeat(a.imgs)
  off = (torch.arange(a.imgs + 1, device=dev) * p1).to(torch.int64)
  K = a.k * a.k
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(3):
    lab = _ffi.kmeans_run(x, off, p1, K, init, a.iters, flags=a.flags)
  e1.record()
  torch.cuda.synchronize()
  # ~50 ms of untimed calls straight in front of everything that is timed (no host synchronisation in between): the
  # firmware takes tens of ms of uninterrupted launches to settle the shader clock behind an idle gap
  # (profiles/r06_kmeans_clock.md); --warm-ms 0 gives the three-call warm-up of the earlier rounds
  n_warm = int(a.warm_ms / max(e0.elapsed_time(e1) / 3, 1e-3)) + 1 if a.warm_ms > 0 else 0

  def warm():
    for _ in range(n_warm):
      _ffi.kmeans_run(x, off, p1, K, init, a.iters, flags=a.flags)

  warm()
  e0.record()
  for _ in range(a.reps):
    lab = _ffi.kmeans_run(x, off, p1, K, init, a.iters, flags=a.flags)
  e1.record()
  torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / a.reps
  passes = a.iters + 1
  bytes_pass = x.numel() * 4 + x.shape[0] * 4
  path = _ffi.kmeans_last_path()
  out = {'path': path, 'P': x.shape[0], 'D': a.d, 'K': K,
         'ms_per_run': ms, 'us_per_iter': ms * 1e3 / a.iters, 'us_per_pass': ms * 1e3 / passes,
         'iters_per_s': a.iters / (ms * 1e-3),
         'GBps_per_pass': bytes_pass / (ms * 1e-3 / passes) / 1e9,
         'hbm_frac_8TB': bytes_pass / (ms * 1e-3 / passes) / 8e12}
  if path != 'mfma_f16x2_bigk' and path != 'generic':
    # per-launch durations of the pass kernels from their device time stamps
    warm()
    _, dur = _ffi.kmeans_run_profiled(x, off, p1, K, init, a.iters, flags=a.flags)
    fused = dur[1:-1]
    if path == 'mfma_f16x2_v4k':
      # an iteration = assign kernel + accumulate kernel: [seed M, (E, M) x (iters - 1), final E]
      e_us, m_us = dur[1:-1:2], dur[2:-1:2]
      out.update({'assign_pass_us_mean': round(e_us.mean().item(), 1) if e_us.numel() else None,
                  'accumulate_pass_us_mean': round(m_us.mean().item(), 1) if m_us.numel() else None})
      fused = e_us + m_us if e_us.numel() else fused
    out.update({'seed_pass_us': round(dur[0].item(), 1), 'final_pass_us': round(dur[-1].item(), 1),
                'fused_pass_us_mean': round(fused.mean().item(), 1) if fused.numel() else None,
                'fused_pass_us_each': [round(v, 1) for v in fused.tolist()],
                'frac_8TB': round((x.numel() * 4 + x.shape[0] * 8) / (fused.mean().item() * 1e-6) / 8e12, 4)
                if fused.numel() else None})
    if fused.numel():
      # matrix-core work of a fused pass: E-step 3 f16 MFMA terms (h*h', h*l', l*h') + M-step 2
      # (one-hot x hi, x lo) of 2*P*D*K flops each, on channels / clusters as padded by the tiles
      kpad = -(-K // 16) * 16
      dpad = -(-a.d // 32) * 32
      fl = 5 * 2.0 * x.shape[0] * dpad * kpad
      out['mfma_f16_tflops'] = round(fl / (fused.mean().item() * 1e-6) / 1e12, 1)
      out['mfma_frac_2500TF'] = round(fl / (fused.mean().item() * 1e-6) / 2.5e15, 4)
  else:
    # MFMA-bound E-step: 2*P*D*K*3 f16 flops per iteration (split-f16 = 3 MFMA passes)
    flops = 2.0 * x.shape[0] * a.d * K * 3
    out['f16_tflops_if_all_time_were_estep'] = flops / (ms * 1e-3 / a.iters) / 1e12
  out['hist'] = torch.bincount(lab, minlength=K).tolist()[:8]
  print(json.dumps(out))


if __name__ == '__main__':
  main()
