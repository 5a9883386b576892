#!/usr/bin/env python3
"""Screened / incremental k-means (kmeans_inc.hip) against the fused-pass path (flag 128) and a
torch fp64 check of one M- and one E-step; timing of both on the roofline shape."""
import argparse, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from spml_amd import _ffi

INC = 128       # SPML_KMEANS_SCREENED_INCREMENTAL


def timed(fn, reps):
  for _ in range(2):
    fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(reps):
    fn()
  b.record()
  torch.cuda.synchronize()
  return a.elapsed_time(b) / reps


def check(x, off, p1, k, init, iters, n_img):
  lab_new, cen_new = _ffi.kmeans_run(x, off, p1, k, init, iters, want_centroids=True, flags=INC)
  path = _ffi.kmeans_last_path()
  lab_old, cen_old = _ffi.kmeans_run(x, off, p1, k, init, iters, want_centroids=True)
  path_old = _ffi.kmeans_last_path()
  res = {'iters': iters, 'path': path, 'path_old': path_old,
         'label_mismatch_vs_fused': (lab_new != lab_old).float().mean().item(),
         'cent_maxdiff_vs_fused': (cen_new - cen_old).abs().max().item()}
  # stepwise exactness of the LAST iteration: prototypes = M-step (fp64) of the labels after
  # iters - 1 iterations, labels = arg-max against them outside a 1e-5 margin
  prev = _ffi.kmeans_run(x, off, p1, k, init, iters - 1, flags=INC) if iters > 1 else init
  worst_c, bad, worst_margin = 0.0, 0, 0.0
  for b in range(n_img):
    lo, hi = int(off[b]), int(off[b + 1])
    if hi == lo:
      continue
    xb = x[lo:hi].double()
    sums = torch.zeros(k, x.shape[1], dtype=torch.float64, device=x.device).index_add_(0, prev[lo:hi], xb)
    pr = sums / sums.norm(dim=1, keepdim=True).clamp_min(1e-12)
    worst_c = max(worst_c, (cen_new[b].double() - pr).abs().max().item())
    sims = xb @ pr.t()
    t2 = sims.topk(2, dim=1).values
    margin = t2[:, 0] - t2[:, 1]
    wrong = lab_new[lo:hi] != sims.argmax(1)
    bad += int(wrong.sum())
    if wrong.any():
      worst_margin = max(worst_margin, margin[wrong].max().item())
  res.update({'cent_err_vs_fp64': worst_c, 'labels_off_fp64_argmax': bad, 'worst_margin_of_those': worst_margin})
  return res


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--side', type=int, default=513)
  ap.add_argument('--d', type=int, default=258)
  ap.add_argument('--k', type=int, default=6)
  ap.add_argument('--imgs', type=int, default=1)
  ap.add_argument('--reps', type=int, default=10)
  ap.add_argument('--noise', action='store_true', help='pure noise rows (the bench data)')
  a = ap.parse_args()
  dev = 'cuda:0'
  g = torch.Generator(device=dev).manual_seed(235)
  p1 = a.side * a.side
  x = torch.randn(a.imgs * p1, a.d, device=dev, generator=g)
  if not a.noise:
    yy = torch.linspace(0, 1, a.side, device=dev).view(-1, 1).expand(a.side, a.side).reshape(-1)
    xx = torch.linspace(0, 1, a.side, device=dev).view(1, -1).expand(a.side, a.side).reshape(-1)
    base = torch.randn(8, a.d, device=dev, generator=g)
    w = torch.stack([torch.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / 0.03)
                     for cy, cx in torch.rand(8, 2, generator=torch.Generator().manual_seed(1)).tolist()], 1)
    x = 0.3 * x + (w @ base).repeat(a.imgs, 1)
  x = (x / x.norm(dim=1, keepdim=True)).contiguous()
  init = _ffi.kmeans_init_grid(a.side, a.side, a.k, a.k, dev).view(-1).repeat(a.imgs)
  off = (torch.arange(a.imgs + 1, device=dev) * p1).to(torch.int64)
  K = a.k * a.k
  for iters in (2, 3, 10):
    print(json.dumps(check(x, off, p1, K, init, iters, a.imgs)), flush=True)
  l1 = _ffi.kmeans_run(x, off, p1, K, init, 10, flags=INC)
  l2 = _ffi.kmeans_run(x, off, p1, K, init, 10, flags=INC)
  print('deterministic:', bool(torch.equal(l1, l2)))
  ms_new = timed(lambda: _ffi.kmeans_run(x, off, p1, K, init, 10, flags=INC), a.reps)
  ms_old = timed(lambda: _ffi.kmeans_run(x, off, p1, K, init, 10), a.reps)
  print('10 iterations: screened+incremental %.1f us (%.0f it/s)   fused passes %.1f us (%.0f it/s)' % (
      ms_new * 1e3, 1e4 / ms_new, ms_old * 1e3, 1e4 / ms_old))


if __name__ == '__main__':
  main()
