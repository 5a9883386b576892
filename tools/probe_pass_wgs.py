#!/usr/bin/env python3
"""Per-workgroup start / end stamps of the fused k-means passes (513x513x258, K = 36): where the time of a launch goes
that is not the tile loop -- dispatch ramp, spread of the workgroups' ends (tail), per-workgroup duration."""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from spml_amd import _ffi
from spml_amd._ffi import lib, ptr, check, workspace, stream_ptr
dev = 'cuda:0'
side, d, k, iters = 513, 258, 36, 10
p = side * side
g = torch.Generator(device=dev).manual_seed(1)
x = torch.nn.functional.normalize(torch.randn(p, d, device=dev, generator=g), dim=1)
init = _ffi.kmeans_init_grid(side, side, 6, 6, dev).view(-1)
off = torch.tensor([0, p], device=dev, dtype=torch.int64)
n_pass, wgs = ctypes.c_int(0), ctypes.c_int(0)
check(lib().spml_kmeans_profile_layout(p, d, k, 1, p, iters, ctypes.byref(n_pass), ctypes.byref(wgs)), 'layout')
for rep in range(3):
  clocks = torch.zeros((n_pass.value, wgs.value, 2), dtype=torch.int64, device=dev)
  labels = torch.empty((p,), dtype=torch.int64, device=dev)
  ws = workspace(lib().spml_kmeans_workspace_bytes(p, d, k, 1, p), dev)
  check(lib().spml_kmeans_run_profiled_f32(ptr(x, torch.float32), p, d, ptr(off, torch.int64), 1, p, k, ptr(init, torch.int64),
                                           iters, ptr(labels), 0, ptr(ws), ws.numel(), ptr(clocks), clocks.numel(), stream_ptr()), 'run')
torch.cuda.synchronize()
c = clocks.cpu().double() * 0.01          # us
print('passes %d, workgroups %d' % (n_pass.value, wgs.value))
for i in range(n_pass.value):
  s, e = c[i, :, 0], c[i, :, 1]
  ok = e > 0
  s, e = s[ok], e[ok]
  t0 = s.min()
  dur = e - s
  print('pass %2d: launch %.1f us | starts spread %.1f | wg duration min %.1f mean %.1f max %.1f | ends: first %.1f mean %.1f last %.1f (from the first start)'
        % (i, e.max() - t0, s.max() - t0, dur.min(), dur.mean(), dur.max(), e.min() - t0, e.mean() - t0, e.max() - t0))
import numpy as np
dur = (c[:, :, 1] - c[:, :, 0]).numpy()          # [pass][wg]
T = (p + 31) // 32
def _rng(g, G=512):                                   # kmeans.hip: tile_range
  H = G // 2; q = g if g < H else g - H
  s0, s1 = T * q // H, T * (q + 1) // H
  first = (s1 - s0 + 1) // 2
  return first if g < H else s1 - s0 - first
ntiles = np.array([_rng(g) for g in range(512)])
if os.environ.get('SPML_KMEANS_STRIDED') != '0':          # default: tiles g, g + 512, ...
  ntiles = np.array([T // 512 + (1 if g < T % 512 else 0) for g in range(512)])
fused = dur[2:9]
m = fused.mean(0)
print('tiles per workgroup: %s' % dict(zip(*np.unique(ntiles, return_counts=True))))
for nt in np.unique(ntiles):
  print('  %d tiles: mean duration %.2f us (min %.2f max %.2f)' % (nt, m[ntiles == nt].mean(), m[ntiles == nt].min(), m[ntiles == nt].max()))
print('correlation of the per-workgroup durations between consecutive fused passes: %.2f' % np.corrcoef(fused[2], fused[3])[0, 1])
print('per XCD (workgroup index mod 8): ' + ' '.join('%.1f' % m[x::8].mean() for x in range(8)))
print('per-tile time by workgroup (us): mean %.3f, std %.3f; slowest 16 workgroups: %s' % ((m / ntiles).mean(), (m / ntiles).std(), np.argsort(-m)[:16].tolist()))
print('first half of the grid vs second half: %.2f %.2f' % (m[:256].mean(), m[256:].mean()))
