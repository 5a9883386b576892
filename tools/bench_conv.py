"""Time the matrix-core convolutions against the framework (MIOpen) ones on the res4/res5 shapes
of the headline recipe (batch 16, 65x65 feature maps)."""
import argparse
import torch
import torch.nn.functional as F

import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import spml_amd
from spml_amd import _ffi


def timeit(fn, reps):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(reps):
    fn()
  b.record()
  torch.cuda.synchronize()
  return a.elapsed_time(b) / reps * 1e3


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--reps', type=int, default=10)
  ap.add_argument('--batch', type=int, default=16)
  ap.add_argument('--side', type=int, default=65)
  ap.add_argument('--no-lib', action='store_true')
  ap.add_argument('--narrow', action='store_true', help='the narrow-output shapes only: res3 units, ASPP head')
  args = ap.parse_args()
  n, h, w = args.batch, args.side, args.side
  shapes = [(1024, 256, 1, 1), (256, 256, 3, 2), (256, 1024, 1, 1), (2048, 512, 1, 1), (512, 512, 3, 4),
            (512, 2048, 1, 1), (512, 256, 1, 1), (1024, 512, 1, 1)]
  if args.narrow:
    shapes = [(512, 128, 1, 1), (128, 128, 3, 1), (128, 512, 1, 1)]
    # ASPP head of the 64-d embedding: sum of four dilated 3x3 convolutions 2048 -> 64, one 36-tap launch
    cin, cout, dils = 2048, 64, (6, 12, 18, 24)
    x = torch.randn(n, cin, h, w, device='cuda').clamp_min(0).contiguous(memory_format=torch.channels_last)
    ws = [(torch.randn(cout, cin, 3, 3, device='cuda') * 0.01).contiguous(memory_format=torch.channels_last) for _ in dils]
    bs = [torch.zeros(cout, device='cuda') for _ in dils]
    xa = _ffi.hl8_from_f32(x)
    flops = 2.0 * n * h * w * cin * cout * 36
    t_own = timeit(lambda: _ffi.conv_hl8_pyramid_forward(xa, ws, bs, dils, n, h, w), args.reps)
    def lib():
      out = None
      for wt, b, d in zip(ws, bs, dils):
        y = F.conv2d(x, wt, b, 1, d, d)
        out = y if out is None else out.add_(y)
      return out
    t_lib = timeit(lib, args.reps)
    print('ASPP fwd 2048->64 x4 dilations: own %7.1f us (%5.1f TFLOP/s fp32-equivalent, %4.2f of f16 MFMA peak)   library %7.1f us'
          % (t_own, flops / t_own / 1e6, 3 * flops / t_own / 1e6 / 2500., t_lib), flush=True)
    t_gemm = timeit(lambda: _ffi.conv_hl8_pyramid_forward_gemm(xa, ws, bs, dils, n, h, w), args.reps)
    print('ASPP fwd as one 1x1 convolution with 36 x 64 columns + tap gather: %7.1f us' % t_gemm, flush=True)
    dy = torch.randn(n, cout, h, w, device='cuda').contiguous(memory_format=torch.channels_last) * 1e-4
    dya = _ffi.hl8_from_f32(dy)
    t_own = timeit(lambda: _ffi.conv_wgrad_pyramid_hl8(dya, xa, n, h, w, dils), args.reps)
    def lib_w():
      return [torch.nn.grad.conv2d_weight(x, wt.shape, dy, padding=d, dilation=d) for wt, d in zip(ws, dils)]
    t_lib = timeit(lib_w, args.reps)
    print('ASPP wgrad 2048->64 x4 dilations: own %7.1f us (%5.1f TFLOP/s fp32-equivalent, %4.2f of f16 MFMA peak)   library %7.1f us'
          % (t_own, flops / t_own / 1e6, 3 * flops / t_own / 1e6 / 2500., t_lib), flush=True)
    if os.environ.get('ASPP_ONLY'):
      return
  for cin, cout, k, dil in shapes:
    x = torch.randn(n, cin, h, w, device='cuda').clamp_min(0).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, k, k, device='cuda') * 0.02).contiguous(memory_format=torch.channels_last)
    flops = 2.0 * n * h * w * cin * cout * k * k
    xa = _ffi.hl8_from_f32(x)
    wf, wtr = _ffi.hl8_weight(wt)
    t_own = timeit(lambda: _ffi.conv_hl8(xa, wf, n, h, w, k * k, dil), args.reps)
    line = 'fwd %4d->%4d k%d d%d: own %7.1f us (%5.1f TFLOP/s fp32-equivalent, %4.2f of f16 MFMA peak)' % (
        cin, cout, k, dil, t_own, flops / t_own / 1e6, 3 * flops / t_own / 1e6 / 2500.)
    if not args.no_lib:
      t_lib = timeit(lambda: F.conv2d(x, wt, padding=dil * (k // 2), dilation=dil), args.reps)
      line += '   library %7.1f us' % t_lib
    if _ffi.conv_wgrad_hl8_supported(cin, cout, k * k):
      dy = torch.randn(n, cout, h, w, device='cuda').contiguous(memory_format=torch.channels_last) * 1e-4
      dya = _ffi.hl8_from_f32(dy)
      t_dg = timeit(lambda: _ffi.conv_hl8(dya, wtr, n, h, w, k * k, dil), args.reps)
      t_wg = timeit(lambda: _ffi.conv_wgrad_hl8(dya, xa, n, h, w, k * k, dil), args.reps)
      line += '  dgrad %7.1f us  wgrad %7.1f us (%4.2f)' % (t_dg, t_wg, 3 * flops / t_wg / 1e6 / 2500.)
      if not args.no_lib:
        t_lw = timeit(lambda: torch.nn.grad.conv2d_weight(x, wt.shape, dy, padding=dil * (k // 2), dilation=dil), args.reps)
        line += ' library wgrad %7.1f us' % t_lw
    t_cv = timeit(lambda: _ffi.hl8_from_f32(x, bound=xa.bound), args.reps)
    print(line + '   (fp32->hl8 of the input alone: %.1f us)' % t_cv, flush=True)


if __name__ == '__main__':
  main()
