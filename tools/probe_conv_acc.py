"""Accumulation error of the f16 MFMA chain: operands exactly representable in f16 (l = 0)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import torch.nn.functional as F
from spml_amd import _ffi

for cin in (256, 1024, 2304, 4608):
  for positive in (False, True):
    g = torch.Generator().manual_seed(cin)
    x = torch.randn(2, cin, 16, 16, generator=g)
    if positive:
      x = x.clamp_min(0)
    x = x.half().float().cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(256, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5).half().float().cuda()
    ref = F.conv2d(x.double(), w.double())
    lib = F.conv2d(x, w)
    wf, _ = _ffi.hl8_weight(w)
    got = _ffi.conv_hl8(_ffi.hl8_from_f32(x), wf, 2, 16, 16, 1)
    d = (got.double() - ref)
    print('K=%5d positive=%d  own max %.2e mean(signed) %.2e | lib max %.2e mean %.2e   (relative to max|out|)' % (
        cin, positive, (d.abs().max() / ref.abs().max()).item(), (d.mean() / ref.abs().max()).item(),
        ((lib.double() - ref).abs().max() / ref.abs().max()).item(), ((lib.double() - ref).mean() / ref.abs().max()).item()))
