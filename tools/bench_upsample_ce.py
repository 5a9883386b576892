"""Time the cross-entropy of the up-sampled softmax-head logits: fused kernels against the framework ops
(batch 16, 21 classes, 130x130 -> 513x513, forward + backward)."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from spml_amd import ops

dev = 'cuda:0'
torch.manual_seed(0)
logits = (torch.randn(16, 21, 130, 130, device=dev) * 3).contiguous(memory_format=torch.channels_last)
labels = torch.randint(0, 21, (16, 513, 513), device=dev)
labels[torch.rand(16, 513, 513, device=dev) < 0.1] = 255


def run(fused):
  x = logits.clone().requires_grad_(True)
  if fused:
    loss = ops.upsample_cross_entropy(x, labels, 255)
  else:
    loss = F.cross_entropy(F.interpolate(x, size=(513, 513), mode='bilinear'), labels, ignore_index=255)
  loss.backward()
  return loss


for fused in (True, False):
  for _ in range(3):
    run(fused)
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(10):
    l = run(fused)
  b.record()
  torch.cuda.synchronize()
  print('fused' if fused else 'framework', '%.1f us per forward + backward' % (a.elapsed_time(b) * 100), 'loss', float(l))

from spml_amd import _ffi
nhwc = logits.permute(0, 2, 3, 1).contiguous()
res, lse = _ffi.upsample_ce_fwd(nhwc, labels, 255)
scale = (1.0 / res[1]).reshape(1).contiguous()
for name, fn in (('fwd kernels', lambda: _ffi.upsample_ce_fwd(nhwc, labels, 255)),
                 ('bwd kernel', lambda: _ffi.upsample_ce_bwd(nhwc, labels, lse, 255, scale))):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(10):
    fn()
  b.record()
  torch.cuda.synchronize()
  print(name, '%.1f us' % (a.elapsed_time(b) * 100))
