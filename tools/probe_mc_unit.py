"""Error of one bottleneck unit (forward, input gradient, weight gradients) against an fp64 run of
the same unit: matrix-core path vs the framework's fp32 path."""
import copy, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import torch
from test_mc_bottleneck_gpu import _make

for inplanes, planes, dil, ds, n, h, w in [(1024, 256, 2, False, 4, 33, 33), (2048, 512, 4, False, 2, 33, 33), (512, 256, 1, True, 4, 33, 33)]:
  blk = _make(inplanes, planes, dil, ds, seed=3)
  g = torch.Generator().manual_seed(1)
  x = torch.randn(n, inplanes, h, w, generator=g).clamp_min(0).cuda().contiguous(memory_format=torch.channels_last)
  up = (torch.randn(n, planes * 4, h, w, generator=g) * 1e-3).cuda().contiguous(memory_format=torch.channels_last)

  def run(b, xx, uu, mc):
    os.environ['SPML_NO_MC_CONV'] = '0' if mc else '1'
    os.environ['SPML_NO_FUSED_BN'] = '1'
    xi = xx.clone().requires_grad_(True)
    y = b(xi)
    (y * uu).sum().backward()
    return [y.detach(), xi.grad] + [p.grad for _, p in sorted(b.named_parameters())]

  ref = run(copy.deepcopy(blk).double(), x.double(), up.double(), False)
  names = ['out', 'dx'] + [k for k, _ in sorted(blk.named_parameters())]
  e_mc = run(copy.deepcopy(blk), x, up, True)
  e_fw = run(copy.deepcopy(blk), x, up, False)
  print('unit %d/%d dil %d ds %d' % (inplanes, planes, dil, ds))
  for nm, r, a, b in zip(names, ref, e_mc, e_fw):
    s = r.abs().max().item()
    print('  %-28s matrix-core %.2e   framework fp32 %.2e   (max abs err / max |ref|)' % (
        nm, (a.double() - r).abs().max().item() / s, (b.double() - r).abs().max().item() / s))
