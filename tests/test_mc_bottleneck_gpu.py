"""Matrix-core bottleneck unit (spml_amd/mc_bottleneck.py) against the same unit on framework ops
(spml/models/backbones/resnet.py:11-63): outputs, input gradient, every parameter gradient and the
running statistics."""
import copy

import os

import pytest
import torch

from spml_amd import mc_bottleneck
from spml_amd.models.backbones.resnet import Bottleneck, _bn

DEV = 'cuda:0'
pytestmark = pytest.mark.gpu


def _make(inplanes, planes, dilation, downsample, seed):
  torch.manual_seed(seed)
  ds = None
  if downsample:
    ds = torch.nn.Sequential(torch.nn.Conv2d(inplanes, planes * 4, 1, bias=False), _bn(planes * 4))
  blk = Bottleneck(inplanes, planes, 1, dilation=dilation, downsample=ds)
  for m in blk.modules():
    if isinstance(m, torch.nn.Conv2d):
      fan = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
      m.weight.data.normal_(0, (2.0 / fan) ** 0.5)
    elif isinstance(m, torch.nn.BatchNorm2d):
      m.weight.data.uniform_(0.5, 1.5)
      m.bias.data.uniform_(-0.2, 0.2)
  return blk.to(DEV).to(memory_format=torch.channels_last).train()


def _run(blk, x, up, fused, monkeypatch):
  monkeypatch.setenv('SPML_NO_MC_CONV', '0' if fused else '1')
  monkeypatch.setenv('SPML_NO_FUSED_BN', '1')              # reference side: plain framework ops
  xi = x.clone().requires_grad_(True)
  assert mc_bottleneck.available(blk, xi) == fused
  y = blk(xi)
  (y * up).sum().backward()
  grads = {n: p.grad.clone() for n, p in blk.named_parameters()}
  stats = {n: b.clone() for n, b in blk.named_buffers()}
  return y.detach(), xi.grad, grads, stats


@pytest.mark.parametrize('inplanes,planes,dil,ds,n,h,w', [(1024, 256, 2, False, 2, 17, 19), (512, 256, 1, True, 2, 12, 9),
                                                          (2048, 512, 4, False, 1, 11, 13),
                                                          (512, 128, 1, False, 2, 15, 14),      # res3: 128-wide tiles
                                                          (256, 128, 1, True, 1, 9, 13)])
def test_unit_matches_framework_ops(inplanes, planes, dil, ds, n, h, w, monkeypatch):
  blk = _make(inplanes, planes, dil, ds, seed=inplanes + dil)
  ref = copy.deepcopy(blk)
  g = torch.Generator().manual_seed(7)
  x = torch.randn(n, inplanes, h, w, generator=g).clamp_min(0).to(DEV).contiguous(memory_format=torch.channels_last)
  up = (torch.randn(n, planes * 4, h, w, generator=g) * 1e-3).to(DEV).contiguous(memory_format=torch.channels_last)
  y1, dx1, g1, s1 = _run(blk, x, up, True, monkeypatch)
  y0, dx0, g0, s0 = _run(ref, x, up, False, monkeypatch)

  def close(a, b, tol, what):
    scale = b.abs().max().item()
    err = (a - b).abs().max().item()
    assert err <= tol * max(scale, 1e-30), (what, err, scale)

  close(y1, y0, 2e-5, 'output')
  close(dx1, dx0, 2e-4, 'input gradient')
  for k in g0:
    close(g1[k], g0[k], 5e-4, k)
  for k in s0:
    if k.endswith('num_batches_tracked'):
      assert int(s1[k]) == int(s0[k])
    else:
      close(s1[k], s0[k], 1e-5, k)


@pytest.mark.parametrize('inplanes,planes,ds,n,h,w', [(64, 64, True, 2, 33, 29), (256, 64, False, 3, 17, 20)])
def test_frozen_narrow_unit_runs_forward_only_on_the_matrix_cores(inplanes, planes, ds, n, h, w, monkeypatch):
  """res2 of the training recipes (`spml/models/backbones/resnet.py:66-178`; in no optimizer group, batch norms in
  training mode, input without gradient): 64-channel units have no gradient tiles and need none -- forward on the
  matrix-core path against the framework ops and fp64: output and running statistics.  With a trainable parameter
  the unit is refused (the framework path computes its gradients)."""
  blk = _make(inplanes, planes, 1, ds, seed=inplanes + planes)
  for p in blk.parameters():
    p.requires_grad_(False)
  ref, ref64 = copy.deepcopy(blk), copy.deepcopy(blk).double()
  g = torch.Generator().manual_seed(11)
  x = torch.randn(n, inplanes, h, w, generator=g).clamp_min(0).to(DEV).contiguous(memory_format=torch.channels_last)
  monkeypatch.setenv('SPML_NO_MC_CONV', '0')
  monkeypatch.setenv('SPML_MC_FROZEN_UNITS', '1')          # (opt-in: no faster than the library at 64 channels)
  assert mc_bottleneck.available(blk, x)
  y1 = blk(x)
  assert hasattr(y1, '_spml_hl8') and not y1.requires_grad
  monkeypatch.delenv('SPML_MC_FROZEN_UNITS')
  assert not mc_bottleneck.available(ref, x)
  y0 = ref(x)
  y64 = ref64(x.double())
  scale = y64.abs().max().item()
  e1, e0 = (y1.double() - y64).abs().max().item() / scale, (y0.double() - y64).abs().max().item() / scale
  assert e1 <= max((0.0 if os.environ.get('SPML_TEST_STRICT_FLOOR') == '1' else 2.0) * e0, 2e-6), (e1, e0)
  s1, s0 = dict(blk.named_buffers()), dict(ref.named_buffers())
  for k in s0:
    if k.endswith('num_batches_tracked'):
      assert int(s1[k]) == int(s0[k]) == 1
    else:
      torch.testing.assert_close(s1[k], s0[k], rtol=1e-5, atol=1e-5 * s0[k].abs().max().item())
  monkeypatch.setenv('SPML_MC_FROZEN_UNITS', '1')
  blk.conv1.weight.requires_grad_(True)
  assert not mc_bottleneck.available(blk, x)
  blk.conv1.weight.requires_grad_(False)
  assert not mc_bottleneck.available(blk, x.clone().requires_grad_(True))


def test_units_chain_through_the_hl8_side_channel(monkeypatch):
  """Two units in a row: the second takes the first one's split copy instead of converting again."""
  monkeypatch.setenv('SPML_NO_MC_CONV', '0')
  a, b = _make(1024, 256, 2, False, 1), _make(1024, 256, 2, False, 2)
  x = torch.randn(1, 1024, 9, 9).clamp_min(0).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
  mid = a(x)
  assert hasattr(mid, '_spml_hl8')
  out = b(mid)
  out.square().mean().backward()
  assert torch.isfinite(x.grad).all() and x.grad.abs().max() > 0


def _two_rank_worker(rank, port, state, x_all, up_all, out_q, cut=None, arch=(1024, 256, 2, False)):
  """One of two ranks (both on cuda:0, gloo): SyncBatchNorm statistics across the ranks."""
  import os
  import torch.distributed as dist
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), SPML_NO_MC_CONV='0')
  dist.init_process_group('gloo', rank=rank, world_size=2)
  real_all_gather = dist.all_gather

  def all_gather_via_host(outs, t, group=None):        # gloo has no CUDA all_gather: test-only staging
    host = [o.cpu() for o in outs]
    real_all_gather(host, t.cpu(), group=group)
    for o, h in zip(outs, host):
      o.copy_(h)
  dist.all_gather = all_gather_via_host
  blk = _make(*arch, seed=11)
  blk.load_state_dict(state)
  blk = torch.nn.SyncBatchNorm.convert_sync_batchnorm(blk).to(DEV).to(memory_format=torch.channels_last).train()
  cut = x_all.shape[0] // 2 if cut is None else cut
  lo, hi = (0, cut) if rank == 0 else (cut, x_all.shape[0])
  x = x_all[lo:hi].to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
  up = up_all[lo:hi].to(DEV).contiguous(memory_format=torch.channels_last)
  assert mc_bottleneck.available(blk, x)
  from spml_amd import parallel
  with parallel.count_collectives() as cc:
    y = blk(x)
    (y * up).sum().backward()
  # one all_gather / all_reduce per batch norm; the third and the downsample batch norm of a unit share theirs
  assert cc.total == 6, (cc.total, dict(cc.calls))
  grads = {n: p.grad.cpu().numpy() for n, p in blk.named_parameters()}      # numpy: pickled by value
  out_q.put((rank, y.detach().cpu().numpy(), x.grad.cpu().numpy(), grads,
             {n: b.cpu().numpy() for n, b in blk.named_buffers()}))
  dist.barrier()
  dist.destroy_process_group()


@pytest.mark.parametrize('cut,arch', [(2, (1024, 256, 2, False)), (1, (1024, 256, 2, False)), (2, (512, 256, 2, True)),
                                      (1, (512, 256, 2, True))])
def test_two_ranks_with_sync_batchnorm_match_one_rank_on_the_joint_batch(cut, arch, monkeypatch):
  """SyncBatchNorm path of the unit (statistics all-gathered / all-reduced between the kernel
  halves): two ranks with half the batch each == one rank with the whole batch; parameter
  gradients of the ranks sum to the single-rank ones.  cut = 1: the ranks hold 1 and 3 images --
  the backward divides by the SUM of the gathered row counts (lib/nn/sync_batchnorm/
  batchnorm.py:124-145), not by rows x world.  A unit with a downsample branch exchanges the statistics of its third and
  of its downsample batch norm together (forward: one all_gather of [3, 2 C]; backward: one all_reduce of [2, 2 C]): six
  collectives for four batch norms, counted."""
  import torch.multiprocessing as mp
  monkeypatch.setenv('SPML_NO_MC_CONV', '0')
  # the ranks pool statistics taken by the batch-norm pass; the single rank does the same here (the
  # convolution-epilogue statistics differ from them in the last bits, which may flip a ReLU mask bit
  # of a pre-activation next to zero: a large local change of the input gradient at this tiny size)
  monkeypatch.setenv('SPML_CONV_BN_STATS', '0')
  blk = _make(*arch, seed=11)
  state = {k: v.cpu() for k, v in blk.state_dict().items()}
  g = torch.Generator().manual_seed(3)
  x_all = torch.randn(4, arch[0], 9, 11, generator=g).clamp_min(0)
  up_all = torch.randn(4, 4 * arch[1], 9, 11, generator=g) * 1e-3
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  import socket
  sk = socket.socket()
  sk.bind(('127.0.0.1', 0))
  port = sk.getsockname()[1]
  sk.close()
  procs = [ctx.Process(target=_two_rank_worker, args=(r, port, state, x_all, up_all, q, cut, arch)) for r in range(2)]
  for p in procs:
    p.start()
  T = torch.from_numpy
  got = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
  got = [(r, T(y), T(dx), {k: T(v) for k, v in gr.items()}, {k: T(v) for k, v in bf.items()})
         for r, y, dx, gr, bf in got]
  for p in procs:
    p.join(60)
    assert p.exitcode == 0
  x = x_all.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
  y = blk(x)
  (y * up_all.to(DEV).contiguous(memory_format=torch.channels_last)).sum().backward()

  def close(a, b, tol, what):
    scale = b.abs().max().item()
    assert (a - b).abs().max().item() <= tol * max(scale, 1e-30), what

  close(torch.cat([got[0][1], got[1][1]]), y.detach().cpu(), 1e-5, 'output')
  close(torch.cat([got[0][2], got[1][2]]), x.grad.cpu(), 1e-4, 'input gradient')
  for n, p in blk.named_parameters():
    close(got[0][3][n] + got[1][3][n], p.grad.cpu(), 2e-4, n)
  for n, b in blk.named_buffers():
    if not n.endswith('num_batches_tracked'):
      close(got[0][4][n], b.cpu(), 1e-5, n)
      close(got[1][4][n], b.cpu(), 1e-5, n)


@pytest.mark.parametrize('inplanes,planes,dil,ds', [(1024, 256, 2, False), (512, 256, 1, True), (2048, 512, 4, False),
                                                    (512, 128, 1, False)])
def test_inference_unit_matches_framework_eval(inplanes, planes, dil, ds, monkeypatch):
  """Eval mode / no_grad: batch norm folded into the matrix-core convolutions vs the framework's eval
  forward (running statistics), chained over two calls (the second takes the first one's hl8 output)."""
  blk = _make(inplanes, planes, dil, ds, seed=5)
  with torch.no_grad():
    for m in blk.modules():
      if isinstance(m, torch.nn.BatchNorm2d):
        m.running_mean.uniform_(-0.3, 0.3)
        m.running_var.uniform_(0.5, 2.0)
  blk.eval()
  g = torch.Generator().manual_seed(2)
  x = torch.randn(2, inplanes, 13, 11, generator=g).clamp_min(0).to(DEV).contiguous(memory_format=torch.channels_last)
  with torch.no_grad():
    monkeypatch.setenv('SPML_NO_MC_CONV', '0')
    assert mc_bottleneck.eval_available(blk, x)
    got = blk(x)
    monkeypatch.setenv('SPML_NO_MC_CONV', '1')
    assert not mc_bottleneck.eval_available(blk, x)
    want = blk(x)
  assert hasattr(got, '_spml_hl8')
  err = (got - want).abs().max().item() / want.abs().max().item()
  assert err < 1e-5, err
  monkeypatch.setenv('SPML_NO_MC_CONV', '0')
  assert not mc_bottleneck.eval_available(blk, x)          # autograd on: framework path


def _close_up_to_relu_flips(a, b, tol, what, frac=2e-3):
  """Gradients of two fp32 evaluations of a network with ReLUs: a pre-activation within round-off of zero
  may get a different mask bit on the two sides, which changes the gradient inside that element's receptive
  field by O(1) of its size.  So: all but a small fraction of the elements within `tol` of the largest one,
  and the whole tensor close in the L2 sense."""
  scale = max(b.abs().max().item(), 1e-30)
  err = (a - b).abs()
  off = (err > tol * scale).float().mean().item()
  l2 = (err.double().norm() / b.double().norm().clamp_min(1e-30)).item()
  assert off <= frac and l2 <= 50 * tol, (what, off, l2, err.max().item() / scale)


@pytest.mark.parametrize('cin,cout,k,dil,n,h,w', [(384, 256, 3, 1, 2, 15, 13), (512, 512, 3, 2, 1, 12, 17),
                                                  (1024, 128, 3, 1, 2, 9, 11), (256, 256, 1, 1, 2, 14, 14)])
def test_conv_bn_act_matches_framework_ops(cin, cout, k, dil, n, h, w, monkeypatch):
  """One convolution + training batch norm + ReLU on the matrix-core path (the closing convolution of the
  pyramid-pooling head) against the same three framework ops."""
  torch.manual_seed(cin + cout)
  conv = torch.nn.Conv2d(cin, cout, k, 1, dil * (k // 2), dil, bias=False).to(DEV).to(memory_format=torch.channels_last)
  conv.weight.data.normal_(0, (2.0 / (k * k * cout)) ** 0.5)
  bn = _bn(cout).to(DEV).train()
  bn.weight.data.uniform_(0.5, 1.5)
  bn.bias.data.uniform_(-0.2, 0.2)
  bn_ref = copy.deepcopy(bn)
  g = torch.Generator().manual_seed(5)
  x = torch.randn(n, cin, h, w, generator=g).clamp_min(0).to(DEV).contiguous(memory_format=torch.channels_last)
  up = (torch.randn(n, cout, h, w, generator=g) * 1e-3).to(DEV).contiguous(memory_format=torch.channels_last)
  monkeypatch.setenv('SPML_NO_MC_CONV', '0')
  x1 = x.clone().requires_grad_(True)
  assert mc_bottleneck.conv_bn_act_available(conv, bn, x1)
  y1 = mc_bottleneck.conv_bn_act(conv, bn, x1)
  (y1 * up).sum().backward()
  dw1, conv.weight.grad = conv.weight.grad.clone(), None
  # reference: the same three ops in fp64 on the CPU
  conv64, bn64 = copy.deepcopy(conv).cpu().double(), bn_ref.cpu().double()
  x0 = x.cpu().double().contiguous().requires_grad_(True)
  y0 = torch.relu(bn64(conv64(x0)))
  (y0 * up.cpu().double()).sum().backward()

  def close(a, b, tol, what):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    scale = b.abs().max().item()
    err = (a - b).abs().max().item()
    assert err <= tol * max(scale, 1e-30), (what, err, scale)

  close(y1, y0, 2e-5, 'output')
  _close_up_to_relu_flips(x1.grad.cpu().double(), x0.grad, 2e-4, 'input gradient')
  close(dw1, conv64.weight.grad, 5e-4, 'weight gradient')
  close(bn.weight.grad, bn64.weight.grad, 5e-4, 'gamma')
  close(bn.bias.grad, bn64.bias.grad, 5e-4, 'beta')
  close(bn.running_mean, bn64.running_mean, 1e-5, 'running mean')
  close(bn.running_var, bn64.running_var, 1e-5, 'running var')


def test_pyramid_pooling_head_fast_paths(monkeypatch):
  """PSPP on a channels-last map: the four adaptive pools as one GEMM and the closing convolution on the
  matrix cores, against the plain module (framework adaptive pools and convolution)."""
  from spml_amd.models.heads.spp import PSPP, _pyramid_pool
  torch.manual_seed(3)
  head = PSPP(512, 128).to(DEV).to(memory_format=torch.channels_last).train()
  ref = copy.deepcopy(head)
  g = torch.Generator().manual_seed(9)
  x = torch.randn(2, 512, 13, 17, generator=g).clamp_min(0).to(DEV).contiguous(memory_format=torch.channels_last)
  for b, p in zip((1, 2, 3, 6), _pyramid_pool(x, [1, 2, 3, 6])):
    torch.testing.assert_close(p, torch.nn.functional.adaptive_avg_pool2d(x, b), rtol=1e-5, atol=1e-6)
  up = (torch.randn(2, 128, 13, 17, generator=g) * 1e-3).to(DEV)
  monkeypatch.setenv('SPML_NO_MC_CONV', '0')
  x1 = x.clone().requires_grad_(True)
  y1 = head(x1)
  (y1 * up).sum().backward()
  # reference: the plain module in fp64 on the CPU.  (At batch 2 the batch norm of the 1x1-pooled branch sees
  # two samples per channel -- ill-conditioned: the GPU framework path's run-to-run noise from the atomics of
  # the bilinear backward is amplified to 1-5 % of the input gradient in 5 of 12 runs, while every op on its own
  # is reproducible; the fast path above is stable at 3e-6 of the fp64 result.)
  ref64 = ref.cpu().double()
  x0 = x.cpu().double().contiguous().requires_grad_(True)
  y0 = ref64(x0)
  (y0 * up.cpu().double()).sum().backward()
  sc = y0.abs().max().item()
  assert (y1.detach().cpu().double() - y0.detach()).abs().max().item() <= 5e-5 * sc
  _close_up_to_relu_flips(x1.grad.cpu().double(), x0.grad, 5e-4, 'input gradient')
  for (k, p1), (_, p0) in zip(head.named_parameters(), ref64.named_parameters()):
    _close_up_to_relu_flips(p1.grad.cpu().double(), p0.grad, 1e-3, k, frac=2e-2)
