"""Fused training-mode batch norm + ReLU + residual (spml_amd/csrc/bn_act.hip) against the
framework ops it replaces in the bottleneck unit (spml/models/backbones/resnet.py:42-63)."""
import pytest
import torch

from spml_amd import ops

DEV = 'cuda:0'


def test_rank_statistics_merge_is_the_pooled_variance():
  """merge_bn_statistics (the SyncBatchNorm combination of per-rank statistics) on CPU."""
  g = torch.Generator().manual_seed(0)
  parts = [torch.randn(n, 7, generator=g) * (i + 1) + i for i, n in enumerate((50, 31, 64))]
  counts = torch.tensor([[float(p.shape[0])] * 7 for p in parts])
  means = torch.stack([p.mean(0) for p in parts])
  m2s = torch.stack([((p - p.mean(0)) ** 2).sum(0) for p in parts])
  total, mean, m2 = ops.merge_bn_statistics(counts, means, m2s)
  allx = torch.cat(parts)
  torch.testing.assert_close(mean, allx.mean(0), rtol=1e-5, atol=1e-6)
  torch.testing.assert_close(m2 / total, allx.var(0, unbiased=False), rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('n,c,h,w,relu,res', [(4, 64, 17, 19, True, False), (2, 256, 9, 33, True, True),
                                              (3, 2048, 5, 7, True, True), (2, 128, 12, 12, False, False),
                                              (16, 512, 33, 33, True, True), (1, 4, 3, 3, True, True)])
def test_fused_bn_act_matches_framework_ops(n, c, h, w, relu, res):
  gen = torch.Generator().manual_seed(n * 1000 + c)
  x = (torch.randn(n, c, h, w, generator=gen) * 2.0 + 0.7).to(DEV).contiguous(memory_format=torch.channels_last)
  r = torch.randn(n, c, h, w, generator=gen).to(DEV).contiguous(memory_format=torch.channels_last) if res else None
  up = torch.randn(n, c, h, w, generator=gen).to(DEV).contiguous(memory_format=torch.channels_last)

  def run(fused):
    bn = torch.nn.BatchNorm2d(c, momentum=3e-4).to(DEV)
    with torch.no_grad():
      bn.weight.copy_(torch.linspace(0.5, 1.5, c))
      bn.bias.copy_(torch.linspace(-0.3, 0.3, c))
    bn.train()
    xi = x.clone().requires_grad_(True)
    ri = r.clone().requires_grad_(True) if res else None
    if fused:
      assert ops.fused_bn_act_available(xi, bn)
      y = ops.batch_norm_act(xi, bn, relu=relu, residual=ri)
    else:
      y = bn(xi)
      if res:
        y = y + ri
      if relu:
        y = torch.relu(y)
    (y * up).sum().backward()
    return (y.detach(), xi.grad, ri.grad if res else None, bn.weight.grad, bn.bias.grad,
            bn.running_mean.clone(), bn.running_var.clone(), int(bn.num_batches_tracked))

  got, want = run(True), run(False)
  assert got[0].is_contiguous(memory_format=torch.channels_last)
  torch.testing.assert_close(got[0], want[0], rtol=1e-5, atol=1e-5)
  scale = want[1].abs().max().item()
  torch.testing.assert_close(got[1], want[1], rtol=1e-4, atol=1e-5 * max(1.0, scale))
  if res:
    torch.testing.assert_close(got[2], want[2], rtol=0, atol=0)
  for i in (3, 4):
    torch.testing.assert_close(got[i], want[i], rtol=1e-4, atol=1e-4 * max(1.0, want[i].abs().max().item()))
  torch.testing.assert_close(got[5], want[5], rtol=1e-5, atol=1e-7)
  torch.testing.assert_close(got[6], want[6], rtol=1e-5, atol=1e-7)
  assert got[7] == want[7] == 1


@pytest.mark.gpu
def test_fused_bn_falls_back_outside_its_domain():
  bn = torch.nn.BatchNorm2d(8).to(DEV)
  x = torch.randn(2, 8, 5, 5, device=DEV)                     # NCHW: framework ops
  assert not ops.fused_bn_act_available(x, bn)
  y = ops.batch_norm_act(x, bn)
  assert y.shape == x.shape and (y >= 0).all()
  bn.eval()                                                   # eval mode: running statistics
  xl = x.contiguous(memory_format=torch.channels_last)
  assert not ops.fused_bn_act_available(xl, bn)
  torch.testing.assert_close(ops.batch_norm_act(xl, bn, relu=False), bn(xl))


@pytest.mark.gpu
def test_large_mean_small_variance_is_stable():
  """Chunked statistics merged with Chan's formula: no E[x^2] - E[x]^2 cancellation."""
  gen = torch.Generator().manual_seed(1)
  x = (100.0 + 1e-2 * torch.randn(8, 16, 40, 40, generator=gen)).to(DEV).contiguous(memory_format=torch.channels_last)
  bn = torch.nn.BatchNorm2d(16).to(DEV).train()
  y = ops.batch_norm_act(x, bn, relu=False)
  ref = torch.nn.functional.batch_norm(x.double(), None, None, bn.weight.double(), bn.bias.double(), True, 0.1, bn.eps)
  assert (y.double() - ref).abs().max().item() < 5e-2       # fp32 input resolution at |x| ~ 100 is ~1e-5 / 1e-2
  var = x.double().var(dim=(0, 2, 3), unbiased=True)
  torch.testing.assert_close(bn.running_var.double(), 0.9 + 0.1 * var, rtol=5e-3, atol=0)


@pytest.mark.gpu
def test_two_call_and_single_call_forms_agree():
  """The split form (statistics -> finalisation -> apply, what SyncBatchNorm interleaves with its
  collectives) and the single-call form of the hl8-producing batch norm give the same tensors, bounds,
  masks and running statistics; the gathered-ranks finalisation with one rank equals the plain one."""
  from spml_amd import _ffi
  g = torch.Generator().manual_seed(9)
  n, c, h, w = 3, 256, 9, 11
  rows = n * h * w
  x = (torch.randn(n, c, h, w, generator=g) * 1.7 + 0.3).to(DEV).contiguous(memory_format=torch.channels_last)
  res = torch.randn(n, c, h, w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
  gamma = torch.linspace(0.5, 1.5, c, device=DEV)
  beta = torch.linspace(-0.2, 0.2, c, device=DEV)
  resb = res.abs().max().reshape(1).clone()
  rm1, rv1 = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
  y1, yh1, b1, m1, saved1 = _ffi.bn_fwd_hl8(x, rows, c, res, resb, gamma, beta, rm1, rv1, 0.1, 1e-5, True, True, True,
                                            True)
  st = _ffi.bn_stats_ext(x, rows, c)
  rm2, rv2 = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
  invstd = _ffi.bn_finalize(st[1], st[2], rows, 1e-5, 0.1, rm2, rv2)
  y2, yh2, b2, m2 = _ffi.bn_act_apply_hl8(x, rows, c, res, resb, st[1], invstd, gamma, beta, st[3], st[4], True, True,
                                          True, want_mask=True)
  torch.testing.assert_close(y2, y1, rtol=1e-6, atol=1e-6)
  torch.testing.assert_close(b2, b1, rtol=1e-6, atol=0)
  assert torch.equal(m2, m1)
  assert (yh2.data != yh1.data).float().mean().item() < 1e-3           # last-bit differences at most
  torch.testing.assert_close(rm2, rm1, rtol=1e-6, atol=1e-7)
  torch.testing.assert_close(rv2, rv1, rtol=1e-6, atol=1e-7)
  rm3, rv3 = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
  mean3, invstd3, total3 = _ffi.bn_finalize_ranks(st[:3].unsqueeze(0).contiguous(), 1e-5, 0.1, rm3, rv3)
  torch.testing.assert_close(mean3, st[1], rtol=1e-6, atol=1e-7)
  torch.testing.assert_close(invstd3, invstd, rtol=1e-6, atol=1e-7)
  torch.testing.assert_close(rv3, rv2, rtol=1e-6, atol=1e-7)
  # against the framework
  bn = torch.nn.BatchNorm2d(c, momentum=0.1).to(DEV).train()
  with torch.no_grad():
    bn.weight.copy_(gamma)
    bn.bias.copy_(beta)
  want = torch.relu(bn(x) + res)
  torch.testing.assert_close(y1, want, rtol=1e-5, atol=1e-5)
  assert float(b1) >= float(want.abs().max()) * 0.9999
  assert float(total3) == float(rows)


def _unequal_ranks_worker(rank, port, x_all, r_all, up_all, out_q):
  import os
  import torch.distributed as dist
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  dist.init_process_group('gloo', rank=rank, world_size=2)
  try:
    lo, hi = (0, 1) if rank == 0 else (1, x_all.shape[0])
    c = x_all.shape[1]
    bn = torch.nn.SyncBatchNorm(c, momentum=0.1).to(DEV).train()
    with torch.no_grad():
      bn.weight.copy_(torch.linspace(0.5, 1.5, c))
      bn.bias.copy_(torch.linspace(-0.3, 0.3, c))
    cl = dict(memory_format=torch.channels_last)
    x = x_all[lo:hi].to(DEV).contiguous(**cl).requires_grad_(True)
    r = r_all[lo:hi].to(DEV).contiguous(**cl).requires_grad_(True)
    up = up_all[lo:hi].to(DEV).contiguous(**cl)
    assert ops.fused_bn_act_available(x, bn)
    y = ops.batch_norm_act(x, bn, relu=True, residual=r)
    (y * up).sum().backward()
    out_q.put((rank, 'ok', y.detach().cpu().numpy(), x.grad.cpu().numpy(), r.grad.cpu().numpy(),
               bn.weight.grad.cpu().numpy(), bn.bias.grad.cpu().numpy(), bn.running_var.cpu().numpy()))
  except Exception as e:       # noqa: BLE001 -- reported to the parent
    import traceback
    out_q.put((rank, traceback.format_exc() + repr(e)))
  dist.barrier()
  dist.destroy_process_group()


@pytest.mark.gpu
def test_sync_batchnorm_with_unequal_rank_batches():
  """SyncBatchNorm through the fused kernels with 1 image on rank 0 and 3 on rank 1 (sharing this GPU
  over gloo) == one rank on the joint batch: the backward uses the SUM of the gathered row counts
  (lib/nn/sync_batchnorm/batchnorm.py:124-145), read on the device."""
  import socket
  import torch.multiprocessing as mp
  g = torch.Generator().manual_seed(12)
  n, c, h, w = 4, 64, 9, 7
  x_all = torch.randn(n, c, h, w, generator=g) * 1.5 + 0.4
  r_all = torch.randn(n, c, h, w, generator=g)
  up_all = torch.randn(n, c, h, w, generator=g)
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  sk = socket.socket()
  sk.bind(('127.0.0.1', 0))
  port = sk.getsockname()[1]
  sk.close()
  procs = [ctx.Process(target=_unequal_ranks_worker, args=(k, port, x_all, r_all, up_all, q)) for k in range(2)]
  for p in procs:
    p.start()
  got = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
  for p in procs:
    p.join(60)
  for t in got:
    assert t[1] == 'ok', t[1]
  T = torch.from_numpy
  bn = torch.nn.BatchNorm2d(c, momentum=0.1).to(DEV).train()
  with torch.no_grad():
    bn.weight.copy_(torch.linspace(0.5, 1.5, c))
    bn.bias.copy_(torch.linspace(-0.3, 0.3, c))
  x = x_all.to(DEV).requires_grad_(True)
  r = r_all.to(DEV).requires_grad_(True)
  y = torch.relu(bn(x) + r)
  (y * up_all.to(DEV)).sum().backward()
  torch.testing.assert_close(torch.cat([T(got[0][2]), T(got[1][2])]), y.detach().cpu(), rtol=1e-5, atol=1e-5)
  torch.testing.assert_close(torch.cat([T(got[0][3]), T(got[1][3])]), x.grad.cpu(), rtol=1e-4, atol=1e-5)
  torch.testing.assert_close(torch.cat([T(got[0][4]), T(got[1][4])]), r.grad.cpu(), rtol=1e-5, atol=1e-6)
  torch.testing.assert_close(T(got[0][5]) + T(got[1][5]), bn.weight.grad.cpu(), rtol=1e-4, atol=1e-4)
  torch.testing.assert_close(T(got[0][6]) + T(got[1][6]), bn.bias.grad.cpu(), rtol=1e-4, atol=1e-4)
  torch.testing.assert_close(T(got[0][7]), bn.running_var.cpu(), rtol=1e-5, atol=1e-6)
