"""Edge shapes through the C-ABI against the oracle: single rows / single clusters, sizes
below one tile, widths at the kernels' limits, empty inputs, many tiny images."""
import pytest
import torch

from oracle import spml_oracle as O
from spml_amd import _ffi

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def unit(gen, n, d):
  return torch.nn.functional.normalize(torch.randn(n, d, generator=gen), dim=1)


@pytest.mark.parametrize('lens,d,k', [([5], 66, 4), ([1, 1, 1], 34, 1), ([40, 0, 0, 7], 258, 36),
                                      ([33] * 40, 66, 9), ([64], 32, 64), ([31], 130, 2)])
def test_kmeans_tiny_and_degenerate_batches(lens, d, k):
  gen = torch.Generator().manual_seed(sum(lens) + d)
  xs = [unit(gen, n, d) for n in lens]
  inits = [torch.randint(0, k, (n,), generator=gen) for n in lens]
  off = torch.zeros(len(lens) + 1, dtype=torch.long)
  off[1:] = torch.cumsum(torch.tensor(lens), 0)
  x, init = torch.cat(xs).to(DEV), torch.cat(inits).to(DEV)
  for iters in (0, 1, 3):
    lab = _ffi.kmeans_run(x, off.to(DEV), max(max(lens), 1), k, init, iters)
    o = 0
    for xi, ii, n in zip(xs, inits, lens):
      if n:
        trace = []
        want = O.kmeans_with_initial_labels(xi, ii, k, iters, trace=trace) if iters else ii
        got = lab[o:o + n].cpu()
        if iters and k > 1:
          safe = torch.stack([t['margin'] for t in trace]).min(0).values > 1e-4
          assert torch.equal(got[safe], want[safe])
        else:
          assert torch.equal(got, want)
      o += n


@pytest.mark.parametrize('p,m,d', [(1, 1, 8), (33, 31, 2), (100, 1000, 272), (7, 2, 64), (65, 33, 17),
                                   (40, 70, 273), (90, 300, 528)])
def test_nll_small_and_limit_shapes(p, m, d):
  gen = torch.Generator().manual_seed(p * 7 + m)
  protos = unit(gen, m, d)
  own = torch.randint(0, m, (p,), generator=gen)
  emb = torch.nn.functional.normalize(protos[own] + 0.5 * torch.randn(p, d, generator=gen), dim=1)
  pr_lab = torch.randint(0, 3, (m,), generator=gen)
  px_lab = pr_lab[own]
  e = emb.clone().requires_grad_(True)
  pr = protos.clone().requires_grad_(True)
  want = O.segsort_nll(e, px_lab, own, pr, pr_lab, 10.0)
  g = torch.rand(p, generator=gen) + 0.1
  (want.view(-1) * g).sum().backward()
  nll, stats = _ffi.segsort_nll_fwd(emb.to(DEV), own.to(DEV), px_lab.to(DEV), protos.to(DEV),
                                    pr_lab.to(DEV), 10.0, 0)
  # (single-prototype rows hit the reference's own fallback: num = own similarity)
  torch.testing.assert_close(nll.cpu(), want.detach().view(-1), rtol=2e-5, atol=2e-5)
  d_emb, d_pr = _ffi.segsort_nll_bwd(emb.to(DEV), own.to(DEV), px_lab.to(DEV), protos.to(DEV),
                                     pr_lab.to(DEV), 10.0, 0, stats, g.to(DEV))
  torch.testing.assert_close(d_emb.cpu(), e.grad, rtol=1e-3, atol=2e-5 * max(1.0, e.grad.abs().max().item()))
  torch.testing.assert_close(d_pr.cpu(), pr.grad, rtol=1e-3, atol=2e-5 * max(1.0, pr.grad.abs().max().item()))


def test_nll_width_limit_and_empty_input():
  with pytest.raises(_ffi.SpmlHipError):
    _ffi.segsort_nll_fwd(torch.zeros(4, 529, device=DEV), torch.zeros(4, dtype=torch.long, device=DEV),
                         torch.zeros(4, dtype=torch.long, device=DEV), torch.zeros(2, 529, device=DEV),
                         torch.zeros(2, dtype=torch.long, device=DEV), 10.0, 0)
  from spml_amd import ops
  out = ops.segsort_nll(torch.zeros(0, 16, device=DEV), torch.zeros(0, dtype=torch.long, device=DEV),
                        torch.zeros(0, dtype=torch.long, device=DEV), torch.ones(3, 16, device=DEV),
                        torch.zeros(3, dtype=torch.long, device=DEV), 10.0)
  assert out.shape == (0,)


@pytest.mark.parametrize('q,m,d,k', [(1, 1, 4, 1), (5, 3, 64, 3), (70, 40, 528, 32), (33, 1000, 66, 20)])
def test_topk_small_and_limit_shapes(q, m, d, k):
  gen = torch.Generator().manual_seed(q + m + d)
  qs, pr = unit(gen, q, d), unit(gen, m, d)
  idx, val = _ffi.topk_affinity(qs.to(DEV), pr.to(DEV), k)
  sim = qs @ pr.t()
  want_v, want_i = torch.sort(sim, dim=1, descending=True, stable=True)
  torch.testing.assert_close(val.cpu(), want_v[:, :k], rtol=0, atol=3e-6)
  # indices agree wherever neighbouring values are not a near tie
  gaps = (want_v[:, :k] - want_v[:, 1:k + 1]).abs() if m > k else torch.ones(q, k)
  prev = torch.cat([torch.ones(q, 1), gaps[:, :-1]], 1)
  safe = (gaps > 1e-5) & (prev > 1e-5) if m > k else torch.ones(q, k, dtype=torch.bool)
  assert torch.equal(idx.cpu()[safe], want_i[:, :k][safe])
  with pytest.raises(_ffi.SpmlHipError):
    _ffi.topk_affinity(qs.to(DEV), pr.to(DEV), 33)


@pytest.mark.parametrize('n,c,h,w', [(1, 1, 1, 1), (2, 3, 1, 5), (1, 700, 3, 3)])
def test_k1_degenerate_maps(n, c, h, w):
  gen = torch.Generator().manual_seed(c)
  emb = torch.randn(n, c, h, w, generator=gen)
  oe, ol = _ffi.normalize_concat_loc(emb.to(DEV))
  e = O.normalize_embedding(emb.permute(0, 2, 3, 1).contiguous())
  loc = (O.generate_location_features((h, w), 'float') - 0.5).view(1, h, w, 2).expand(n, h, w, 2)
  el = O.normalize_embedding(torch.cat([e, loc], -1))
  torch.testing.assert_close(oe.cpu(), e.reshape(-1, c), rtol=0, atol=1e-6)
  torch.testing.assert_close(ol.cpu(), el.reshape(-1, c + 2), rtol=0, atol=1e-6)


def test_segment_prototypes_gaps_and_single_segment():
  gen = torch.Generator().manual_seed(9)
  x = torch.randn(50, 66, generator=gen)
  ids = torch.tensor([0] * 20 + [3] * 25 + [7] * 5)          # segments 1, 2, 4, 5, 6 are empty
  protos, _ = _ffi.segment_sum_normalize(x.to(DEV), ids.to(DEV), 9)
  want = O.calculate_prototypes_from_labels(x, ids, 9)
  torch.testing.assert_close(protos.cpu(), want, rtol=0, atol=2e-6)
  assert (protos[[1, 2, 4, 5, 6, 8]] == 0).all()
  one, _ = _ffi.segment_sum_normalize(x.to(DEV), torch.zeros(50, dtype=torch.long, device=DEV), 1)
  torch.testing.assert_close(one.cpu(), O.calculate_prototypes_from_labels(x, torch.zeros(50, dtype=torch.long), 1),
                             rtol=0, atol=2e-6)


def test_kmeans_dispatch_fuzz():
  """Random shapes across every code path (tile kernels, the two many-cluster kernels, v2,
  generic): run == oracle away from near ties, stand-alone assign == run's last E-step."""
  import random
  rng = random.Random(1234)
  dims = [8, 16, 18, 32, 33, 34, 37, 40, 64, 66, 69, 96, 98, 128, 130, 136, 258, 264, 320, 400, 514]
  seen = set()
  for trial in range(36):
    d = rng.choice(dims)
    k = rng.choice([1, 2, 5, 16, 17, 36, 63, 64, 65, 100, 144, 200, 256, 257, 300])
    n_img = rng.choice([1, 1, 2, 3])
    lens = [rng.randint(1, 1500) for _ in range(n_img)]
    if k >= 128 and n_img == 1:
      lens = [rng.randint(1024, 2500)]
    iters = rng.choice([1, 2, 3])
    gen = torch.Generator().manual_seed(trial)
    cent = unit(gen, k, d)
    xs, inits = [], []
    for n in lens:
      own = torch.randint(0, k, (n,), generator=gen)
      xs.append(torch.nn.functional.normalize(cent[own] + 0.4 * torch.randn(n, d, generator=gen), dim=1))
      inits.append(torch.randint(0, k, (n,), generator=gen))
    off = torch.zeros(n_img + 1, dtype=torch.long)
    off[1:] = torch.cumsum(torch.tensor(lens), 0)
    x, init = torch.cat(xs).to(DEV), torch.cat(inits).to(DEV)
    lab, cen = _ffi.kmeans_run(x, off.to(DEV), max(lens), k, init, iters, want_centroids=True)
    seen.add(_ffi.kmeans_last_path())
    what = 'trial %d: d=%d k=%d lens=%s iters=%d path=%s' % (trial, d, k, lens, iters, _ffi.kmeans_last_path())
    o = 0
    for b, (xi, ii, n) in enumerate(zip(xs, inits, lens)):
      trace = []
      O.kmeans_with_initial_labels(xi, ii, k, iters, trace=trace)
      # prototypes of the last E-step: exact up to rounding only while no earlier near tie
      # flipped, so compare the decision against OUR prototypes (independent fp32 GEMM)
      pr = cen[b].cpu()
      sim = xi @ pr.t()
      got = lab[o:o + n].cpu()
      if k > 1:
        top2 = sim.topk(2, dim=1).values
        safe = (top2[:, 0] - top2[:, 1]) > 1e-4
      else:
        safe = torch.ones(n, dtype=torch.bool)
      assert torch.equal(got[safe], sim.argmax(1)[safe]), what
      if iters == 1:
        torch.testing.assert_close(pr, trace[0]['prototypes'], rtol=0, atol=3e-6, msg=what)
      o += n
    lab2 = _ffi.kmeans_assign(x, off.to(DEV), max(lens), cen)
    assert (lab2 != lab).float().mean().item() < 5e-3, what
  assert {'mfma_f16x2_v4p', 'mfma_f16x2_v3p', 'mfma_f16x2_v3', 'mfma_f16x2_v3k', 'mfma_f16x2_bigk'} <= seen, seen


def test_k1_refuses_a_wide_map_that_needs_a_gradient_in_the_forward():
  """spml_normalize_concat_*_bwd_f32 take at most 512 channels: the wrapper refuses such a map before the forward
  runs (not first in the backward); without a gradient the forward still takes it."""
  from spml_amd import ops, _ffi
  x = torch.randn(1, 520, 9, 9, device=DEV)
  out, _ = ops.normalize_concat_loc(x)
  assert out.shape == (81, 520)
  with pytest.raises(_ffi.SpmlHipError):
    ops.normalize_concat_loc(x.clone().requires_grad_(True))


@pytest.mark.parametrize('n,c,h,w', [(2, 128, 257, 257), (1, 4, 5, 7), (3, 64, 8, 8), (1, 12, 1, 1), (2, 8, 2, 3)])
def test_channels_last_max_pool_matches_the_framework(n, c, h, w):
  """`spml_maxpool3x3s2_nhwc_f32` (the frozen stem's nn.MaxPool2d(3, 2, 1), resnet.py:66-110): exact, odd / even /
  degenerate sizes, negative inputs (padding must not win), NaN propagation."""
  import torch.nn.functional as F
  from spml_amd import _ffi
  gen = torch.Generator().manual_seed(n + c + h + w)
  x = (torch.randn(n, c, h, w, generator=gen) - 2.0).to(DEV).contiguous(memory_format=torch.channels_last)
  if h > 4:
    x[0, 0, 2, 3] = float('nan')
  got = _ffi.maxpool3x3s2_nhwc(x)
  want = F.max_pool2d(x, 3, 2, 1)
  assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
  assert torch.equal(torch.nan_to_num(got, nan=12345.0), torch.nan_to_num(want, nan=12345.0))
