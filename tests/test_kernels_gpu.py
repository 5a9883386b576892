"""Parity of the HIP kernels (through the C-ABI) against the CPU oracle and the
golden vectors generated from the reference.  Needs a real MI355X."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import spml_oracle as O

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


def ffi():
  from spml_amd import _ffi
  return _ffi


def check_labels(got, want, margin, tol=1e-5, what=''):
  """Labels must be identical except where the oracle's own top-2 margin is
  below `tol` (a near tie that fp32 summation order may flip)."""
  got = got.cpu()
  bad = (got != want).nonzero().view(-1)
  if bad.numel() == 0:
    return 0
  worst = margin[bad].max().item()
  assert worst < tol, '%s: %d label mismatches, worst oracle margin %.3g' % (
      what, bad.numel(), worst)
  return bad.numel()


def seg_offsets(lengths):
  off = torch.zeros(len(lengths) + 1, dtype=torch.int64)
  off[1:] = torch.cumsum(torch.tensor(lengths), 0)
  return off.to(DEV)


def check_kmeans_stepwise(xs, inits, lens, k, iters, path, flags=0, first=1):
  """Pins every M- and E-step of a run without letting a flipped near tie cascade: for
  it = 1..iters the run with `it` iterations must return (a) prototypes equal (2e-6) to
  the oracle's M-step on OUR labels of iteration it-1 and (b) labels equal to the
  oracle's E-step against those prototypes except where the oracle's own top-2 margin is
  below 1e-5.  Returns the labels of the last iteration."""
  x = torch.cat(xs).to(DEV)
  init = torch.cat(inits).to(DEV)
  off = seg_offsets(lens)
  prev = init.cpu()
  if first > 1:
    prev = ffi().kmeans_run(x, off, max(lens), k, init, first - 1, flags=flags).cpu()
  for it in range(first, iters + 1):
    lab, cen = ffi().kmeans_run(x, off, max(lens), k, init, it, want_centroids=True, flags=flags)
    if path is not None:
      assert ffi().kmeans_last_path() == path, (it, ffi().kmeans_last_path())
    lab = lab.cpu()
    o = 0
    for b, (xi, n) in enumerate(zip(xs, lens)):
      if n:
        pr = O.calculate_prototypes_from_labels(xi, prev[o:o + n], k)
        torch.testing.assert_close(cen[b].cpu(), pr, rtol=0, atol=2e-6)
        sims = xi @ pr.t()
        t2 = sims.topk(2, dim=1).values
        check_labels(lab[o:o + n], sims.argmax(1), t2[:, 0] - t2[:, 1],
                     what='it %d img %d' % (it, b))
      o += n
    prev = lab
  return prev


# --------------------------------------------------------------------------
def test_init_grid_matches_reference():
  g = load_golden('a03_init_grid')
  hw = {'k3_17': (17, 17), 'k6_130': (130, 130), 'k6_128': (128, 128),
        'k12_194': (194, 194), 'k32_258': (258, 258), 'k6_513': (513, 513),
        'k4x5_33x29': (33, 29), 'k12_512': (512, 512), 'k2_3': (3, 3)}
  for tag, (h, w) in hw.items():
    ky, kx = g['argk_' + tag].tolist()
    out = ffi().kmeans_init_grid(h, w, ky, kx, DEV)
    assert torch.equal(out.cpu(), g['init_' + tag]), tag


@pytest.mark.parametrize('tag,path', [('tiny', 'mfma_f16x2'), ('small', 'mfma_f16x2_v4p'),
                                      ('k144', 'mfma_f16x2_v3k')])
def test_kmeans_golden_every_iteration(tag, path):
  """tiny: 289 x 10, K = 9 (32x32-tile kernel); small: 1089 x 66, K = 36 -- the shape class of
  the training step and of the roofline kernel (seed pass on kmeans_pass16, E-step passes on
  kmeans_pass64 on pre-converted tiles from the 2nd iteration on, in-LDS split for a single one); k144: the 12x12 many-cluster
  kernel (a single iteration is not worth a pre-conversion: kmeans_big.hip)."""
  g = load_golden('a06_kmeans_' + tag)
  x = g.emb.to(DEV)
  init = g.init.to(DEV)
  off = seg_offsets([x.shape[0]])
  first = {'tiny': 'mfma_f16x2', 'small': 'mfma_f16x2_v3', 'k144': 'mfma_f16x2_bigk'}[tag]
  for it in range(1, g.iterations + 1):
    lab, cent = ffi().kmeans_run(x, off, x.shape[0], g.k, init, it, want_centroids=True)
    assert ffi().kmeans_last_path() == (first if it == 1 else path)
    n_bad = check_labels(lab, g.labels_per_iter[it - 1], g.margin_per_iter[it - 1],
                         what='%s it%d' % (tag, it))
    if n_bad == 0:
      torch.testing.assert_close(cent[0].cpu(), g.protos_per_iter[it - 1], rtol=0, atol=2e-6)
  # zero iterations: labels pass through
  lab0 = ffi().kmeans_run(x, off, x.shape[0], g.k, init, 0)
  assert torch.equal(lab0.cpu(), g.init)


def test_kmeans_empty_cluster_gives_zero_prototype():
  g = load_golden('a06_kmeans_tiny')       # cluster 4 starts empty
  x = g.emb.to(DEV)
  off = seg_offsets([x.shape[0]])
  _, cent = ffi().kmeans_run(x, off, x.shape[0], g.k, g.init.to(DEV), 1, want_centroids=True)
  assert torch.count_nonzero(cent[0, 4]).item() == 0
  assert torch.count_nonzero(g.protos_per_iter[0][4]).item() == 0


def coherent(gen, n, d, side, noise=0.3):
  from tools_synth import coherent_rows
  return coherent_rows(gen, n, d, side, noise)


@pytest.mark.parametrize('d,k,side', [(258, 36, 96), (66, 36, 130), (34, 25, 64), (130, 64, 48),
                                      (18, 9, 40), (320, 36, 33)])
def test_kmeans_vs_oracle_ragged_batch(d, k, side):
  """3 images of different lengths (one not a multiple of the tile, one empty),
  grid initialisation, 10 iterations -- every image against the oracle."""
  gen = torch.Generator().manual_seed(d * 1000 + k)
  ky = int(round(k ** 0.5))
  kx = k // ky
  k = ky * kx
  imgs, inits, lens = [], [], []
  for n_rows in (side * side, 0, side * side - 37):
    if n_rows == 0:
      imgs.append(torch.zeros(0, d)); inits.append(torch.zeros(0, dtype=torch.long)); lens.append(0)
      continue
    e = coherent(gen, 1, d, side)[0]
    init = O.initialize_cluster_labels((ky, kx), (side, side)).view(-1)
    _, init = torch.unique(init, return_inverse=True)
    imgs.append(e[:n_rows]); inits.append(init[:n_rows]); lens.append(n_rows)
  x = torch.cat(imgs).to(DEV)
  init = torch.cat(inits).to(DEV)
  off = seg_offsets(lens)
  lab = ffi().kmeans_run(x, off, side * side, k, init, 10)
  lab2 = ffi().kmeans_run(x, off, side * side, k, init, 10)
  assert torch.equal(lab, lab2), 'k-means must be run-to-run deterministic'
  lab = lab.cpu()
  o = 0
  for e, i0, n_rows in zip(imgs, inits, lens):
    if n_rows == 0:
      continue
    trace = []
    want = O.kmeans_with_initial_labels(e, i0, k, 10, trace=trace)
    got = lab[o:o + n_rows]
    o += n_rows
    mism = (got != want).float().mean().item()
    # a near-tie flip at iteration t legitimately changes later iterations (k-means is
    # chaotic, and the oracle's own fp32 GEMM depends on the host's BLAS threading), so the
    # 10-iteration end-to-end check is statistical -- label agreement and the clustering
    # objective; exactness is pinned per M-/E-step below and by
    # test_kmeans_golden_every_iteration / test_kmeans_assign_exact.
    assert mism < 5e-2, 'd=%d k=%d: %.4f of the labels differ' % (d, k, mism)
    def objective(lab):
      pr = O.calculate_prototypes_from_labels(e, lab, k)
      return (e * pr[lab]).sum(1).mean().item()
    assert abs(objective(got) - objective(want)) < 3e-4
  # every M- and E-step of the first iterations, exactly (no cascade)
  keep = [i for i, n in enumerate(lens) if n]
  check_kmeans_stepwise([imgs[i] for i in keep], [inits[i] for i in keep], [lens[i] for i in keep],
                        k, 3, None)


@pytest.mark.parametrize('d,k', [(258, 36), (66, 36), (34, 25), (130, 64), (32, 7), (256, 16)])
def test_kmeans_preconverted_and_in_kernel_split_agree(d, k):
  """The run path converts X once to the MFMA operand layout (kmeans_preconvert); with
  SPML_KMEANS_NO_PRECONVERT the same pass kernel splits each tile in LDS.  Both must give
  the same labels and prototypes bit for bit (ragged batch: 3 images, one empty, one
  with a partial last tile, unaligned image starts)."""
  gen = torch.Generator().manual_seed(7 * d + k)
  lens = [4099, 0, 2500 + 13]
  x = torch.nn.functional.normalize(torch.randn(sum(lens), d, generator=gen), dim=1).to(DEV)
  init = torch.randint(0, k, (sum(lens),), generator=gen).to(DEV)
  off = seg_offsets(lens)
  # (flag 512: every pass on kmeans_pass16, the kernel that exists in both forms)
  lab_a, cen_a = ffi().kmeans_run(x, off, max(lens), k, init, 4, want_centroids=True, flags=512)
  assert ffi().kmeans_last_path() == 'mfma_f16x2_v3p'
  lab_b, cen_b = ffi().kmeans_run(x, off, max(lens), k, init, 4, want_centroids=True, flags=8)
  assert ffi().kmeans_last_path() == 'mfma_f16x2_v3'
  assert torch.equal(lab_a, lab_b)
  assert torch.equal(cen_a, cen_b)
  # the seed pass writes the converted tiles itself; a separate conversion kernel (flag 16)
  # must leave exactly the same bytes behind
  lab_c, cen_c = ffi().kmeans_run(x, off, max(lens), k, init, 4, want_centroids=True, flags=16 | 512)
  assert ffi().kmeans_last_path() == 'mfma_f16x2_v3p'
  assert torch.equal(lab_a, lab_c) and torch.equal(cen_a, cen_c)
  # the default for K <= 48: E-step passes on the pixel-split 64-pixel-tile kernel.  Same products, other
  # summation order of the M-step: near ties may fall the other way, nothing else
  lab_d, cen_d = ffi().kmeans_run(x, off, max(lens), k, init, 4, want_centroids=True)
  assert ffi().kmeans_last_path() == ('mfma_f16x2_v4p' if k <= 48 else 'mfma_f16x2_v3p')
  lab_e, cen_e = ffi().kmeans_run(x, off, max(lens), k, init, 4, want_centroids=True, flags=16)
  assert torch.equal(lab_d, lab_e) and torch.equal(cen_d, cen_e)
  assert (lab_d != lab_a).float().mean().item() < 2e-3
  assert (cen_d - cen_a).abs().max().item() < 2e-3
  # and the single-iteration run (no pre-conversion by default) against the oracle E-step
  lab_1, cen_1 = ffi().kmeans_run(x, off, max(lens), k, init, 1, want_centroids=True)
  assert ffi().kmeans_last_path() == 'mfma_f16x2_v3'
  o = 0
  for n in lens:
    if n:
      pr = O.calculate_prototypes_from_labels(x[o:o + n].cpu(), init[o:o + n].cpu(), k)
      sim = x[o:o + n].cpu() @ pr.t()
      top2 = sim.topk(2, dim=1).values
      safe = (top2[:, 0] - top2[:, 1]) > 1e-5
      assert torch.equal(lab_1[o:o + n].cpu()[safe], sim.argmax(1)[safe])
    o += n


@pytest.mark.parametrize('d,k', [(34, 144), (66, 144), (66, 256), (32, 100), (130, 128), (64, 65)])
def test_kmeans_many_clusters_mfma_path(d, k):
  """64 < K <= 256 (12x12 grids of the DensePose recipe / full-resolution inference) runs
  on kmeans_pass16k.  Two iterations against the oracle: labels agree except at near
  ties, prototypes of the last E-step agree, the run is deterministic and equals the
  generic fp32 path statistically."""
  gen = torch.Generator().manual_seed(d * 3 + k)
  lens = [6000 + 7, 0, 3000 + 21]
  cent = torch.nn.functional.normalize(torch.randn(k, d, generator=gen), dim=1)
  xs, inits = [], []
  for n in lens:
    own = torch.randint(0, k, (n,), generator=gen)
    xs.append(torch.nn.functional.normalize(cent[own] + 0.35 * torch.randn(n, d, generator=gen), dim=1))
    inits.append((own + (torch.rand(n, generator=gen) < 0.3).long() * torch.randint(0, k, (n,), generator=gen)) % k)
  x = torch.cat(xs).to(DEV)
  init = torch.cat(inits).to(DEV)
  off = seg_offsets(lens)
  lab, cen = ffi().kmeans_run(x, off, max(lens), k, init, 2, want_centroids=True)
  assert ffi().kmeans_last_path() == 'mfma_f16x2_v3k'
  lab2, cen2 = ffi().kmeans_run(x, off, max(lens), k, init, 2, want_centroids=True)
  assert torch.equal(lab, lab2) and torch.equal(cen, cen2)
  lab_g = ffi().kmeans_run(x, off, max(lens), k, init, 2, flags=1)
  assert ffi().kmeans_last_path() == 'generic'
  assert (lab != lab_g).float().mean().item() < 2e-3
  # a single iteration is not worth a pre-conversion (many-cluster kernel of kmeans_big.hip)
  last = check_kmeans_stepwise(xs, inits, lens, k, 1, 'mfma_f16x2_bigk')
  last = check_kmeans_stepwise(xs, inits, lens, k, 3, None)
  assert ffi().kmeans_last_path() == 'mfma_f16x2_v3k'


@pytest.mark.parametrize('d,k,lens', [(258, 144, [6007, 0, 3021]), (258, 144, [64 * 40 + 17]), (256, 80, [4000, 900]),
                                      (258, 70, [4000]), (258, 100, [2500, 2500]), (130, 144, [5000, 31]),
                                      (136, 129, [3000])])
def test_kmeans_many_clusters_on_wide_rows(d, k, lens):
  """64 < K <= 144 at D >= 128 (the 12 x 12 grid of the 513 x 513 x 258 configuration): the wave-split assign kernel
  (kmeans64k.hip: 5, 6, 8 or 9 prototype tiles, a padding tile at K = 100, a ragged last tile, an empty image) + the
  M-only pass of kmeans64.hip.  Every M- and E-step of 3 iterations against the oracle, run-to-run bit-identical,
  statistically equal to the generic fp32 path, and the same labels as the many-cluster kernels the call took before
  (flag 2048) wherever the oracle's top-2 margin is not a near tie."""
  gen = torch.Generator().manual_seed(d * 5 + k)
  cent = torch.nn.functional.normalize(torch.randn(k, d, generator=gen), dim=1)
  xs, inits = [], []
  for n in lens:
    own = torch.randint(0, k, (n,), generator=gen)
    xs.append(torch.nn.functional.normalize(cent[own] + 0.35 * torch.randn(n, d, generator=gen), dim=1))
    inits.append((own + (torch.rand(n, generator=gen) < 0.3).long() * torch.randint(0, k, (n,), generator=gen)) % k)
  x, init, off = torch.cat(xs).to(DEV), torch.cat(inits).to(DEV), seg_offsets(lens)
  assert ffi().kmeans_path_name(x.shape[0], d, k, len(lens), max(lens), 3) == 'mfma_f16x2_v4k'
  assert ffi().kmeans_path_name(x.shape[0], d, k, len(lens), max(lens), 3, flags=2048) == 'mfma_f16x2_bigk'
  check_kmeans_stepwise(xs, inits, lens, k, 3, None, first=2)
  assert ffi().kmeans_last_path() == 'mfma_f16x2_v4k'
  lab, cen = ffi().kmeans_run(x, off, max(lens), k, init, 3, want_centroids=True)
  lab2, cen2 = ffi().kmeans_run(x, off, max(lens), k, init, 3, want_centroids=True)
  assert torch.equal(lab, lab2) and torch.equal(cen, cen2), 'must be run-to-run deterministic'
  lab_g = ffi().kmeans_run(x, off, max(lens), k, init, 3, flags=1)
  assert ffi().kmeans_last_path() == 'generic'
  assert (lab != lab_g).float().mean().item() < 3e-3
  # one fused pass (given prototypes -> labels + raw sums) on the same route; duplicate row -> lowest index
  c = cen.clone()
  c[0, 5] = c[0, 2]
  ws = ffi().kmeans_workspace(x, off, max(lens), k)
  ffi().kmeans_preconvert(x, off, max(lens), k, ws)
  lab_f, sums = ffi().kmeans_fused_pass(x, off, max(lens), c, ws=ws, preconverted=True)
  assert ffi().kmeans_last_path() == 'mfma_f16x2_v4k'
  lab_f, sums = lab_f.cpu(), sums.cpu()
  o = 0
  for b, (xi, n) in enumerate(zip(xs, lens)):
    if n:
      sims = xi @ c[b].cpu().t()
      t2 = sims.topk(2, dim=1).values
      check_labels(lab_f[o:o + n], sims.argmax(1), t2[:, 0] - t2[:, 1], what='fused pass img %d' % b)
      if b == 0:
        assert (lab_f[o:o + n] == 5).sum().item() == 0
      ref = torch.zeros(k, d).index_add_(0, lab_f[o:o + n], xi)
      torch.testing.assert_close(sums[b], ref, rtol=0, atol=2e-5 * max(1.0, ref.abs().max().item()))
    o += n


@pytest.mark.parametrize('d,k,lens', [(514, 1024, [6000]), (130, 300, [4097, 0, 777]),
                                      (258, 512, [3001, 2000]), (66, 1024, [5000]),
                                      (514, 70, [300, 129]), (34, 4097, [9000]), (515, 96, [1000])])
def test_kmeans_many_clusters_bigk_path(d, k, lens):
  """More centroids than the tile kernels take (e.g. the 32x32 stress configuration at
  D = 514): kmeans_big.hip -- pixel-stationary MFMA E-step with a running arg-max (the
  [P,K] similarity never exists), counting-sort + fixed-point gather M-step.  Every M- and
  E-step of 3 iterations against the oracle, run-to-run bit-identical, and against the
  generic fp32 kernel (flag 1) on the same input."""
  gen = torch.Generator().manual_seed(d + k)
  cent = torch.nn.functional.normalize(torch.randn(k, d, generator=gen), dim=1)
  xs, inits = [], []
  for p in lens:
    own = torch.randint(0, k, (p,), generator=gen)
    xs.append(torch.nn.functional.normalize(cent[own] + 0.3 * torch.randn(p, d, generator=gen), dim=1))
    inits.append((own + (torch.rand(p, generator=gen) < 0.3).long() * torch.randint(0, k, (p,), generator=gen)) % k)
  x, init, off = torch.cat(xs).to(DEV), torch.cat(inits).to(DEV), seg_offsets(lens)
  check_kmeans_stepwise(xs, inits, lens, k, 3, 'mfma_f16x2_bigk')
  lab, cen = ffi().kmeans_run(x, off, max(lens), k, init, 3, want_centroids=True)
  lab2, cen2 = ffi().kmeans_run(x, off, max(lens), k, init, 3, want_centroids=True)
  assert torch.equal(lab, lab2) and torch.equal(cen, cen2), 'must be run-to-run deterministic'
  lab_g = ffi().kmeans_run(x, off, max(lens), k, init, 3, flags=1)
  assert ffi().kmeans_last_path() == 'generic'
  assert (lab != lab_g).float().mean().item() < 3e-3
  # from K = 256 on the E-step first screens on the hi halves and scores only the ambiguous
  # pixels exactly: the result must be identical to scoring every pixel exactly (flag 64)
  lab_ns, cen_ns = ffi().kmeans_run(x, off, max(lens), k, init, 3, want_centroids=True, flags=64)
  assert torch.equal(lab, lab_ns) and torch.equal(cen, cen_ns)
  # the stand-alone assign entry point takes the same route; duplicate row -> lowest index,
  # zero prototype takes part
  c = cen.clone()
  c[0, 5] = c[0, 2]
  c[0, 7] = 0.0
  lab_a = ffi().kmeans_assign(x, off, max(lens), c).cpu()
  assert ffi().kmeans_last_path() == 'mfma_f16x2_bigk'
  o = 0
  for b, (xi, n) in enumerate(zip(xs, lens)):
    if n:
      sims = xi @ c[b].cpu().t()
      t2 = sims.topk(2, dim=1).values
      check_labels(lab_a[o:o + n], sims.argmax(1), t2[:, 0] - t2[:, 1], what='assign img %d' % b)
      if b == 0:
        assert (lab_a[o:o + n] == 5).sum().item() == 0
    o += n


@pytest.mark.parametrize('d,k,p', [(258, 36, 20011), (66, 36, 16900), (34, 144, 5000),
                                   (514, 64, 3000), (64, 10, 4097), (66, 33, 1000)])
def test_kmeans_assign_exact(d, k, p):
  """One E-step against given prototypes: labels identical to the oracle except
  at near ties (oracle margin < 1e-5); ties resolve to the lowest index."""
  gen = torch.Generator().manual_seed(p)
  x = O.normalize_embedding(torch.randn(p, d, generator=gen))
  c = O.normalize_embedding(torch.randn(2, k, d, generator=gen))
  c[1, 3] = c[1, 1]                 # exact duplicate -> tie -> lowest index wins
  c[0, 2] = 0.0                     # zero prototype (empty cluster) takes part
  lens = [p // 3, p - p // 3]
  off = seg_offsets(lens)
  lab = ffi().kmeans_assign(x.to(DEV), off, max(lens), c.to(DEV)).cpu()
  o = 0
  for b, n_rows in enumerate(lens):
    sims = x[o:o + n_rows] @ c[b].t()
    want = sims.argmax(1)
    t2 = sims.topk(2, dim=1).values
    check_labels(lab[o:o + n_rows], want, t2[:, 0] - t2[:, 1], what='img %d' % b)
    if b == 1:
      assert (lab[o:o + n_rows] == 3).sum().item() == 0
    o += n_rows


# --------------------------------------------------------------------------
def k1_oracle(emb, loc, keep):
  e = O.normalize_embedding(emb.permute(0, 2, 3, 1).contiguous())
  n, h, w, c = e.shape
  el = O.normalize_embedding(torch.cat([e, loc], -1))
  return e.reshape(-1, c)[keep], el.reshape(-1, c + loc.shape[-1])[keep]


@pytest.mark.parametrize('n,c,h,w', [(2, 8, 17, 17), (2, 64, 33, 35), (1, 256, 20, 23),
                                     (1, 512, 9, 11), (3, 32, 1, 70)])
def test_k1_forward_backward(n, c, h, w):
  gen = torch.Generator().manual_seed(c)
  emb = torch.randn(n, c, h, w, generator=gen)
  emb[0, :, 0, 0] = 0.0             # zero vector -> eps branch
  loc = (O.generate_location_features((h, w), 'float') - 0.5).view(1, h, w, 2).expand(
      n, h, w, 2).contiguous()
  mask = torch.rand(n * h * w, generator=gen) > 0.2
  keep = mask.nonzero().view(-1)
  row_map = torch.full((n * h * w,), -1, dtype=torch.long)
  row_map[keep] = torch.arange(keep.numel())
  emb_r = emb.clone().requires_grad_(True)
  we, wl = k1_oracle(emb_r, loc, keep)
  g1 = torch.randn(we.shape, generator=gen)
  g2 = torch.randn(wl.shape, generator=gen)
  ((we * g1).sum() + (wl * g2).sum()).backward()

  F = ffi()
  oe, ol = F.normalize_concat_loc(emb.to(DEV), loc.to(DEV), row_map.to(DEV), keep.numel())
  torch.testing.assert_close(oe.cpu(), we.detach(), rtol=0, atol=1e-6)
  torch.testing.assert_close(ol.cpu(), wl.detach(), rtol=0, atol=1e-6)
  # in-kernel location features == the reference's linspace grid
  oe2, ol2 = F.normalize_concat_loc(emb.to(DEV), None, row_map.to(DEV), keep.numel())
  torch.testing.assert_close(ol2.cpu(), wl.detach(), rtol=0, atol=1e-6)
  # identity row map
  oe3, ol3 = F.normalize_concat_loc(emb.to(DEV), loc.to(DEV))
  full = torch.arange(n * h * w)
  fe, fl = k1_oracle(emb, loc, full)
  torch.testing.assert_close(oe3.cpu(), fe, rtol=0, atol=1e-6)
  torch.testing.assert_close(ol3.cpu(), fl, rtol=0, atol=1e-6)
  d = F.normalize_concat_loc_bwd(emb.to(DEV), loc.to(DEV), row_map.to(DEV), g1.to(DEV),
                                 g2.to(DEV)).cpu()
  ref = emb_r.grad
  # the zero-norm pixel has gradient g/eps = O(1e12): compare relatively
  scale = ref.abs().amax(dim=1, keepdim=True).clamp(min=1.0)
  assert ((d - ref).abs() / scale).max().item() < 2e-5


@pytest.mark.parametrize('n,c,h,w,nl', [(2, 64, 33, 35, 2), (1, 256, 20, 23, 2), (1, 512, 9, 11, 2), (3, 32, 5, 70, 2),
                                        (2, 16, 13, 11, 2), (2, 128, 12, 19, 2), (2, 32, 21, 19, 5), (1, 64, 9, 70, 5)])
def test_k1_channels_last_map(n, c, h, w, nl):
  """K1 on a channels-last map (what the NHWC backbone hands over): the row-wise kernels, forward and
  backward against the oracle chain and against the NCHW kernels, ignore-pixel compaction, eps branch,
  in-kernel location features; the gradient comes back channels-last."""
  gen = torch.Generator().manual_seed(c * 3 + nl)
  emb = torch.randn(n, c, h, w, generator=gen)
  emb[0, :, 0, 0] = 0.0
  if nl == 2:
    loc = (O.generate_location_features((h, w), 'float') - 0.5).view(1, h, w, 2).expand(n, h, w, 2).contiguous()
  else:
    loc = torch.randn(n, h, w, nl, generator=gen) * 0.4
  keep = (torch.rand(n * h * w, generator=gen) > 0.2).nonzero().view(-1)
  row_map = torch.full((n * h * w,), -1, dtype=torch.long)
  row_map[keep] = torch.arange(keep.numel())
  emb_r = emb.clone().requires_grad_(True)
  we, wl = k1_oracle(emb_r, loc, keep)
  g1 = torch.randn(we.shape, generator=gen)
  g2 = torch.randn(wl.shape, generator=gen)
  ((we * g1).sum() + (wl * g2).sum()).backward()
  F = ffi()
  cl = emb.to(DEV).contiguous(memory_format=torch.channels_last)
  assert F.k1_channels_last(cl, nl) and not F.k1_channels_last(emb.to(DEV), nl)
  oe, ol = F.normalize_concat_loc(cl, loc.to(DEV), row_map.to(DEV), keep.numel())
  torch.testing.assert_close(oe.cpu(), we.detach(), rtol=0, atol=1e-6)
  torch.testing.assert_close(ol.cpu(), wl.detach(), rtol=0, atol=1e-6)
  if nl == 2:                                # location features generated in the kernel
    _, ol2 = F.normalize_concat_loc(cl, None, row_map.to(DEV), keep.numel())
    torch.testing.assert_close(ol2.cpu(), wl.detach(), rtol=0, atol=1e-6)
  d = F.normalize_concat_loc_bwd(cl, loc.to(DEV), row_map.to(DEV), g1.to(DEV), g2.to(DEV))
  assert d.is_contiguous(memory_format=torch.channels_last)
  ref = emb_r.grad
  scale = ref.abs().amax(dim=1, keepdim=True).clamp(min=1.0)
  assert ((d.cpu() - ref).abs() / scale).max().item() < 2e-5
  d_nchw = F.normalize_concat_loc_bwd(emb.to(DEV), loc.to(DEV), row_map.to(DEV), g1.to(DEV), g2.to(DEV))
  assert ((d.cpu() - d_nchw.cpu()).abs() / scale).max().item() < 2e-5
  # through autograd: the op takes the channels-last tensor without a layout copy
  from spml_amd import ops
  x = cl.clone().requires_grad_(True)
  a, b = ops.normalize_concat_loc(x, loc.to(DEV), row_map.to(DEV), keep.numel())
  ((a * g1.to(DEV)).sum() + (b * g2.to(DEV)).sum()).backward()
  assert ((x.grad.cpu() - ref).abs() / scale).max().item() < 2e-5


@pytest.mark.parametrize('n,c,h,w,nl', [(2, 32, 21, 19, 5), (1, 64, 9, 70, 5), (2, 8, 17, 17, 1),
                                        (1, 16, 12, 12, 8)])
def test_k1_with_colour_and_location_features(n, c, h, w, nl):
  """K1 with L local-feature channels (the DensePose recipe appends (y, x) + 3 colours:
  resnet_pspnet_densepose.py:37-38), forward and backward against the oracle chain."""
  gen = torch.Generator().manual_seed(c + nl)
  emb = torch.randn(n, c, h, w, generator=gen)
  emb[0, :, 1, 2] = 0.0
  loc = torch.randn(n, h, w, nl, generator=gen) * 0.4
  mask = torch.rand(n * h * w, generator=gen) > 0.15
  keep = mask.nonzero().view(-1)
  row_map = torch.full((n * h * w,), -1, dtype=torch.long)
  row_map[keep] = torch.arange(keep.numel())
  emb_r = emb.clone().requires_grad_(True)
  we, wl = k1_oracle(emb_r, loc, keep)
  assert wl.shape[1] == c + nl
  g1 = torch.randn(we.shape, generator=gen)
  g2 = torch.randn(wl.shape, generator=gen)
  ((we * g1).sum() + (wl * g2).sum()).backward()
  F = ffi()
  oe, ol = F.normalize_concat_loc(emb.to(DEV), loc.to(DEV), row_map.to(DEV), keep.numel())
  torch.testing.assert_close(oe.cpu(), we.detach(), rtol=0, atol=1e-6)
  torch.testing.assert_close(ol.cpu(), wl.detach(), rtol=0, atol=1e-6)
  d = F.normalize_concat_loc_bwd(emb.to(DEV), loc.to(DEV), row_map.to(DEV), g1.to(DEV),
                                 g2.to(DEV)).cpu()
  ref = emb_r.grad
  scale = ref.abs().amax(dim=1, keepdim=True).clamp(min=1.0)
  assert ((d - ref).abs() / scale).max().item() < 2e-5
  # no in-kernel generation of local features other than the 2 location channels
  rc = F.lib().spml_normalize_concat_local_f32(F.ptr(emb.to(DEV)), n, c, h, w, None, 5, None,
                                               F.ptr(oe), None, F.stream_ptr())
  assert rc == -1


@pytest.mark.parametrize('d,k', [(37, 36), (37, 144), (69, 36), (33, 16), (136, 25)])
def test_kmeans_with_colour_channels_runs_on_the_mfma_path(d, k):
  """D = C + 5 (odd row length, 5-channel tail) goes through the separate conversion
  kernel onto the 16x16-tile passes; against the oracle over a ragged batch."""
  gen = torch.Generator().manual_seed(d * 7 + k)
  lens = [3000 + 5, 1777]
  cent = torch.nn.functional.normalize(torch.randn(k, d, generator=gen), dim=1)
  xs, inits = [], []
  for n in lens:
    own = torch.randint(0, k, (n,), generator=gen)
    xs.append(torch.nn.functional.normalize(cent[own] + 0.3 * torch.randn(n, d, generator=gen), dim=1))
    inits.append((own + (torch.rand(n, generator=gen) < 0.3).long() * torch.randint(0, k, (n,), generator=gen)) % k)
  x = torch.cat(xs).to(DEV)
  init = torch.cat(inits).to(DEV)
  off = seg_offsets(lens)
  lab, cen = ffi().kmeans_run(x, off, max(lens), k, init, 3, want_centroids=True)
  assert ffi().kmeans_last_path() == ('mfma_f16x2_v3k' if k > 64 else 'mfma_f16x2_v4p' if k <= 48 else 'mfma_f16x2_v3p')
  assert torch.equal(lab, ffi().kmeans_run(x, off, max(lens), k, init, 3))
  last = check_kmeans_stepwise(xs, inits, lens, k, 3, None)
  assert torch.equal(last, lab.cpu())


def test_normalize_rows():
  g = load_golden('a01_normalize')
  y = ffi().normalize_rows(g.x.to(DEV)).cpu()
  torch.testing.assert_close(y, g.y, rtol=0, atol=1e-6)


@pytest.mark.parametrize('p,d,m', [(5000, 64, 300), (3000, 66, 40), (1000, 258, 7), (700, 10, 700)])
def test_segment_prototypes_fwd_bwd(p, d, m):
  gen = torch.Generator().manual_seed(p + d)
  x = O.normalize_embedding(torch.randn(p, d, generator=gen))
  ids = torch.randint(0, m, (p // 50 + 1,), generator=gen).repeat_interleave(50)[:p]
  ids[ids == 1] = 0                 # segment 1 empty -> zero prototype
  xr = x.clone().requires_grad_(True)
  want = O.calculate_prototypes_from_labels(xr, ids, m)
  gsel = torch.randn(m, d, generator=gen)
  (want * gsel).sum().backward()
  F = ffi()
  protos, sums = F.segment_sum_normalize(x.to(DEV), ids.to(DEV), m)
  torch.testing.assert_close(protos.cpu(), want.detach(), rtol=0, atol=2e-6)
  dx = F.segment_sum_normalize_bwd(gsel.to(DEV), sums, ids.to(DEV), p).cpu()
  scale = xr.grad.abs().max().item()
  assert (dx - xr.grad).abs().max().item() < 1e-5 * max(scale, 1.0)


# --------------------------------------------------------------------------
def tags_to_mask(tags):
  """multi-hot [N,T] int64 -> packed 64-bit tag sets."""
  w = (2 ** torch.arange(tags.shape[1], dtype=torch.long)).view(1, -1)
  return (tags.long() * w).sum(1)


def ill_conditioned_pixels(emb, own, px_code, protos, pr_code, kappa, mode):
  """Pixels where the reference formula cancels: `pos = sum_same - own_sim` (loss.py:61-70) with pos < own / 256;
  there the value depends on the summation order inside the reference itself."""
  sim = ((emb.double() @ protos.double().t()) * kappa).exp()
  own_s = sim.gather(1, own.view(-1, 1)).view(-1)
  if mode & 1:
    same = (px_code.view(-1, 1) & pr_code.view(1, -1)) != 0
  else:
    same = px_code.view(-1, 1) == pr_code.view(1, -1)
  pos = (sim * same).sum(1) - own_s
  return (pos > 0) & (pos < own_s / 256)


def nll_check(emb, own, px_code, protos, pr_code, kappa, mode, want_nll, d_nll, want_de, want_dp):
  """Per-pixel NLL + gradients vs the oracle.  `pos = sum_same - own_sim` is a
  genuine fp32 cancellation in the reference: where own_sim dwarfs
  the other positives the value depends on summation order, so a handful of
  ill-conditioned pixels may deviate more; the mean loss must agree to 1e-5.  Gradients: every element within
  1e-4 of the gradient's scale outside the ill-conditioned pixels (1e-3 inside, and for the prototype gradient
  when such pixels exist: they feed many prototypes), 2e-5 on average."""
  F = ffi()
  ill = ill_conditioned_pixels(emb, own, px_code, protos, pr_code, kappa, mode)
  nll, stats = F.segsort_nll_fwd(emb.to(DEV), own.to(DEV), px_code.to(DEV), protos.to(DEV),
                                 pr_code.to(DEV), kappa, mode)
  got, want = nll.cpu(), want_nll.view(-1)
  rel = (got - want).abs() / want.abs().clamp(min=1.0)
  assert (rel > 2e-5).float().mean().item() <= 5e-3, 'too many pixels off: %g' % (rel > 2e-5).float().mean()
  assert rel.max().item() < 5e-3
  assert abs(got.mean().item() - want.mean().item()) <= 1e-5 * max(1.0, abs(want.mean().item()))
  de, dp = F.segsort_nll_bwd(emb.to(DEV), own.to(DEV), px_code.to(DEV), protos.to(DEV),
                             pr_code.to(DEV), kappa, mode, stats, d_nll.to(DEV))
  for g_, w_, name in ((de.cpu(), want_de, 'd_emb'), (dp.cpu(), want_dp, 'd_protos')):
    scale = w_.abs().max().item()
    err = (g_ - w_).abs()
    assert err.max().item() <= 1e-3 * scale + 1e-12, '%s: max err %.3g vs scale %.3g' % (
        name, err.max().item(), scale)
    assert err.mean().item() <= 2e-5 * scale + 1e-12, '%s: mean err %.3g vs scale %.3g' % (
        name, err.mean().item(), scale)
    good = err[~ill] if name == 'd_emb' else (err if not ill.any() else err[:0])
    if good.numel():
      assert good.max().item() <= 1e-4 * scale + 1e-12, '%s: max err %.3g vs scale %.3g outside the %d ' \
          'ill-conditioned pixels' % (name, good.max().item(), scale, int(ill.sum()))


@pytest.mark.parametrize('tag', ['tiny', 'small', 'loc'])
def test_nll_golden(tag):
  g = load_golden('a09_loss_' + tag)
  p = g.emb.shape[0]
  d_nll = torch.full((p,), 1.0 / p)
  nll_check(g.emb, g.own, g.sem, g.protos, g.p_sem, g.kappa, 0, g.nll, d_nll, g.d_emb,
            g.d_protos)
  nll_check(g.emb, g.own, tags_to_mask(g.tags), g.protos, tags_to_mask(g.p_tags), g.kappa, 1,
            g.set_nll, d_nll, g.set_d_emb, g.set_d_protos)


@pytest.mark.parametrize('p,m,d,kappa', [(5000, 700, 64, 12.0), (3001, 95, 66, 16.0),
                                         (1000, 33, 34, 6.0), (2000, 300, 130, 8.0),
                                         (777, 40, 258, 10.0), (64, 5, 16, 6.0),
                                         (700, 130, 514, 12.0), (300, 70, 512, 6.0),
                                         (257, 33, 300, 10.0), (130, 97, 528, 8.0)])
def test_nll_vs_oracle_weighted_grad(p, m, d, kappa):
  gen = torch.Generator().manual_seed(p + m)
  protos = O.normalize_embedding(torch.randn(m, d, generator=gen))
  own = torch.randint(0, m, (p,), generator=gen)
  emb = O.normalize_embedding(protos[own] + 0.8 * torch.randn(p, d, generator=gen))
  p_sem = torch.randint(0, 21, (m,), generator=gen)
  sem = p_sem[own].clone()
  flip = torch.rand(p, generator=gen) < 0.2
  sem[flip] = torch.randint(0, 21, (int(flip.sum()),), generator=gen)
  wgt = torch.rand(p, generator=gen) * 1e-4          # non-uniform upstream gradient
  e = emb.clone().requires_grad_(True)
  pr = protos.clone().requires_grad_(True)
  nll = O.segsort_nll(e, sem, own, pr, p_sem, kappa)
  (nll.view(-1) * wgt).sum().backward()
  nll_check(emb, own, sem, protos, p_sem, kappa, 0, nll.detach(), wgt, e.grad, pr.grad)
  # 32-bit codes: the v2 / v3 kernels for narrow embeddings (a fifth of the pixels have an own prototype of
  # another class: their T exceeds 1 and takes the per-pixel scale of nll_t_scale)
  nll_check(emb, own, sem, protos, p_sem, kappa, 0 | 4, nll.detach(), wgt, e.grad, pr.grad)


def test_topk_golden_and_masked():
  g = load_golden('a11_topk')
  F = ffi()
  for k, want in ((5, g.top5), (20, g.top20)):
    idx, val = F.topk_affinity(g.q.to(DEV), g.pr.to(DEV), k)
    assert torch.equal(g.prl[idx.cpu()], want)
    ref = (g.q @ g.pr.t()).topk(k, dim=1)
    torch.testing.assert_close(val.cpu(), ref.values, rtol=0, atol=2e-6)
  idx, _ = F.topk_affinity(g.pr.to(DEV), g.pr.to(DEV), 5)
  assert torch.equal(g.prl[idx.cpu()], g.top_self)
  # masked variant (models/utils.py:198-214): only same-group, valid prototypes compete
  gen = torch.Generator().manual_seed(5)
  qg = torch.randint(0, 4, (g.q.shape[0],), generator=gen)
  pg = torch.randint(0, 4, (g.pr.shape[0],), generator=gen)
  valid = (torch.rand(g.pr.shape[0], generator=gen) > 0.3)
  idx, val = F.topk_affinity(g.q.to(DEV), g.pr.to(DEV), 3, qg.to(DEV), pg.to(DEV),
                             valid.to(torch.uint8).to(DEV), -2.0)
  d = g.q @ g.pr.t()
  ok = (qg.view(-1, 1) == pg.view(1, -1)) & valid.view(1, -1)
  ref = torch.where(ok, d, torch.full_like(d, -2.0)).topk(3, dim=1)
  torch.testing.assert_close(val.cpu(), ref.values, rtol=0, atol=2e-6)
  real = ref.values > -1.5
  assert torch.equal(idx.cpu()[real], ref.indices[real])


# --------------------------------------------------------------------------
@pytest.mark.parametrize('d,k,lens,pre', [(258, 36, [5000, 0, 777], True), (258, 36, [5000, 0, 777], False),
                                          (66, 36, [4099, 2500], True), (34, 144, [3000], True),
                                          (514, 1024, [2500], False), (40, 20, [999], False),
                                          (400, 7, [300], False)])
def test_kmeans_fused_pass_export(d, k, lens, pre):
  """spml_kmeans_fused_pass_f32: centroids in -> labels + raw sums of X by the new
  labels, on pre-converted tiles (spml_kmeans_preconvert_f32) or splitting inside the pass.
  Labels against the oracle E-step (near-tie rule), sums against the oracle's scatter-add
  on OUR labels; normalising them gives the M-step of the reference."""
  gen = torch.Generator().manual_seed(d + k + len(lens))
  n_img = len(lens)
  xs = [O.normalize_embedding(torch.randn(n, d, generator=gen)) for n in lens]
  cent = O.normalize_embedding(torch.randn(n_img, k, d, generator=gen))
  cent[0, 1] = 0.0                          # an empty cluster's zero prototype takes part
  x, off = torch.cat(xs).to(DEV), seg_offsets(lens)
  F = ffi()
  ws = F.kmeans_workspace(x, off, max(lens), k)
  if pre:
    F.kmeans_preconvert(x, off, max(lens), k, ws)
  lab, sums = F.kmeans_fused_pass(x, off, max(lens), cent.to(DEV), ws=ws, preconverted=pre)
  want_path = F.kmeans_path_name(x.shape[0], d, k, n_img, max(lens), 1, True, 32 if pre else 0)
  assert F.kmeans_last_path() == want_path
  if d == 258 and k == 36:
    assert want_path == ('mfma_f16x2_v4p' if pre else 'mfma_f16x2_v3')
  lab, sums = lab.cpu(), sums.cpu()
  o = 0
  for b, (xi, n) in enumerate(zip(xs, lens)):
    if n:
      sims = xi @ cent[b].t()
      t2 = sims.topk(2, dim=1).values
      check_labels(lab[o:o + n], sims.argmax(1), t2[:, 0] - t2[:, 1], what='img %d' % b)
      raw = torch.zeros(k, d).index_add_(0, lab[o:o + n], xi)
      torch.testing.assert_close(sums[b], raw, rtol=1e-5, atol=2e-5)
      torch.testing.assert_close(F.normalize_rows(sums[b].to(DEV)).cpu(),
                                 O.calculate_prototypes_from_labels(xi, lab[o:o + n], k),
                                 rtol=0, atol=2e-6)
    else:
      assert torch.count_nonzero(sums[b]).item() == 0
    o += n


def test_kmeans_profiled_run_matches_plain_run():
  """spml_kmeans_run_profiled_f32: same labels as the plain run; one duration per pass from
  the per-workgroup device time stamps, all positive and of a plausible size."""
  gen = torch.Generator().manual_seed(3)
  lens = [20000, 12345]
  x = O.normalize_embedding(torch.randn(sum(lens), 66, generator=gen)).to(DEV)
  init = torch.randint(0, 36, (sum(lens),), generator=gen).to(DEV)
  off = seg_offsets(lens)
  lab = ffi().kmeans_run(x, off, max(lens), 36, init, 4)
  lab_p, dur = ffi().kmeans_run_profiled(x, off, max(lens), 36, init, 4)
  assert torch.equal(lab, lab_p)
  assert dur.shape == (5,) and (dur > 1.0).all() and (dur < 5000.0).all(), dur
  with pytest.raises(ffi().SpmlHipError):
    ffi().kmeans_run_profiled(x, off, max(lens), 1024, init, 4)      # not a tile-kernel shape


@pytest.mark.parametrize('p,m,d', [(600, 90, 514), (200, 40, 400)])
def test_set_nll_wide_embeddings(p, m, d):
  """Set-SegSort NLL (tag-set predicate) on the wide-embedding kernels (272 < D <= 528: the
  backward runs once per 96-channel chunk), forward + both gradients against the oracle."""
  gen = torch.Generator().manual_seed(p + d)
  protos = O.normalize_embedding(torch.randn(m, d, generator=gen))
  own = torch.randint(0, m, (p,), generator=gen)
  emb = O.normalize_embedding(protos[own] + 0.8 * torch.randn(p, d, generator=gen))
  p_tags = (torch.rand(m, 20, generator=gen) < 0.15).long()
  p_tags[torch.arange(m), torch.randint(0, 20, (m,), generator=gen)] = 1
  tags = p_tags[own].clone()
  wgt = torch.rand(p, generator=gen) * 1e-3
  e = emb.clone().requires_grad_(True)
  pr = protos.clone().requires_grad_(True)
  nll = O.set_segsort_nll(e, tags, own, pr, p_tags, 8.0)
  (nll.view(-1) * wgt).sum().backward()
  nll_check(emb, own, tags_to_mask(tags), protos, tags_to_mask(p_tags), 8.0, 1, nll.detach(), wgt,
            e.grad, pr.grad)


def test_wide_nll_backward_in_strips_is_identical(monkeypatch):
  """Wide embeddings: the backward keeps the weight tiles of one STRIP of pixel tiles at a time
  (workspace bounded by SPML_NLL_TCACHE_MB, not P x M).  Three strips (1-MB caches) must give the
  gradients of the single-strip run: dE bit for bit (strips are disjoint pixel ranges), dPr up to the
  order of its atomics."""
  gen = torch.Generator().manual_seed(5)
  p, m, d = 6000, 90, 514
  protos = O.normalize_embedding(torch.randn(m, d, generator=gen)).to(DEV)
  own = torch.randint(0, m, (p,), generator=gen).to(DEV)
  emb = O.normalize_embedding(protos[own].cpu() + 0.8 * torch.randn(p, d, generator=gen)).to(DEV)
  pr_code = torch.randint(1, 2 ** 20, (m,), generator=gen).to(DEV)
  px_code = pr_code[own]
  g = (torch.rand(p, generator=gen) / p).to(DEV)
  F = ffi()
  _, stats = F.segsort_nll_fwd(emb, own, px_code, protos, pr_code, 8.0, 1)
  de0, dp0 = F.segsort_nll_bwd(emb, own, px_code, protos, pr_code, 8.0, 1, stats, g)
  monkeypatch.setenv('SPML_NLL_TCACHE_MB', '1')
  assert F.lib().spml_segsort_nll_workspace_bytes(p, m, d) < 40 * 2 ** 20
  de1, dp1 = F.segsort_nll_bwd(emb, own, px_code, protos, pr_code, 8.0, 1, stats, g)
  assert torch.equal(de0, de1)
  torch.testing.assert_close(dp0, dp1, rtol=1e-5, atol=2e-5 * float(dp0.abs().max()))


@pytest.mark.parametrize('p,m,d,run', [(3000, 700, 64, 100), (1500, 4000, 34, 997), (700, 3100, 66, 64)])
def test_nll_image_major_codes_uniform_tiles(p, m, d, run):
  """Prototypes in runs that share one tag set (image-major prototypes of the co-occurrence term): the
  v2 forward kernel takes its one-predicate-per-tile epilogue for the 32-prototype tiles inside a run
  and the general one for tiles that straddle a boundary or are ragged; more than one prototype chunk
  (m > 3072).  Forward and both gradients against the oracle, 32-bit codes."""
  gen = torch.Generator().manual_seed(p + m)
  protos = O.normalize_embedding(torch.randn(m, d, generator=gen))
  own = torch.randint(0, m, (p,), generator=gen)
  emb = O.normalize_embedding(protos[own] + 0.8 * torch.randn(p, d, generator=gen))
  n_run = (m + run - 1) // run
  run_tags = torch.zeros(n_run, 20, dtype=torch.long)                 # two tags per run: most runs are disjoint
  run_tags.scatter_(1, torch.stack([torch.randperm(20, generator=gen)[:2] for _ in range(n_run)]), 1)
  p_tags = run_tags.repeat_interleave(run, dim=0)[:m]
  tags = p_tags[own]
  wgt = torch.rand(p, generator=gen) / p
  e = emb.clone().requires_grad_(True)
  pr = protos.clone().requires_grad_(True)
  nll = O.set_segsort_nll(e, tags, own, pr, p_tags, 12.0)
  (nll.view(-1) * wgt).sum().backward()
  nll_check(emb, own, tags_to_mask(tags), protos, tags_to_mask(p_tags), 12.0, 1 | 4, nll.detach(), wgt,
            e.grad, pr.grad)


@pytest.mark.parametrize('mode', [0, 1])
@pytest.mark.parametrize('d', [64, 66, 514])
def test_nll_32_bit_code_path_is_identical(mode, d):
  """SPML_NLL_CODE32 (codes promised to fit in 32 bits) changes the width of the positive-set
  predicate.  Wide embeddings (D = 514) run the same kernels either way: bit-identical.  Narrow ones
  take the round-3 forward kernel (`nll_fwd2`: one accumulator per product, chunked prototype range),
  which sums in another order: values agree to fp32 rounding, and so do the gradients computed from
  the two sets of saved statistics."""
  gen = torch.Generator().manual_seed(d + mode)
  p, m = 3000, 333
  protos = O.normalize_embedding(torch.randn(m, d, generator=gen)).to(DEV)
  own = torch.randint(0, m, (p,), generator=gen).to(DEV)
  emb = O.normalize_embedding(protos[own].cpu() + 0.8 * torch.randn(p, d, generator=gen)).to(DEV)
  if mode == 0:
    pr_code = torch.randint(0, 21, (m,), generator=gen).to(DEV)
  else:
    pr_code = torch.randint(1, 2 ** 20, (m,), generator=gen).to(DEV)
  px_code = pr_code[own]
  g = (torch.rand(p, generator=gen) / p).to(DEV)
  F = ffi()
  n0, s0 = F.segsort_nll_fwd(emb, own, px_code, protos, pr_code, 12.0, mode)
  n1, s1 = F.segsort_nll_fwd(emb, own, px_code, protos, pr_code, 12.0, mode | 4)
  de0, dp0 = F.segsort_nll_bwd(emb, own, px_code, protos, pr_code, 12.0, mode, s0, g)
  de1, dp1 = F.segsort_nll_bwd(emb, own, px_code, protos, pr_code, 12.0, mode | 4, s1, g)
  if d > 80:
    assert torch.equal(n0, n1) and torch.equal(s0, s1)
    assert torch.equal(de0, de1)
  else:
    torch.testing.assert_close(n0, n1, rtol=0, atol=5e-6)
    torch.testing.assert_close(s0[:, :3], s1[:, :3], rtol=5e-6, atol=0)
    assert torch.equal(s0[:, 3], s1[:, 3])
    torch.testing.assert_close(de0, de1, rtol=0, atol=2e-5 * float(de0.abs().max()))
  torch.testing.assert_close(dp0, dp1, rtol=1e-5, atol=2e-5 * float(dp0.abs().max()))   # fp32 atomics reorder the sum


def test_kmeans_assign_input_domain():
  """spml_kmeans_assign_f32 documents its domain (include/spml_hip.h): the MFMA paths split
  operands into two f16 halves, so centroids need not be normalised but every element must
  stay below 65504 in magnitude; SPML_KMEANS_FORCE_GENERIC (flag 1) takes arbitrary fp32.
  Scaled centroids (x100, x1e-3, per-row scales) keep the oracle's arg-max; out-of-range
  values are handled by the generic path."""
  gen = torch.Generator().manual_seed(9)
  p, d, k = 5000, 66, 36
  x = O.normalize_embedding(torch.randn(p, d, generator=gen))
  c = O.normalize_embedding(torch.randn(1, k, d, generator=gen))
  off = seg_offsets([p])
  for scale in (torch.full((k, 1), 100.0), torch.full((k, 1), 1e-3),
                torch.logspace(-2, 2, k).view(k, 1)):
    cs = (c[0] * scale).unsqueeze(0)
    lab = ffi().kmeans_assign(x.to(DEV), off, p, cs.to(DEV)).cpu()
    assert ffi().kmeans_last_path() == 'mfma_f16x2_v3'
    sims = x @ cs[0].t()
    t2 = sims.topk(2, dim=1).values
    # near-tie rule relative to the magnitude of the scores
    check_labels(lab, sims.argmax(1), (t2[:, 0] - t2[:, 1]) / sims.abs().max(), tol=1e-5)
  big = (c[0] * 1e6).unsqueeze(0)                      # beyond the f16 range: generic path
  lab = ffi().kmeans_assign(x.to(DEV), off, p, big.to(DEV), flags=1).cpu()
  assert ffi().kmeans_last_path() == 'generic'
  sims = x @ big[0].t()
  t2 = sims.topk(2, dim=1).values
  check_labels(lab, sims.argmax(1), (t2[:, 0] - t2[:, 1]) / sims.abs().max(), tol=1e-5)


@pytest.mark.parametrize('p,span,seed', [(0, 10, 0), (1, 10, 1), (5000, 7, 2), (270400, 17000, 3),
                                         (100000, 2 ** 62, 4), (4096, 4096 * 8, 5)])
def test_relabel_unique_matches_torch_unique(p, span, seed):
  """spml_relabel_unique_i64 against torch.unique(return_inverse=True): sorted distinct keys, inverse
  indices, count -- few and many distinct keys, negative and 2^62-sized keys, empty input; two calls
  agree bit for bit (the hash insertion order does not matter)."""
  gen = torch.Generator().manual_seed(seed)
  keys = torch.randint(-span // 3, span, (p,), generator=gen, dtype=torch.int64)
  F = ffi()
  uniq, inv, count = F.relabel_unique(keys.to(DEV))
  want_u, want_i = torch.unique(keys, return_inverse=True)
  assert int(count) == want_u.numel()
  assert torch.equal(uniq.cpu(), want_u) and torch.equal(inv.cpu(), want_i)
  _, inv2, count2 = F.relabel_unique(keys.to(DEV), with_uniq=False)
  assert torch.equal(inv2, inv) and int(count2) == int(count)


def _sparse_tag_codes(n, gen):
  """Tag sets of two classes out of 20 (random 20-bit patterns would intersect almost surely: no negatives)."""
  return (1 << torch.randint(0, 20, (n,), generator=gen)) | (1 << torch.randint(0, 20, (n,), generator=gen))


@pytest.mark.parametrize('p,m,mode,codes', [(3000, 700, 5, 'runs'), (3000, 700, 5, 'random'), (3000, 700, 4, 'labels'),
                                            (1111, 7000, 5, 'runs'), (50000, 3100, 5, 'runs'), (129, 33, 4, 'labels'),
                                            (70000, 9000, 4, 'labels')])
def test_pipelined_embedding_gradient_kernel_matches_the_v2_kernel(p, m, mode, codes, monkeypatch):
  """csrc/nll_de3.hip (software-pipelined dE kernel, D = 64, 32-bit codes: what the semantic terms take) against
  nll_bwd_de2 on the same call: uniform and mixed prototype tiles, more than one prototype chunk, ragged pixel
  and prototype counts, pixels whose own prototype is / is not of their class, weighted upstream gradient.  Both
  kernels sum the same products in the same order up to the own prototype's term: 5e-6 of the gradient scale."""
  gen = torch.Generator().manual_seed(p + m + mode)
  d = 64
  protos = O.normalize_embedding(torch.randn(m, d, generator=gen)).to(DEV)
  own = torch.randint(0, m, (p,), generator=gen).to(DEV)
  emb = O.normalize_embedding(protos[own].cpu() + 0.8 * torch.randn(p, d, generator=gen)).to(DEV)
  if codes == 'labels':
    pr_code = torch.randint(0, 21, (m,), generator=gen)
  elif codes == 'random':
    pr_code = _sparse_tag_codes(m, gen)
  else:
    run = max(10, m // 20)
    pr_code = _sparse_tag_codes((m + run - 1) // run, gen).repeat_interleave(run)[:m]
  pr_code = pr_code.to(DEV)
  px_code = pr_code[own].clone()
  flip = (torch.rand(p, generator=gen) < 0.1).to(DEV)          # pixels whose own prototype is not of their class
  px_code[flip] = pr_code[torch.randint(0, m, (int(flip.sum()),), generator=gen).to(DEV)]
  g = (torch.rand(p, generator=gen) / p).to(DEV)
  F = ffi()
  _, stats = F.segsort_nll_fwd(emb, own, px_code, protos, pr_code, 12.0, mode)
  monkeypatch.setenv('SPML_NLL_DE3', '0')
  de2, _ = F.segsort_nll_bwd(emb, own, px_code, protos, pr_code, 12.0, mode, stats, g, m_grad=0)
  monkeypatch.setenv('SPML_NLL_DE3', '1')
  de3, _ = F.segsort_nll_bwd(emb, own, px_code, protos, pr_code, 12.0, mode, stats, g, m_grad=0)
  de3b, _ = F.segsort_nll_bwd(emb, own, px_code, protos, pr_code, 12.0, mode, stats, g, m_grad=0)
  assert torch.equal(de3, de3b)                                 # deterministic
  scale = de2.abs().max().item()
  assert scale > 0 and torch.isfinite(de3).all()
  assert (de3 - de2).abs().max().item() <= 5e-6 * scale


@pytest.mark.parametrize('p,m,mode,codes', [(3000, 700, 5, 'runs'), (3000, 700, 5, 'random'), (3000, 700, 4, 'labels'),
                                            (1111, 7000, 5, 'runs'), (50000, 3100, 5, 'runs'), (129, 33, 4, 'labels'),
                                            (70000, 9000, 4, 'labels')])
def test_pipelined_prototype_gradient_kernel_matches_the_round_2_kernel(p, m, mode, codes, monkeypatch):
  """csrc/nll_dp3.hip (software-pipelined dPr kernel, D = 64, 32-bit codes) against nll_bwd_dp on the same call:
  pixel tiles with one code (image-major pixels) and with mixed codes, ragged counts, gradient for the first third of
  the prototypes only and for all of them, pixels whose own prototype is not of their class (own-term atomics).
  Same products, other summation order (fp32 atomics in both): 2e-5 of the gradient scale."""
  gen = torch.Generator().manual_seed(p + m + mode + 1)
  d = 64
  protos = O.normalize_embedding(torch.randn(m, d, generator=gen)).to(DEV)
  own = torch.randint(0, m, (p,), generator=gen)
  if codes == 'runs':
    run = max(10, m // 20)
    own = own[torch.argsort(own // run, stable=True)]       # image-major pixels: uniform pixel tiles
    pr_code = _sparse_tag_codes((m + run - 1) // run, gen).repeat_interleave(run)[:m]
  elif codes == 'random':
    pr_code = _sparse_tag_codes(m, gen)
  else:
    pr_code = torch.randint(0, 21, (m,), generator=gen)
  own, pr_code = own.to(DEV), pr_code.to(DEV)
  emb = O.normalize_embedding(protos[own].cpu() + 0.8 * torch.randn(p, d, generator=gen)).to(DEV)
  px_code = pr_code[own].clone()
  if codes != 'runs':
    flip = (torch.rand(p, generator=gen) < 0.1).to(DEV)
    px_code[flip] = pr_code[torch.randint(0, m, (int(flip.sum()),), generator=gen).to(DEV)]
  g = (torch.rand(p, generator=gen) / p).to(DEV)
  F = ffi()
  _, stats = F.segsort_nll_fwd(emb, own, px_code, protos, pr_code, 12.0, mode)
  for m_grad in (m // 3, -1):
    monkeypatch.setenv('SPML_NLL_DP3', '0')
    de2, dp2 = F.segsort_nll_bwd(emb, own, px_code, protos, pr_code, 12.0, mode, stats, g, m_grad=m_grad)
    monkeypatch.setenv('SPML_NLL_DP3', '1')
    de3, dp3 = F.segsort_nll_bwd(emb, own, px_code, protos, pr_code, 12.0, mode, stats, g, m_grad=m_grad)
    assert torch.equal(de2, de3)
    scale = dp2.abs().max().item()
    assert scale > 0 and torch.isfinite(dp3).all()
    assert (dp3 - dp2).abs().max().item() <= 2e-5 * scale, (m_grad, (dp3 - dp2).abs().max().item() / scale)


def test_nll_at_the_eight_gpu_prototype_count_against_the_oracle():
  """M = 100 003 prototypes (what every rank sees on 8 GPUs incl. the memory bank), D = 64, tag-set predicate,
  image-major codes: forward, dEmbedding and the dPrototypes of the live third against the CPU oracle evaluated
  in chunks of pixels (the [P, M] temporaries of the reference formula: 400 MB per chunk).  The yardstick is the
  oracle in fp64 (sums over 1e5 terms: the fp32 oracle itself is only good to ~1e-5 there); per element 1e-4 of the
  gradient scale outside the pixels where the reference formula cancels (`pos = sum_same - own`, loss.py:61-70,
  with pos < own / 256), mean error 2e-6."""
  gen = torch.Generator().manual_seed(100003)
  p, m, d, kappa = 1500, 100003, 64, 12.0
  protos = O.normalize_embedding(torch.randn(m, d, generator=gen))
  own = torch.randint(0, m, (p,), generator=gen)
  emb = O.normalize_embedding(protos[own] + 0.8 * torch.randn(p, d, generator=gen))
  n_run = (m + 999) // 1000
  run_tags = torch.zeros(n_run, 20, dtype=torch.long)
  run_tags.scatter_(1, torch.stack([torch.randperm(20, generator=gen)[:2] for _ in range(n_run)]), 1)
  p_tags = run_tags.repeat_interleave(1000, dim=0)[:m]
  tags = p_tags[own]
  wgt = torch.rand(p, generator=gen) / p

  def oracle(dtype):
    pr = protos.to(dtype).requires_grad_(True)
    nll_parts, de_parts, cond = [], [], []
    for lo in range(0, p, 500):
      e = emb[lo:lo + 500].to(dtype).requires_grad_(True)
      part = O.set_segsort_nll(e, tags[lo:lo + 500], own[lo:lo + 500], pr, p_tags, kappa).view(-1)
      (part * wgt[lo:lo + 500].to(dtype)).sum().backward()
      nll_parts.append(part.detach())
      de_parts.append(e.grad)
      with torch.no_grad():
        sim = ((e @ pr.t()) * kappa).exp()
        own_s = sim.gather(1, own[lo:lo + 500].view(-1, 1)).view(-1)
        same = (tags[lo:lo + 500].to(dtype) @ p_tags.to(dtype).t()) > 0
        pos = (sim * same).sum(1) - own_s
        cond.append((pos <= 0) | (pos > own_s / 256))
    return torch.cat(nll_parts), torch.cat(de_parts), pr.grad, torch.cat(cond)

  want_nll, want_de, want_dp, well = oracle(torch.float64)
  F = ffi()
  px_code, pr_code = tags_to_mask(tags).to(DEV), tags_to_mask(p_tags).to(DEV)
  nll, stats = F.segsort_nll_fwd(emb.to(DEV), own.to(DEV), px_code, protos.to(DEV), pr_code, kappa, 1 | 4)
  rel = (nll.cpu().double() - want_nll).abs() / want_nll.abs().clamp(min=1.0)
  assert rel[well].max().item() < 2e-5 and rel.max().item() < 5e-3
  assert abs(nll.double().mean().item() - want_nll.mean().item()) <= 1e-5 * max(1.0, abs(want_nll.mean().item()))
  live = m // 3
  de, dp = F.segsort_nll_bwd(emb.to(DEV), own.to(DEV), px_code, protos.to(DEV), pr_code, kappa, 1 | 4, stats,
                             wgt.to(DEV), m_grad=live)
  scale = want_de.abs().max().item()
  err = (de.cpu().double() - want_de).abs()
  assert err[well].max().item() <= 1e-4 * scale, 'd_emb: max err %.3g vs scale %.3g' % (err[well].max().item(), scale)
  assert err.max().item() <= 1e-3 * scale and err.mean().item() <= 2e-6 * scale
  assert int((~well).sum()) <= p // 100, 'the data are not meant to be ill-conditioned'
  scale = want_dp[:live].abs().max().item()
  err = (dp.cpu().double()[:live] - want_dp[:live]).abs()
  assert err.max().item() <= 1e-4 * scale and err.mean().item() <= 2e-6 * scale, (err.max().item(), scale)
  assert torch.count_nonzero(dp[(live + 31) // 32 * 32:]).item() == 0      # (whole 32-prototype tiles are skipped)
