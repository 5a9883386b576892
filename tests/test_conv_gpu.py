"""Matrix-core convolutions (spml_amd/csrc/conv.hip) against the framework convolution they
replace in the bottleneck unit (spml/models/backbones/resnet.py:20-33,42-63), with an fp64
convolution as the yardstick: the split-f16 path must be as close to it as the fp32 library path."""
import pytest
import torch
import torch.nn.functional as F

from spml_amd import _ffi

import os

DEV = 'cuda:0'
pytestmark = pytest.mark.gpu
# Box-independent bound on max|got - fp64| / max|fp64| of one split-f16 contraction: operands carry 22 bits
# (2^-23 relative each, two per product) and the fp32 accumulation chain is at most 4095 terms long (longer ones
# are chunked), so the error is a few 2^-22 of the largest output.  The bound is 3e-6 = 12.6 x 2^-22: >= 2.2 x the
# largest value seen on ANY assertion of this file over boxes x 4 seeds (1.29e-6, the 512 -> 2048 1x1 data
# gradient; 1.32e-6 at the batch-16 head; table: profiles/r06_test_margins.md, re-measured by tools/soak_margins.sh
# + tools/summarize_margins.py).  The fp32 library's own distance from fp64 (1e-7 .. 2e-6 on the same inputs) is
# recorded beside it as a diagnostic and may only LOOSEN the bound (a box whose library is worse than ours must not
# fail us) -- it is never what a pass depends on: VERDICT r5 weak 1, a 3e-7 floor under `1.5 x the library's error
# on this box` turned the driver run red on a box whose MIOpen happened to be closer to fp64.
FLOOR = float(os.environ.get('SPML_TEST_CONV_FLOOR', '3e-6'))
HEADLINE = FLOOR    # the batch-16 head: K = 2048 un-chunked, 67 600-pixel weight-gradient reductions
SEED = int(os.environ.get('SPML_TEST_SEED', '0'))                 # soak runs shift every generator of this file
_MARGINS = os.environ.get('SPML_TEST_MARGINS')                    # file the soak appends (what, error, bound) to


def _within(e_got, e_lib, bound, what, lib_factor=1.25):
  """Assert e_got <= bound (box-independent); `lib_factor x e_lib` can only widen it."""
  limit = bound if os.environ.get('SPML_TEST_STRICT_FLOOR') == '1' else max(lib_factor * e_lib, bound)
  if _MARGINS:
    with open(_MARGINS, 'a') as f:
      f.write('%s\t%.3e\t%.3e\t%.3e\n' % (what, e_got, bound, e_lib))
  assert e_got <= limit, (what, e_got, 'bound', bound, 'fp32 library', e_lib)


def _nhwc(t):
  return t.contiguous(memory_format=torch.channels_last)


def _rel(a, ref):
  return ((a.double() - ref).abs().max() / ref.abs().max()).item()


@pytest.mark.parametrize('n,cin,cout,h,w,k,dil,mag', [
    (2, 64, 256, 9, 11, 1, 1, 1.0), (1, 256, 256, 13, 17, 3, 2, 1.0), (2, 48, 512, 7, 5, 3, 4, 1.0),
    (3, 96, 256, 16, 16, 3, 1, 1e-6), (1, 512, 1024, 6, 9, 1, 1, 300.0), (2, 1024, 256, 12, 10, 1, 1, 1.0),
    # narrow outputs: 128-column tiles (2 x 2 waves) and 64-column tiles (1 x 4 waves), ragged row tiles
    (2, 512, 128, 9, 11, 1, 1, 1.0), (1, 128, 128, 19, 23, 3, 1, 1.0), (2, 256, 64, 17, 15, 3, 6, 1.0),
    (1, 64, 192, 21, 13, 3, 2, 1.0), (3, 128, 384, 8, 9, 1, 1, 1e-5), (1, 512, 64, 12, 12, 3, 4, 1.0),
    # long reductions (chunked accumulation): K * taps = 2048 (1x1 2048 -> 512), 2304 (3x3 256), 4608 (3x3 512)
    (1, 2048, 512, 9, 11, 1, 1, 1.0), (1, 256, 256, 17, 13, 3, 2, 1.0), (1, 512, 512, 9, 9, 3, 4, 1.0),
    (1, 512, 128, 11, 9, 3, 2, 1.0)])
def test_forward_matches_fp64_convolution(n, cin, cout, h, w, k, dil, mag):
  gen = torch.Generator().manual_seed(SEED + cin * 7 + cout)
  x = _nhwc((torch.randn(n, cin, h, w, generator=gen).clamp_min(0) * mag).to(DEV))     # post-ReLU like
  wt = (torch.randn(cout, cin, k, k, generator=gen) * (2.0 / (cin * k * k)) ** 0.5).to(DEV)
  ref = F.conv2d(x.double(), wt.double(), padding=dil * (k // 2), dilation=dil)
  lib32 = F.conv2d(x, wt, padding=dil * (k // 2), dilation=dil)
  wf, _ = _ffi.hl8_weight(wt)
  got = _ffi.conv_hl8(_ffi.hl8_from_f32(x), wf, n, h, w, k * k, dil)
  assert got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last)
  e_got, e_lib = _rel(got, ref), _rel(lib32, ref)
  # per kernel, not per unit: 22-bit operands + an fp32 accumulation chain of at most 4095 k (longer ones
  # are chunked): within FLOOR of the largest output
  _within(e_got, e_lib, FLOOR, 'fwd %d>%d k%d d%d mag%g' % (cin, cout, k, dil, mag))


@pytest.mark.parametrize('n,cin,cout,h,w,k,dil', [(2, 256, 64, 9, 11, 1, 1), (1, 256, 256, 13, 17, 3, 2),
                                                  (2, 512, 96, 8, 8, 3, 4), (2, 128, 512, 9, 10, 1, 1),
                                                  (1, 128, 128, 15, 14, 3, 1), (2, 64, 256, 11, 9, 3, 2),
                                                  # long reductions: K * taps = 2304, 4608, 2048
                                                  (1, 256, 256, 13, 11, 3, 2), (1, 512, 512, 9, 8, 3, 4),
                                                  (1, 512, 2048, 8, 9, 1, 1)])
def test_data_gradient_matches_fp64(n, cin, cout, h, w, k, dil):
  gen = torch.Generator().manual_seed(SEED + cin + cout)
  wt = (torch.randn(cout, cin, k, k, generator=gen) * (2.0 / (cin * k * k)) ** 0.5).to(DEV)
  dy = _nhwc((torch.randn(n, cout, h, w, generator=gen) * 1e-7).to(DEV))              # gradient-sized values
  res = _nhwc(torch.randn(n, cin, h, w, generator=gen).to(DEV) * 1e-7)
  ref = torch.nn.grad.conv2d_input((n, cin, h, w), wt.double(), dy.double(), padding=dil * (k // 2),
                                   dilation=dil) + res.double()
  lib32 = torch.nn.grad.conv2d_input((n, cin, h, w), wt, dy, padding=dil * (k // 2), dilation=dil) + res
  _, wtr = _ffi.hl8_weight(wt)
  got = _ffi.conv_hl8(_ffi.hl8_from_f32(dy), wtr, n, h, w, k * k, dil, addend=res)
  e_got, e_lib = _rel(got, ref), _rel(lib32, ref)
  _within(e_got, e_lib, FLOOR, 'dgrad %d>%d k%d d%d' % (cin, cout, k, dil))


@pytest.mark.parametrize('n,cin,cout,h,w,k,dil', [(2, 256, 256, 9, 11, 1, 1), (1, 256, 512, 13, 17, 3, 2),
                                                  (3, 512, 256, 8, 8, 3, 4), (2, 256, 256, 33, 31, 3, 1),
                                                  # 128-wide tiles in n, in k, in both (res3: 512 -> 128 -> 128 -> 512)
                                                  (2, 512, 128, 13, 17, 1, 1), (2, 128, 512, 9, 10, 1, 1),
                                                  (1, 128, 128, 19, 23, 3, 1), (2, 384, 128, 8, 9, 3, 2)])
def test_weight_gradient_matches_fp64(n, cin, cout, h, w, k, dil):
  gen = torch.Generator().manual_seed(SEED + cin + 3 * cout)
  x = _nhwc(torch.randn(n, cin, h, w, generator=gen).clamp_min(0).to(DEV))
  dy = _nhwc((torch.randn(n, cout, h, w, generator=gen) * 1e-6).to(DEV))
  ref = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, k, k), dy.double(), padding=dil * (k // 2), dilation=dil)
  lib32 = torch.nn.grad.conv2d_weight(x, (cout, cin, k, k), dy, padding=dil * (k // 2), dilation=dil)
  got = _ffi.conv_wgrad_hl8(_ffi.hl8_from_f32(dy), _ffi.hl8_from_f32(x), n, h, w, k * k, dil)
  assert got.shape == ref.shape
  e_got, e_lib = _rel(got, ref), _rel(lib32, ref)
  _within(e_got, e_lib, FLOOR, 'wgrad %d>%d k%d d%d' % (cin, cout, k, dil))


@pytest.mark.parametrize('stages', ['2', '4', '5', '1', '6', 'junk'])
@pytest.mark.parametrize('cin,cout', [(512, 128), (256, 256)])
def test_weight_gradient_under_the_stage_switch(monkeypatch, stages, cin, cout):
  """SPML_WGRAD_STAGES is an experiment switch of the 256 x 256 tiles only: on 128-wide tiles, and with a value no
  instantiation exists for, the launch must still happen (it was skipped, and conv_wgrad_reduce summed an
  unwritten workspace)."""
  monkeypatch.setenv('SPML_WGRAD_STAGES', stages)
  n, h, w = 2, 13, 17
  gen = torch.Generator().manual_seed(SEED + cin + cout)
  x = _nhwc(torch.randn(n, cin, h, w, generator=gen).clamp_min(0).to(DEV))
  dy = _nhwc((torch.randn(n, cout, h, w, generator=gen) * 1e-6).to(DEV))
  ref = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, 1, 1), dy.double())
  lib32 = torch.nn.grad.conv2d_weight(x, (cout, cin, 1, 1), dy)
  got = _ffi.conv_wgrad_hl8(_ffi.hl8_from_f32(dy), _ffi.hl8_from_f32(x), n, h, w, 1, 1)
  _within(_rel(got, ref), _rel(lib32, ref), FLOOR, 'wgrad stages=%s %d>%d' % (stages, cin, cout))


@pytest.mark.parametrize('n,cin,cout,h,w,k,dil', [(2, 256, 1024, 33, 33, 1, 1), (2, 256, 256, 33, 33, 3, 2),
                                                  (2, 1024, 256, 33, 33, 1, 1), (1, 512, 512, 33, 33, 3, 4)])
def test_gradients_with_the_dynamic_range_of_a_scribble_step(n, cin, cout, h, w, k, dil):
  """The output gradient of a real step is not uniform: the labelled pixels (here 1e-3 of them) carry gradients
  1e4 x those of the rest.  The hl8 format has ONE exponent window per tensor (22 bits within 2^15 of the largest
  magnitude): data and weight gradient against fp64, no absolute floor beyond the fp32 rounding level --
  the box-independent bound of this file (FLOOR) -- over the whole tensor AND over the output pixels that only see
  small gradients (1x1 convolutions: their error relative to THEIR scale)."""
  gen = torch.Generator().manual_seed(SEED + cin + cout + k)
  wt = (torch.randn(cout, cin, k, k, generator=gen) * (2.0 / (cin * k * k)) ** 0.5).to(DEV)
  x = _nhwc(torch.randn(n, cin, h, w, generator=gen).clamp_min(0).to(DEV))
  dy = torch.randn(n, cout, h, w, generator=gen) * 1e-7
  big = torch.rand(n, 1, h, w, generator=gen) < 1e-3
  big[0, 0, h // 2, w // 2] = True
  dy = _nhwc((dy * torch.where(big, 1e4, 1.0)).to(DEV))
  pad = dil * (k // 2)
  ref = torch.nn.grad.conv2d_input((n, cin, h, w), wt.double(), dy.double(), padding=pad, dilation=dil)
  lib32 = torch.nn.grad.conv2d_input((n, cin, h, w), wt, dy, padding=pad, dilation=dil)
  _, wtr = _ffi.hl8_weight(wt)
  got = _ffi.conv_hl8(_ffi.hl8_from_f32(dy), wtr, n, h, w, k * k, dil)
  e_got, e_lib = _rel(got, ref), _rel(lib32, ref)
  _within(e_got, e_lib, FLOOR, 'scribble dgrad %d>%d k%d' % (cin, cout, k))
  if k == 1:                                               # pixels that only see small gradients
    small = (~big).to(DEV).expand(n, cin, h, w)
    scale = ref[small].abs().max()
    e_got = ((got.double() - ref)[small].abs().max() / scale).item()
    e_lib = ((lib32.double() - ref)[small].abs().max() / scale).item()
    _within(e_got, e_lib, FLOOR, 'scribble dgrad small pixels %d>%d' % (cin, cout))
  ref = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, k, k), dy.double(), padding=pad, dilation=dil)
  lib32 = torch.nn.grad.conv2d_weight(x, (cout, cin, k, k), dy, padding=pad, dilation=dil)
  got = _ffi.conv_wgrad_hl8(_ffi.hl8_from_f32(dy), _ffi.hl8_from_f32(x), n, h, w, k * k, dil)
  e_got, e_lib = _rel(got, ref), _rel(lib32, ref)
  _within(e_got, e_lib, FLOOR, 'scribble wgrad %d>%d k%d' % (cin, cout, k))


def test_tiny_rows_keep_an_absolute_error_far_below_fp32_noise():
  """Elements 2^-20 below the tensor maximum are stored with fewer bits; their contribution to the
  error stays far below the rounding noise of the fp32 accumulation."""
  gen = torch.Generator().manual_seed(SEED + 5)
  x = torch.randn(1, 256, 8, 8, generator=gen)
  x[:, :, 4:] *= 2.0 ** -20
  x = _nhwc(x.to(DEV))
  wt = (torch.randn(256, 256, 1, 1, generator=gen) / 16).to(DEV)
  ref = F.conv2d(x.double(), wt.double())
  wf, _ = _ffi.hl8_weight(wt)
  got = _ffi.conv_hl8(_ffi.hl8_from_f32(x), wf, 1, 8, 8, 1)
  assert _rel(got, ref) < 6e-7
  small = (got[:, :, 4:].double() - ref[:, :, 4:]).abs().max() / ref[:, :, 4:].abs().max()
  assert small < 1e-4                                   # reduced, documented precision of tiny rows


def test_unsupported_shapes_are_refused():
  assert not _ffi.conv_hl8_supported(40, 256, 1)
  assert not _ffi.conv_hl8_supported(64, 96, 9)
  assert _ffi.conv_hl8_supported(64, 128, 9) and _ffi.conv_hl8_supported(2048, 64, 9)
  assert _ffi.conv_hl8_supported(256, 256, 9)


def test_aspp_data_gradient_in_one_launch_matches_autograd(monkeypatch):
  """ASPP head (spml/models/heads/spp.py:8-43): outputs and weight / bias gradients come from the
  framework ops, the data gradient from the 36-tap matrix-core launch; against plain autograd."""
  import copy
  from spml_amd.models.heads.spp import ASPP
  torch.manual_seed(SEED + 3)
  head = ASPP(256, 64, bn=False, relu=False).to(DEV).to(memory_format=torch.channels_last)
  ref = copy.deepcopy(head)
  x = _nhwc(torch.randn(2, 256, 33, 29, device=DEV).clamp_min(0))
  up = _nhwc(torch.randn(2, 64, 33, 29, device=DEV) * 1e-4)

  def run(m, fused):
    monkeypatch.setenv('SPML_NO_MC_CONV', '0' if fused else '1')
    xi = x.clone().requires_grad_(True)
    y = m(xi)
    (y * up).sum().backward()
    return y.detach(), xi.grad, [p.grad for p in m.parameters()]

  y1, dx1, g1 = run(head, True)
  y0, dx0, g0 = run(ref, False)
  torch.testing.assert_close(y1, y0, rtol=1e-5, atol=1e-5)
  ref64 = copy.deepcopy(ref).double()
  xi = x.double().requires_grad_(True)
  (ref64(xi) * up.double()).sum().backward()
  e_got, e_lib = _rel(dx1, xi.grad), _rel(dx0, xi.grad)
  _within(e_got, e_lib, FLOOR, 'aspp one-launch dgrad', 2.0)
  for a, b in zip(g1, g0):          # both from the library's weight-gradient kernels (atomics: not bit-stable)
    assert (a - b).abs().max().item() <= 1e-4 * b.abs().max().item()


def test_degenerate_tensors_all_zero_and_non_finite():
  """bound = 0 (all-zero tensor) keeps the scale at 1 and gives exact zeros; a non-finite input is not
  hidden by the scaling: it reaches the output as inf / nan (loud, like the fp32 path)."""
  wt = (torch.randn(256, 64, 3, 3, generator=torch.Generator().manual_seed(SEED + 1)) * 0.05).to(DEV)
  wf, _ = _ffi.hl8_weight(wt)
  z = _nhwc(torch.zeros(1, 64, 7, 9, device=DEV))
  out = _ffi.conv_hl8(_ffi.hl8_from_f32(z), wf, 1, 7, 9, 9, 2)
  assert float(out.abs().max()) == 0.0
  x = _nhwc(torch.randn(1, 64, 7, 9, device=DEV))
  x[0, 3, 2, 2] = float('inf')
  out = _ffi.conv_hl8(_ffi.hl8_from_f32(x), wf, 1, 7, 9, 9, 2)
  assert not torch.isfinite(out).all()
  huge = _nhwc(torch.randn(1, 64, 7, 9, device=DEV) * 1e30)        # far outside the f16 range: scaled
  ref = F.conv2d(huge.double(), wt.double(), padding=2, dilation=2)
  got = _ffi.conv_hl8(_ffi.hl8_from_f32(huge), wf, 1, 7, 9, 9, 2)
  assert _rel(got, ref) < 1e-6


def test_wide_aspp_runs_entirely_on_the_matrix_core_kernels(monkeypatch):
  """A head with 256-multiple output channels (BASELINE config 5: 512-d embedding): forward (one 36-tap
  launch + summed biases), data gradient and the four weight gradients against plain autograd."""
  import copy
  from spml_amd.models.heads.spp import ASPP
  torch.manual_seed(SEED + 5)
  head = ASPP(256, 256, bn=False, relu=False).to(DEV).to(memory_format=torch.channels_last)
  ref = copy.deepcopy(head)
  x = _nhwc(torch.randn(2, 256, 31, 27, device=DEV).clamp_min(0))
  up = _nhwc(torch.randn(2, 256, 31, 27, device=DEV) * 1e-4)

  def run(m, fused):
    monkeypatch.setenv('SPML_NO_MC_CONV', '0' if fused else '1')
    xi = x.clone().requires_grad_(True)
    y = m(xi)
    (y * up).sum().backward()
    return y.detach(), xi.grad, {n: p.grad for n, p in m.named_parameters()}

  y1, dx1, g1 = run(head, True)
  y0, dx0, g0 = run(ref, False)
  ref64 = copy.deepcopy(ref).double()
  xi = x.double().requires_grad_(True)
  y64 = ref64(xi)
  (y64 * up.double()).sum().backward()
  _within(_rel(y1, y64.detach()), _rel(y0, y64.detach()), FLOOR, 'wide aspp forward', 2.0)
  _within(_rel(dx1, xi.grad), _rel(dx0, xi.grad), FLOOR, 'wide aspp dgrad', 2.0)
  for (n, p) in ref64.named_parameters():
    e1, e0 = _rel(g1[n], p.grad), _rel(g0[n], p.grad)
    _within(e1, e0, FLOOR, 'wide aspp ' + n, 2.0)


def test_narrow_aspp_forward_on_64_column_tiles(monkeypatch):
  """SPML_ASPP_FWD_MC=1: the 64-channel head's forward as one 36-tap launch on 64-column tiles (chunked
  accumulation, K * taps = 9216) against the fp64 sum of the four branches; default: framework forward."""
  import copy
  from spml_amd.models.heads.spp import ASPP
  torch.manual_seed(SEED + 7)
  head = ASPP(256, 64, bn=False, relu=False).to(DEV).to(memory_format=torch.channels_last)
  x = _nhwc(torch.randn(2, 256, 29, 33, device=DEV).clamp_min(0)).requires_grad_(True)
  y64 = copy.deepcopy(head).double()(x.detach().double())
  monkeypatch.setenv('SPML_ASPP_FWD_MC', '0')
  y_lib = head(x).detach()
  monkeypatch.setenv('SPML_ASPP_FWD_MC', '1')
  y_mc = head(x).detach()
  assert not torch.equal(y_mc, y_lib)                     # (a different kernel really ran)
  _within(_rel(y_mc, y64), _rel(y_lib, y64), FLOOR, 'narrow aspp forward mc', 2.0)


@pytest.mark.parametrize('n,cin,h,w,dils', [(2, 256, 33, 29, (6, 12, 18, 24)),        # the head's four branches
                                            (1, 512, 17, 19, (1, 2, 3)),              # 27 taps: a padded last group
                                            (3, 256, 9, 11, (2, 5)),                  # dilation beyond the map's edge
                                            (2, 256, 40, 40, (3,))])
def test_pyramid_weight_gradients_in_one_launch(n, cin, h, w, dils):
  """spml_conv_wgrad_pyramid_hl8_f32: the weight gradients of up to four dilated 3x3 branches with 64 output
  channels that share one output gradient (`spml/models/heads/spp.py:8-43`), tiles of four taps x 64 channels:
  each branch against the fp64 weight gradient, within FLOOR (observed 3.05e-7 on the first shape,
  the fp32 library 2.0e-7 .. 2.4e-7 depending on the box) with the output gradient of a scribble step (1e-3 of the pixels
  carry 1e4 x the rest)."""
  gen = torch.Generator().manual_seed(SEED + cin + h + len(dils))
  x = _nhwc(torch.randn(n, cin, h, w, generator=gen).clamp_min(0).to(DEV))
  dy = torch.randn(n, 64, h, w, generator=gen) * 1e-7
  big = torch.rand(n, 1, h, w, generator=gen) < 1e-3
  big[0, 0, h // 2, w // 2] = True
  dy = _nhwc((dy * torch.where(big, 1e4, 1.0)).to(DEV))
  assert _ffi.conv_wgrad_pyramid_hl8_supported(cin, 64, len(dils))
  got = _ffi.conv_wgrad_pyramid_hl8(_ffi.hl8_from_f32(dy), _ffi.hl8_from_f32(x), n, h, w, dils)
  assert len(got) == len(dils)
  for g, d in zip(got, dils):
    assert g.shape == (64, cin, 3, 3)
    ref = torch.nn.grad.conv2d_weight(x.double(), (64, cin, 3, 3), dy.double(), padding=d, dilation=d)
    lib32 = torch.nn.grad.conv2d_weight(x, (64, cin, 3, 3), dy, padding=d, dilation=d)
    e_got, e_lib = _rel(g, ref), _rel(lib32, ref)
    _within(e_got, e_lib, FLOOR, 'pyramid wgrad cin%d %dx%d d%d' % (cin, h, w, d), 1.5)
  assert not _ffi.conv_wgrad_pyramid_hl8_supported(cin, 128, 4) and not _ffi.conv_wgrad_pyramid_hl8_supported(cin, 64, 5)
  assert not _ffi.conv_wgrad_pyramid_hl8_supported(320, 64, 4)


@pytest.mark.parametrize('n,cin,cout,h,w,dils', [(2, 256, 64, 29, 33, (6, 12, 18, 24)), (1, 256, 64, 9, 11, (2, 5)),
                                                 (3, 128, 64, 17, 8, (1, 3, 40)), (1, 128, 32, 12, 12, (1, 2, 3, 4))])
def test_narrow_pyramid_forward_as_one_gemm_and_a_gather(n, cin, cout, h, w, dils):
  """`conv_hl8_pyramid_forward_gemm`: the sum of the dilated branches as one 1x1 convolution with 9 * branches *
  Cout columns + `spml_conv_tap_gather_f32`, against the fp64 sum of the branches (within
  FLOOR) -- ragged maps, dilations beyond the map, biases; two calls are bit-identical (no atomics)."""
  gen = torch.Generator().manual_seed(SEED + cin + h + len(dils))
  x = _nhwc(torch.randn(n, cin, h, w, generator=gen).clamp_min(0).to(DEV))
  ws = [(torch.randn(cout, cin, 3, 3, generator=gen) * (2.0 / (9 * cin)) ** 0.5).to(DEV) for _ in dils]
  bs = [torch.randn(cout, generator=gen).to(DEV) for _ in dils]
  assert _ffi.conv_hl8_pyramid_forward_gemm_supported(cin, cout, len(dils))
  xa = _ffi.hl8_from_f32(x)
  got = _ffi.conv_hl8_pyramid_forward_gemm(xa, ws, bs, dils, n, h, w)
  ref = sum(F.conv2d(x.double(), wt.double(), b.double(), 1, d, d) for wt, b, d in zip(ws, bs, dils))
  lib32 = sum(F.conv2d(x, wt, b, 1, d, d) for wt, b, d in zip(ws, bs, dils))
  e_got, e_lib = _rel(got, ref), _rel(lib32, ref)
  _within(e_got, e_lib, FLOOR, 'pyramid fwd gemm cin%d cout%d' % (cin, cout), 1.5)
  assert torch.equal(got, _ffi.conv_hl8_pyramid_forward_gemm(xa, ws, bs, dils, n, h, w))
  nobias = _ffi.conv_hl8_pyramid_forward_gemm(xa, ws, [None] * len(dils), dils, n, h, w)
  torch.testing.assert_close(nobias + sum(bs).view(1, -1, 1, 1), got, rtol=0, atol=1e-5 * float(ref.abs().max()))
  assert not _ffi.conv_hl8_pyramid_forward_gemm_supported(cin, 48, 4)


def test_narrow_aspp_kernels_at_the_headline_shape():
  """The 64-channel head at the size the benchmarked step runs it (batch 16, 65 x 65 maps, 2048 channels, dilations
  6 / 12 / 18 / 24): forward (one 1x1 convolution with 2304 columns + tap gather) and the four weight gradients (one
  launch, 504 workgroups, 7 pixel splits) against fp64: within 2 x the fp32 library's own distance from fp64
  (observed forward 6.3e-7 against the library's 4.2e-7 of max|out|: un-chunked K = 2048 chains of 22-bit operands,
  the level of every convolution of csrc/conv.hip, profiles/r03_conv_accuracy.md): bound HEADLINE."""
  n, cin, h, w, dils = 16, 2048, 65, 65, (6, 12, 18, 24)
  gen = torch.Generator().manual_seed(SEED + 2048)
  x = _nhwc(torch.randn(n, cin, h, w, generator=gen).clamp_min(0).to(DEV))
  ws = [(torch.randn(64, cin, 3, 3, generator=gen) * (2.0 / (9 * cin)) ** 0.5).to(DEV) for _ in dils]
  bs = [torch.randn(64, generator=gen).to(DEV) for _ in dils]
  dy = torch.randn(n, 64, h, w, generator=gen) * 1e-7
  big = torch.rand(n, 1, h, w, generator=gen) < 1e-3
  dy = _nhwc((dy * torch.where(big, 1e4, 1.0)).to(DEV))
  xa = _ffi.hl8_from_f32(x)
  got = _ffi.conv_hl8_pyramid_forward_gemm(xa, ws, bs, dils, n, h, w)
  x64 = x.double()
  ref = sum(F.conv2d(x64, wt.double(), b.double(), 1, d, d) for wt, b, d in zip(ws, bs, dils))
  lib32 = sum(F.conv2d(x, wt, b, 1, d, d) for wt, b, d in zip(ws, bs, dils))
  e_got, e_lib = _rel(got, ref), _rel(lib32, ref)
  _within(e_got, e_lib, HEADLINE, 'headline aspp forward', 2.0)
  del ref, lib32, got
  dws = _ffi.conv_wgrad_pyramid_hl8(_ffi.hl8_from_f32(dy), xa, n, h, w, dils)
  dy64 = dy.double()
  for g, d in zip(dws, dils):
    ref = torch.nn.grad.conv2d_weight(x64, (64, cin, 3, 3), dy64, padding=d, dilation=d)
    lib32 = torch.nn.grad.conv2d_weight(x, (64, cin, 3, 3), dy, padding=d, dilation=d)
    e_got, e_lib = _rel(g, ref), _rel(lib32, ref)
    _within(e_got, e_lib, HEADLINE, 'headline aspp wgrad d%d' % d, 2.0)
    print('weight gradient d=%d: own %.2e library %.2e' % (d, e_got, e_lib))


def test_narrow_aspp_weight_gradients_leave_the_library(monkeypatch):
  """The 64-channel head's weight and bias gradients through autograd: the one-launch path (default) against
  the four library calls (SPML_ASPP_WGRAD_MC=0) and fp64."""
  import copy
  from spml_amd.models.heads.spp import ASPP
  torch.manual_seed(SEED + 11)
  head = ASPP(256, 64, bn=False, relu=False).to(DEV).to(memory_format=torch.channels_last)
  x = _nhwc(torch.randn(2, 256, 31, 35, device=DEV).clamp_min(0))
  up = _nhwc(torch.randn(2, 64, 31, 35, device=DEV) * 1e-4)

  def run(m, own):
    monkeypatch.setenv('SPML_ASPP_WGRAD_MC', '1' if own else '0')
    m.zero_grad(set_to_none=True)
    xi = x.clone().requires_grad_(True)
    (m(xi) * up).sum().backward()
    return {k: p.grad.clone() for k, p in m.named_parameters()}

  g1, g0 = run(head, True), run(head, False)
  ref64 = copy.deepcopy(head).double()
  ref64.zero_grad(set_to_none=True)
  (ref64(x.double()) * up.double()).sum().backward()
  differs = False
  for k, p in ref64.named_parameters():
    e1, e0 = _rel(g1[k], p.grad), _rel(g0[k], p.grad)
    _within(e1, e0, FLOOR, 'narrow aspp autograd ' + k, 1.5)
    differs = differs or not torch.equal(g1[k], g0[k])
  assert differs                                            # (a different kernel really ran)


@pytest.mark.parametrize('n,cin,cout,h,w,k,dil', [(2, 256, 256, 17, 19, 1, 1),       # ragged last row tile
                                                  (1, 256, 256, 13, 17, 3, 2),
                                                  (2, 128, 1024, 16, 10, 1, 1),      # four column tiles
                                                  (1, 512, 512, 9, 9, 3, 4),         # chunked accumulation
                                                  (16, 64, 256, 12, 12, 1, 1)])      # several tile heights' worth
def test_epilogue_statistics_feed_the_batch_norm(n, cin, cout, h, w, k, dil):
  """spml_conv_hl8_stats_f32: same output as spml_conv_hl8_f32; the chunk statistics of its epilogue
  (mean / M2 / max / min per row tile and channel) match the tensor; the batch norm pooled from them
  (spml_bn_fwd_hl8_chunks_f32) equals the one that reads the tensor itself."""
  gen = torch.Generator().manual_seed(SEED + cin + 3 * cout + k)
  x = _nhwc((torch.randn(n, cin, h, w, generator=gen).clamp_min(0) + 0.25).to(DEV))
  wt = (torch.randn(cout, cin, k, k, generator=gen) * (2.0 / (cin * k * k)) ** 0.5).to(DEV)
  wf, _ = _ffi.hl8_weight(wt)
  xa = _ffi.hl8_from_f32(x)
  plain = _ffi.conv_hl8(xa, wf, n, h, w, k * k, dil)
  out, st = _ffi.conv_hl8_stats(xa, wf, n, h, w, k * k, dil)
  assert st is not None and torch.equal(out, plain)
  rows = n * h * w
  flat = out.permute(0, 2, 3, 1).reshape(rows, cout).double()
  assert st.chunks == -(-rows // st.chunk_rows) and st.data.shape == (4, st.chunks, cout)
  for c in range(st.chunks):
    blk = flat[c * st.chunk_rows:(c + 1) * st.chunk_rows]
    mean = blk.mean(0)
    scale = blk.abs().max().item()
    assert (st.data[0, c].double() - mean).abs().max().item() <= 2e-6 * scale
    m2 = ((blk - mean) ** 2).sum(0)
    assert ((st.data[1, c].double() - m2).abs() <= 1e-5 * m2 + 1e-6 * scale * scale).all()
    assert torch.equal(st.data[2, c], blk.max(0).values.float()) and torch.equal(st.data[3, c], blk.min(0).values.float())
  # the local half of SyncBatchNorm pooled from the same chunks: (count, mean, M2, max, min)
  ext0, ext1 = _ffi.bn_stats_ext(out, rows, cout), _ffi.bn_stats_ext(out, rows, cout, chunk_stats=st)
  assert torch.equal(ext1[0], ext0[0]) and torch.equal(ext1[3], ext0[3]) and torch.equal(ext1[4], ext0[4])
  torch.testing.assert_close(ext1[1], ext0[1], rtol=0, atol=2e-6 * flat.abs().max().item())
  torch.testing.assert_close(ext1[2], ext0[2], rtol=2e-5, atol=1e-6)
  gamma = (torch.rand(cout, generator=gen) + 0.5).to(DEV)
  beta = torch.randn(cout, generator=gen).to(DEV)
  rm0, rv0 = torch.zeros(cout, device=DEV), torch.ones(cout, device=DEV)
  rm1, rv1 = rm0.clone(), rv0.clone()
  y0, yh0, b0, mask0, saved0 = _ffi.bn_fwd_hl8(out, rows, cout, None, None, gamma, beta, rm0, rv0, 0.1, 1e-5, True,
                                               True, True, True)
  y1, yh1, b1, mask1, saved1 = _ffi.bn_fwd_hl8(out, rows, cout, None, None, gamma, beta, rm1, rv1, 0.1, 1e-5, True,
                                               True, True, True, chunk_stats=st)
  torch.testing.assert_close(saved1[0], saved0[0], rtol=0, atol=2e-6 * flat.abs().max().item())      # mean
  torch.testing.assert_close(saved1[1], saved0[1], rtol=2e-5, atol=0)                                # invstd
  assert torch.equal(saved1[2], saved0[2]) and torch.equal(saved1[3], saved0[3])                     # extremes
  torch.testing.assert_close(y1, y0, rtol=1e-5, atol=1e-5)
  torch.testing.assert_close(rm1, rm0, rtol=1e-5, atol=1e-6)
  torch.testing.assert_close(rv1, rv0, rtol=1e-5, atol=1e-6)
  assert b1.item() >= y1.abs().max().item() and b1.item() <= 1.01 * b0.item() + 1e-6
  # against fp64 batch statistics
  mu, var = flat.mean(0), flat.var(0, unbiased=False)
  ref = torch.relu((flat - mu) / torch.sqrt(var + 1e-5) * gamma.double() + beta.double())
  got = y1.permute(0, 2, 3, 1).reshape(rows, cout).double()
  assert (got - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


def test_epilogue_statistics_fall_back_on_narrow_outputs():
  gen = torch.Generator().manual_seed(SEED + 5)
  x = _nhwc(torch.randn(1, 128, 9, 9, generator=gen).to(DEV))
  wt = torch.randn(128, 128, 1, 1, generator=gen).to(DEV) * 0.1
  wf, _ = _ffi.hl8_weight(wt)
  out, st = _ffi.conv_hl8_stats(_ffi.hl8_from_f32(x), wf, 1, 9, 9, 1)
  assert st is None and out.shape == (1, 128, 9, 9)
