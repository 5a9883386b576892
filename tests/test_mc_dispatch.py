"""Host-side dispatch of the matrix-core bottleneck units (no GPU): which units qualify, and that
everything else stays on the framework ops."""
import torch

from spml_amd import _ffi, mc_bottleneck
from spml_amd.models.backbones.resnet import Bottleneck, ResnetBackbone
from spml_amd.models.heads.spp import ASPP, _dilated_sum_available


def test_supported_shapes_follow_the_kernel_tiling():
  assert _ffi.conv_hl8_supported(1024, 256, 1) and _ffi.conv_hl8_supported(256, 256, 9)
  assert _ffi.conv_hl8_supported(1024, 128, 1) and _ffi.conv_hl8_supported(2048, 64, 9)   # narrow tiles
  assert not _ffi.conv_hl8_supported(1024, 96, 1)           # 64-column granularity
  assert not _ffi.conv_hl8_supported(1000, 256, 1)          # 16-channel k steps
  assert not _ffi.conv_hl8_supported(256, 256, 25)
  assert _ffi.conv_wgrad_hl8_supported(1024, 256, 1) and _ffi.conv_wgrad_hl8_supported(128, 256, 9)
  assert not _ffi.conv_wgrad_hl8_supported(64, 256, 9)       # 128-channel granularity


def test_cpu_nchw_eval_and_strided_units_stay_on_the_framework():
  blk = Bottleneck(1024, 256, 1, dilation=2).train()
  x = torch.randn(1, 1024, 5, 5)
  assert not mc_bottleneck.available(blk, x)                # CPU tensor
  y = blk(x)                                                # ... and the framework path still works
  assert y.shape == x.shape
  meta = torch.empty(1, 1024, 5, 5, device='meta')
  assert not mc_bottleneck.available(blk, meta)


def test_only_res4_and_res5_units_qualify_by_shape():
  """Channel counts of the DeepLab-v2 backbone: res4 / res5 units tile (256-multiples), the stride-1 units
  of res3 run the 128-wide tiles, res2 does not tile, the stride-2 unit never does."""
  net = ResnetBackbone([3, 4, 23, 3], [1, 2, 1, 1], [1, 1, 2, 4])

  def shape_ok(b):
    convs = [b.conv1, b.conv2, b.conv3] + ([b.downsample[0]] if b.downsample is not None else [])
    return b.stride == 1 and all(
        _ffi.conv_hl8_supported(c.in_channels, c.out_channels, c.kernel_size[0] * c.kernel_size[1]) and
        _ffi.conv_hl8_supported(c.out_channels, c.in_channels, c.kernel_size[0] * c.kernel_size[1]) and
        _ffi.conv_wgrad_hl8_supported(c.in_channels, c.out_channels, c.kernel_size[0] * c.kernel_size[1])
        for c in convs)
  assert not any(shape_ok(b) for b in list(net.res2))                    # 64-channel convolutions
  assert [shape_ok(b) for b in net.res3] == [False, True, True, True]   # 128-wide tiles; unit 0 has stride 2
  assert all(shape_ok(b) for b in list(net.res4) + list(net.res5))


def test_aspp_fused_data_gradient_needs_gpu_channels_last_and_bare_branches():
  head = ASPP(256, 64, bn=False, relu=False)
  x = torch.randn(1, 256, 9, 9, requires_grad=True)
  assert not _dilated_sum_available(x, [b[0] for b in (head.aspp_1, head.aspp_2, head.aspp_3, head.aspp_4)])
  out = head(x)                                             # framework path on CPU
  out.sum().backward()
  assert x.grad is not None and out.shape == (1, 64, 9, 9)
  with_bn = ASPP(256, 64, bn=True, relu=True)
  assert with_bn(torch.randn(2, 256, 9, 9)).shape == (2, 64, 9, 9)
