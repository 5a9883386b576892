"""Deterministic-mode convolution for the few units that stay on the framework (MIOpen) convolutions.

MIOpen's fp32 NHWC weight-gradient kernels split the reduction over workgroups and meet through atomics: run to run
the bits of dW differ.  The framework's own switch (`torch.backends.cudnn.deterministic`) answers with MIOpen's naive
reference convolutions -- 1.5 s per training step instead of 0.12 (profiles/r06_determinism.md).  In the library's
deterministic mode (`spml_set_deterministic`, include/spml_hip.h) `spml_amd.train.Trainer` therefore re-classes the
`nn.Conv2d` modules that (a) have a trainable weight and (b) are not taken over by the matrix-core units
(`mc_bottleneck`) to `DetConv2d`: same parameters, buffers and state-dict keys; forward and data gradient on the
library as before (no cross-workgroup accumulation there), the weight gradient as ONE matrix product per convolution
(`dW = dY [Cout, N L] x unfold(X) [N L, Cin k k]`, the reduction cut into chunks of one batched GEMM and added in a fixed order), the bias gradient as a plain
reduction.  Reference modules concerned: the stride-2 unit of `spml/models/backbones/resnet.py:42-63` and the classifier
heads (`segsort_softmax.py:33-48`, `softmax_classifier.py:14-30`)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _fixed_order_matmul_t(a, b, chunk=2048):
  """a [M, K] x b [N, K]^T with a summation order that does not depend on the run: the reduction is cut into chunks
  of `chunk` that go through ONE batched GEMM (enough independent problems to fill the chip, so the BLAS library has no
  reason to split K over workgroups and meet through atomics -- which it does for a single [128, 270 k] x [270 k, 576]
  product: the classifier head's weight gradients differed run to run by 1e-7), and the partial products are added
  by a plain reduction."""
  m, k = a.shape
  n = b.shape[0]
  nb = (k + chunk - 1) // chunk
  if nb <= 1:
    return torch.matmul(a, b.t())
  pad = nb * chunk - k
  if pad:
    a = F.pad(a, (0, pad))
    b = F.pad(b, (0, pad))
  a3 = a.reshape(m, nb, chunk).permute(1, 0, 2)                       # [nb, M, chunk]
  b3 = b.reshape(n, nb, chunk).permute(1, 2, 0)                       # [nb, chunk, N]
  return torch.bmm(a3, b3).sum(0)


class _DetConv2dFn(torch.autograd.Function):

  @staticmethod
  def forward(ctx, x, w, b, stride, padding, dilation):
    ctx.conf = (stride, padding, dilation)
    ctx.save_for_backward(x, w)
    ctx.has_bias = b is not None
    ctx.pointwise = (tuple(w.shape[2:]) == (1, 1) and tuple(stride) == (1, 1) and tuple(padding) == (0, 0))
    if ctx.pointwise:
      # a 1x1 convolution as the matrix product it is (rows = pixels): the library's solver for narrow outputs
      # (128 -> 21 classes) gave run-to-run different logits
      n, cin, h, wd = x.shape
      rows = x.permute(0, 2, 3, 1).reshape(-1, cin)
      y = torch.matmul(rows, w.view(w.shape[0], cin).t())
      if b is not None:
        y = y + b
      y = y.view(n, h, wd, -1).permute(0, 3, 1, 2)   # (a channels-last view of the [rows, Cout] product)
      # the layout of the input is kept: an NCHW network stays NCHW (a channels-last activation would switch the units
      # behind it to the matrix-core path)
      nchw = x.is_contiguous() and not x.is_contiguous(memory_format=torch.channels_last)
      return y.contiguous() if nchw else y
    return F.conv2d(x, w, b, stride, padding, dilation)

  @staticmethod
  def backward(ctx, g):
    x, w = ctx.saved_tensors
    stride, padding, dilation = ctx.conf
    dx = dw = db = None
    if ctx.needs_input_grad[0] and ctx.pointwise:
      n, cout, h, wd = g.shape
      rows = g.permute(0, 2, 3, 1).reshape(-1, cout)
      dx = torch.matmul(rows, w.view(cout, -1)).view(n, h, wd, -1).permute(0, 3, 1, 2)
      if x.is_contiguous() and not x.is_contiguous(memory_format=torch.channels_last):
        dx = dx.contiguous()
    elif ctx.needs_input_grad[0]:
      dx = torch.ops.aten.convolution_backward(g, x, w, None, list(stride), list(padding), list(dilation), False,
                                               [0, 0], 1, [True, False, False])[0]
    if ctx.needs_input_grad[1]:
      cout, cin, kh, kw = w.shape
      n = x.shape[0]
      if kh == 1 and kw == 1 and tuple(padding) == (0, 0):
        xs = x[:, :, ::stride[0], ::stride[1]]
        a = g.permute(1, 0, 2, 3).reshape(cout, -1)                    # [Cout, N L]
        bmat = xs.permute(1, 0, 2, 3).reshape(cin, -1)                 # [Cin, N L]
        dw = _fixed_order_matmul_t(a, bmat).view(cout, cin, 1, 1)
      else:
        xu = F.unfold(x.contiguous(), (kh, kw), dilation=dilation, padding=padding, stride=stride)   # [N, Cin k k, L]
        a = g.reshape(n, cout, -1).permute(1, 0, 2).reshape(cout, -1)  # [Cout, N L]
        bmat = xu.permute(1, 0, 2).reshape(cin * kh * kw, -1)          # [Cin k k, N L]
        dw = _fixed_order_matmul_t(a, bmat).view(cout, cin, kh, kw)
      dw = dw.contiguous(memory_format=torch.channels_last) if w.is_contiguous(memory_format=torch.channels_last) \
          and not w.is_contiguous() else dw
    if ctx.has_bias and ctx.needs_input_grad[2]:
      db = g.sum(dim=(0, 2, 3))
    return dx, dw, db, None, None, None


class DetConv2d(nn.Conv2d):
  """nn.Conv2d whose weight gradient has a fixed summation order (see the module docstring)."""

  def forward(self, x):
    if x.is_cuda and self.groups == 1 and self.padding_mode == 'zeros' and not isinstance(self.padding, str):
      pointwise = self.kernel_size == (1, 1) and self.stride == (1, 1) and self.padding == (0, 0)
      # 1x1 products always (also under no_grad: the library's forward is not run-to-run stable for every shape);
      # the others when a gradient will be asked for
      if pointwise or (torch.is_grad_enabled() and (self.weight.requires_grad or x.requires_grad)):
        return _DetConv2dFn.apply(x, self.weight, self.bias, tuple(self.stride), tuple(self.padding),
                                  tuple(self.dilation))
    return super().forward(x)


def make_deterministic(module):
  """Re-class every plain nn.Conv2d with a trainable weight under `module` to DetConv2d (in place); returns the count."""
  count = 0
  for m in module.modules():
    if type(m) is nn.Conv2d and m.weight.requires_grad:
      m.__class__ = DetConv2d
      count += 1
  return count
