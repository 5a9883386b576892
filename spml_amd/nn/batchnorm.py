"""Batch normalisation entry points of the host mirror.

`BatchNorm2d` keeps the reference's module type / parameter names (`nn.BatchNorm2d` inside
`nn.Sequential`s: spml/models/heads/spp.py:21-33,60-75, predictions/segsort_softmax.py:33-45) but
never runs PyTorch-ROCm's NCHW training-mode batch norm.  Measured on MI355X (torch 2.10 + ROCm 7):
that kernel's batch statistics are off by a few elements' worth -- 2.5e-4 relative error of the output
for a [2, 64, 81, 81] fp32 activation, 8e-6 for [16, 64, 257, 257], independent of the
`cudnn_enabled` switch -- while the channels-last kernel is exact to 6e-8 like the CPU's
(`profiles/r03_step_accuracy.md`).  Through a 4-stage network this compounds to 2-3e-3 on the
embedding and 10 % on the gradients of the first trainable layer, 1000 x the CPU fp32 path's error.
So: channels-last fp32 GPU activations in training mode go through this repository's fused kernels
(`ops.batch_norm_act`); NCHW GPU activations are normalised in channels-last form (two layout
copies, NCHW is not the benchmarked layout) and handed back in NCHW."""
import torch
import torch.nn as nn


def _needs_sync(bn):
  if not (isinstance(bn, nn.SyncBatchNorm) and bn.training):
    return False
  import torch.distributed as dist
  if not (dist.is_available() and dist.is_initialized()):
    return False
  group = bn.process_group if bn.process_group is not None else dist.group.WORLD
  return dist.get_world_size(group) > 1


def native_batch_norm(bn, x):
  """`bn(x)` for an nn.BatchNorm2d / SyncBatchNorm module without the library's batch-norm kernels.
  A SyncBatchNorm that really has peers keeps its own forward (ATen statistics + collectives); with
  one rank it would fall back to the library kernel, so it takes the native path here too."""
  if _needs_sync(bn):
    if x.is_cuda and x.dim() == 4 and not x.is_contiguous(memory_format=torch.channels_last):
      return nn.SyncBatchNorm.forward(bn, x.contiguous(memory_format=torch.channels_last)).contiguous()
    return nn.SyncBatchNorm.forward(bn, x)
  if not x.is_cuda:
    return nn.modules.batchnorm._BatchNorm.forward(bn, x)
  training = bn.training or (bn.running_mean is None and bn.running_var is None)
  factor = 0.0 if bn.momentum is None else bn.momentum
  if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
    bn.num_batches_tracked.add_(1)
    if bn.momentum is None:
      factor = 1.0 / float(bn.num_batches_tracked)
  nchw = x.dim() == 4 and training and not x.is_contiguous(memory_format=torch.channels_last)
  if nchw:
    x = x.contiguous(memory_format=torch.channels_last)
  y = torch.batch_norm(x, bn.weight, bn.bias,
                       bn.running_mean if (not bn.training or bn.track_running_stats) else None,
                       bn.running_var if (not bn.training or bn.track_running_stats) else None,
                       training, factor, bn.eps, False)
  return y.contiguous() if nchw else y


class BatchNorm2d(nn.BatchNorm2d):
  """nn.BatchNorm2d (same parameters, buffers and state-dict keys) on the fused / native kernels."""

  def forward(self, x):
    from spml_amd import ops
    return ops.batch_norm_act(x, self, relu=False)
