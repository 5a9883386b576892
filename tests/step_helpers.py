"""Helpers of the training-step parity tests (tests/test_train_step_gpu.py).

`given_clustering`: k-means is chaotic -- one pixel whose top-2 margin is below the rounding
difference between two convolution implementations lands in another segment and moves a loss
by ~5e-4.  The chaotic part is pinned on its own (every M- / E-step of every kernel family
against the oracle, tests/test_kernels_gpu.py::check_kmeans_stepwise; goldens a06 / a08);
here the reference's (or the oracle's) own segment ids are injected into the GPU step, so that
everything downstream -- prototypes, exchange, the contrastive losses, the softmax head, the
backward through the network, SGD -- is a smooth function of the weights and can be compared
at north_star's 1e-4.  The GPU k-means still RUNS on the GPU embeddings (its result is kept for
an agreement statistic); only the ids handed on are replaced.

`count_calls`: asserts that a code path (matrix-core units, fused cross-entropy) was entered."""
import contextlib

import torch

import spml_amd.utils.segsort.common as segsort_common


@contextlib.contextmanager
def given_clustering(ids_per_call, record):
  """ids_per_call: list of 1-D int tensors, one per segment_by_kmeans call, in call order.
  record: dict filled with 'agreement' (fraction of pixel PAIRS (i, i+1) on which the GPU's own
  clustering and the injected one agree about same-segment / different-segment) and
  'd_embedding' (list of gradients w.r.t. the NCHW embedding map, one per call)."""
  real = segsort_common.segment_by_kmeans
  queue = list(ids_per_call)
  record.setdefault('agreement', [])
  record.setdefault('d_embedding', [])

  def patched(embeddings, *args, **kw):
    out = list(real(embeddings, *args, **kw))
    ids = queue.pop(0).to(out[3].device).long()
    assert ids.shape == out[3].shape, (ids.shape, out[3].shape)
    own = out[3]
    if own.numel() > 1:
      record['agreement'].append(((own[1:] == own[:-1]) == (ids[1:] == ids[:-1])).float().mean().item())
    if embeddings.requires_grad:
      embeddings.register_hook(lambda g: record['d_embedding'].append(g.detach().clone()))
    out[3] = ids
    return tuple(out)

  segsort_common.segment_by_kmeans = patched
  try:
    yield record
  finally:
    segsort_common.segment_by_kmeans = real


@contextlib.contextmanager
def count_calls(module, name, counter):
  real = getattr(module, name)

  def counting(*a, **kw):
    counter[name] = counter.get(name, 0) + 1
    return real(*a, **kw)

  setattr(module, name, counting)
  try:
    yield counter
  finally:
    setattr(module, name, real)


def to_gpu(datas, targets, channels_last):
  d = {k: v.cuda() for k, v in datas.items()}
  if channels_last:
    d['image'] = d['image'].contiguous(memory_format=torch.channels_last)
  return d, {k: v.cuda() for k, v in targets.items()}
