"""Data-parallel exchange of segment prototypes: one process per GPU,
`torch.distributed` (backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).

The reference funnels every GPU's PIXELS to one anchor GPU, recomputes all
prototypes there and broadcasts them (models/utils.py:86-127; SURVEY 2.1 C3/C4,
~43 MB per GPU).  A segment never spans images and an image never spans ranks,
so here every rank computes the prototypes of its own images and only the
prototypes move: a variable-length all-gather of [M_r, C] / [M_r, C+2] rows and
three label vectors (hundreds of KB).  Backward: the gradient of every rank's
loss w.r.t. ALL prototypes is summed across ranks and every rank receives only
the slice of the prototypes it owns (reduce-scatter over equal, zero-padded
slices: half the traffic of an all-reduce)."""
import os

import torch
import torch.distributed as dist


def is_distributed():
  """True in a multi-rank job.  SPML_FORCE_DISTRIBUTED=1 also takes the collective code
  path in a 1-rank process group (a single-GPU box can then exercise DDP, SyncBatchNorm
  and the prototype exchange over RCCL end to end)."""
  if not (dist.is_available() and dist.is_initialized()):
    return False
  return dist.get_world_size() > 1 or os.environ.get('SPML_FORCE_DISTRIBUTED') == '1'


def _all_sizes(n, device):
  """Every rank's row count: one collective, one host read."""
  world = dist.get_world_size()
  mine = torch.tensor([n], dtype=torch.long, device=device)
  sizes = torch.zeros((world,), dtype=torch.long, device=device)
  dist.all_gather_into_tensor(sizes, mine)
  return [int(v) for v in sizes.tolist()]


def _through_host(x):
  """gloo has no device all_gather / reduce_scatter: with that backend (ranks sharing one GPU in the
  tests, `bench.py --dist-backend gloo`) those two collectives are staged through host memory."""
  return x.is_cuda and dist.get_backend() == 'gloo'


def _all_gather_rows(x, sizes):
  """Concatenate every rank's [m_r, ...] rows (rank-major)."""
  world = dist.get_world_size()
  mx = max(sizes)
  pad = x.new_zeros((mx,) + tuple(x.shape[1:]))
  pad[:x.shape[0]] = x
  if _through_host(x):
    host = pad.cpu()
    bufs = [torch.empty_like(host) for _ in range(world)]
    dist.all_gather(bufs, host)
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], 0).to(x.device)
  bufs = [torch.empty_like(pad) for _ in range(world)]
  dist.all_gather(bufs, pad.contiguous())
  return torch.cat([b[:s] for b, s in zip(bufs, sizes)], 0)


class _AllGatherRowsGrad(torch.autograd.Function):
  """all-gather whose backward returns d(sum of all ranks' losses)/d(local rows)."""

  @staticmethod
  def forward(ctx, x, sizes):
    ctx.sizes = sizes
    ctx.rank = dist.get_rank()
    return _all_gather_rows(x.detach(), sizes)

  @staticmethod
  def backward(ctx, grad):
    sizes, world = ctx.sizes, len(ctx.sizes)
    mx = max(max(sizes), 1)
    # rank q's rows -> slot q of a [world, mx, ...] buffer (zero padded), reduce-scatter
    padded = grad.new_zeros((world, mx) + tuple(grad.shape[1:]))
    lo = 0
    for q, n in enumerate(sizes):
      padded[q, :n] = grad[lo:lo + n]
      lo += n
    if _through_host(grad):
      full = padded.cpu()
      dist.all_reduce(full, op=dist.ReduceOp.SUM)
      return full[ctx.rank, :sizes[ctx.rank]].to(grad.device), None
    mine = grad.new_empty((mx,) + tuple(grad.shape[1:]))
    dist.reduce_scatter_tensor(mine, padded.view((world * mx,) + tuple(grad.shape[1:])),
                               op=dist.ReduceOp.SUM)
    return mine[:sizes[ctx.rank]], None


def all_gather_rows(x, sizes=None, differentiable=False):
  """Variable-length all-gather along dim 0; identity when not distributed."""
  if not is_distributed():
    return x
  if sizes is None:
    sizes = _all_sizes(x.shape[0], x.device)
  if differentiable and x.requires_grad:
    return _AllGatherRowsGrad.apply(x, sizes)
  return _all_gather_rows(x, sizes)


def gather_prototypes(protos, protos_loc, proto_sem, proto_ins, proto_bat, cluster_indices):
  """Rank-local prototypes -> global prototype set on every rank.

  Rank-major concatenation reproduces the ordering of the reference's global
  `unique(batch * div + cluster)` (models/utils.py:95-97) because rank r owns
  batch ids r*N .. r*N+N-1.  Local segment ids are shifted by the number of
  prototypes owned by lower ranks."""
  if not is_distributed():
    return protos, protos_loc, proto_sem, proto_ins, proto_bat, cluster_indices
  sizes = _all_sizes(protos.shape[0], protos.device)
  rank = dist.get_rank()
  g_protos = all_gather_rows(protos, sizes, differentiable=True)
  g_protos_loc = all_gather_rows(protos_loc, sizes, differentiable=True)
  g_sem = all_gather_rows(proto_sem, sizes)
  g_ins = all_gather_rows(proto_ins, sizes)
  g_bat = all_gather_rows(proto_bat, sizes)
  return g_protos, g_protos_loc, g_sem, g_ins, g_bat, cluster_indices + sum(sizes[:rank])


def gather_tags(semantic_tag):
  """All ranks' image tag vectors, rank-major (train.py:194-202)."""
  if not is_distributed():
    return semantic_tag
  return all_gather_rows(semantic_tag, [semantic_tag.shape[0]] * dist.get_world_size())


def shard_bounds(n, rank, world):
  """Rows [lo, hi) of an n-row problem that rank `rank` of `world` handles (contiguous,
  sizes differ by at most one)."""
  return (n * rank) // world, (n * (rank + 1)) // world


def sharded_retrieval_accuracy(top_k_ranking, prototypes, prototype_labels, top_k):
  """Self-retrieval accuracy of the global prototype set (`top_k_ranking(P, l, P, l, k)`,
  segsort.py:213-218).  Every rank holds the same prototypes, and the metric is quadratic
  in their number (which grows with the ranks through the gathered prototypes and the
  memory bank): each rank ranks only its share of the queries against all prototypes and
  the integer hit counts are summed across ranks -- the same value, 1/world of the work."""
  m = prototypes.shape[0]
  if not is_distributed():
    acc, _ = top_k_ranking(prototypes, prototype_labels, prototypes, prototype_labels, top_k)
    return acc
  lo, hi = shard_bounds(m, dist.get_rank(), dist.get_world_size())
  hits = torch.zeros((), dtype=torch.long, device=prototypes.device)
  if hi > lo:
    _, labels = top_k_ranking(prototypes[lo:hi], prototype_labels[lo:hi], prototypes,
                              prototype_labels, top_k)
    hits = (labels == prototype_labels[lo:hi].reshape(-1, 1)).sum()
  dist.all_reduce(hits, op=dist.ReduceOp.SUM)
  return hits.float() / float(m * top_k)


class count_collectives(object):
  """Context manager: counts the torch.distributed collectives issued from Python while it is active
  (SyncBatchNorm statistics, prototype exchange, accuracy counts -- the calls that sit on the critical path of a
  training step; DistributedDataParallel's bucketed gradient all-reduces run inside the C++ reducer, overlap with
  the backward pass and are not counted: `ddp_buckets` estimates them).  `bench.py` reports the numbers so that a
  multi-GPU line explains its own collective budget; the 1-rank RCCL test pins them."""
  NAMES = ('all_gather_into_tensor', 'all_gather', 'all_reduce', 'reduce_scatter_tensor', 'broadcast', 'all_to_all_single')

  def __init__(self, timed=False):
    """timed: also bracket every call with stream events -- `gpu_ms()` then gives the GPU time the current stream
    spent inside those collectives (transfer + waiting for the slowest peer), per call name."""
    self.calls = {}
    self._saved = {}
    self._timed = timed
    self._events = []

  def __enter__(self):
    import torch.distributed as dist         # (the real module, whatever this file's `dist` has been replaced with)
    self._mod = dist
    for name in self.NAMES:
      fn = getattr(dist, name, None)
      if fn is None:
        continue
      self._saved[name] = fn

      def wrapper(*a, __fn=fn, __name=name, **kw):
        self.calls[__name] = self.calls.get(__name, 0) + 1
        if self._timed and torch.cuda.is_available():
          e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
          e0.record()
          res = __fn(*a, **kw)
          e1.record()
          self._events.append((__name, e0, e1))
          return res
        return __fn(*a, **kw)
      setattr(dist, name, wrapper)
    return self

  def __exit__(self, *exc):
    for name, fn in self._saved.items():
      setattr(self._mod, name, fn)
    return False

  @property
  def total(self):
    return sum(self.calls.values())

  def gpu_ms(self):
    """{call name: ms on the current stream between the events around its calls} (synchronises)."""
    torch.cuda.synchronize()
    out = {}
    for name, e0, e1 in self._events:
      out[name] = out.get(name, 0.0) + e0.elapsed_time(e1)
    return out


def small_collective_latency_us(device, channels=2048, reps=100):
  """Mean duration of one [world, 3, C] fp32 all-gather -- the SyncBatchNorm statistics exchange, ~208 of which sit
  on the critical path of a ResNet-101 step -- over `reps` back-to-back calls (HIP events)."""
  import torch.distributed as rdist        # (the real group: tools/emulate_world.py replaces this module's `dist`)
  world = rdist.get_world_size()
  mine = torch.zeros((3, channels), device=device)
  out = torch.zeros((world * 3, channels), device=device)
  for _ in range(5):
    rdist.all_gather_into_tensor(out, mine)
  if torch.device(device).type != 'cuda':    # (gloo tests on CPU: the wall clock of the blocking calls)
    import time
    t0 = time.perf_counter()
    for _ in range(reps):
      rdist.all_gather_into_tensor(out, mine)
    return (time.perf_counter() - t0) * 1e6 / reps
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    rdist.all_gather_into_tensor(out, mine)
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / reps
