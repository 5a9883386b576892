"""CPU stand-ins for `spml_amd.ops`, built on the oracle.  TEST INFRASTRUCTURE ONLY.

The product has no CPU path (every op launches a HIP kernel).  The multi-process tests on
CPU (gloo) still want to drive the REAL host code -- Trainer, DistributedDataParallel,
SyncBatchNorm, the prototype exchange, the batch-id / shard-id algebra -- so they patch the
op entry points of `spml_amd.ops` with these differentiable torch-CPU equivalents inside
the test process.  Nothing under spml_amd/ imports this file."""
import torch

from oracle import spml_oracle as O

NLL_LABEL, NLL_TAGSET, NLL_PLAIN = 0, 1, 2


def normalize_rows(x):
  return O.normalize_embedding(x)


def normalize_concat_loc(emb_nchw, loc=None, row_map=None, num_rows=None):
  n, c, h, w = emb_nchw.shape
  e = O.normalize_embedding(emb_nchw.permute(0, 2, 3, 1).contiguous()).reshape(-1, c)
  if loc is None:
    loc = (O.generate_location_features((h, w), 'float') - 0.5).unsqueeze(0).expand(n, h, w, 2)
  el = O.normalize_embedding(torch.cat([e, loc.reshape(n * h * w, -1).float()], -1))
  if row_map is not None:
    keep = row_map >= 0
    e, el = e[keep], el[keep]
  return e, el


def segment_prototypes(x, ids, m):
  return O.calculate_prototypes_from_labels(x, ids, int(m))


def segsort_nll(emb, own, px_code, protos, pr_code, kappa, mode=NLL_LABEL, proto_grad_rows=None):
  if emb.shape[0] == 0:
    return emb.new_zeros((0,))
  sim = ((emb @ protos.t()) * kappa).exp()
  if mode & NLL_TAGSET:
    same = (px_code.view(-1, 1) & pr_code.view(1, -1)) != 0
  else:
    same = px_code.view(-1, 1) == pr_code.view(1, -1)
  assert not (mode & NLL_PLAIN)
  return O._nca_from_masks(sim, own, same, ~same).view(-1)


def kmeans(x, seg_offsets, max_seg_len, k, labels_init, iterations, want_centroids=False):
  off = seg_offsets.tolist()
  out = []
  with torch.no_grad():
    for b in range(len(off) - 1):
      lo, hi = off[b], off[b + 1]
      if hi > lo:
        out.append(O.kmeans_with_initial_labels(x[lo:hi].detach(), labels_init[lo:hi], int(k),
                                                int(iterations)))
  return torch.cat(out) if out else labels_init


def topk_affinity(q, protos, k, q_group=None, pr_group=None, pr_valid=None, masked_value=-2.0):
  with torch.no_grad():
    sim = q @ protos.t()
    if q_group is not None:
      ok = (q_group.view(-1, 1) == pr_group.view(1, -1)) & (pr_valid.view(1, -1) != 0)
      sim = torch.where(ok, sim, torch.full_like(sim, masked_value))
    val, idx = torch.sort(sim, dim=1, descending=True, stable=True)
    return idx[:, :k].contiguous(), val[:, :k].contiguous()


def kmeans_init_grid(h, w, ky, kx, device):
  return O.initialize_cluster_labels((ky, kx), (h, w))


def install():
  """Patch spml_amd.ops in THIS process (tests only)."""
  import spml_amd.ops as ops
  for name in ('normalize_rows', 'normalize_concat_loc', 'segment_prototypes', 'segsort_nll',
               'kmeans', 'topk_affinity', 'kmeans_init_grid'):
    setattr(ops, name, globals()[name])
