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


def headline_depth_accuracy(batch=4, crop=513, seed=235, verbose=None):
  """Accuracy of ONE forward + backward of the benchmarked configuration at full depth: ResNet-101 DeepLab-v2,
  `crop` x `crop`, VOC12 scribble recipe, random-init weights, the GPU's own clustering of this batch held fixed.

    A  the benchmarked path: channels-last, matrix-core units (csrc/conv.hip) + fused batch norm + HIP loss kernels
    B  the same step on NCHW / the fp32 library convolutions (same GPU, same clustering)
    C  the embedding network in fp64 on the GPU (framework kernels), forward + backward from A's d loss / d embedding
    D  the CPU oracle's loss head in fp64, evaluated AT A's embedding map (same clustering)

  -> dict: 'stages' {name: (rel L2 of A vs C, of B vs C)} for conv1 / res2..res5 / aspp / embedding;
  'losses' {name: (A, D)}; 'd_embedding' rel L2 of A vs D; 'param_grad' list of (name, A vs C, B vs C) with the network
  gradients of all three taken from A's upstream gradient; 'mc_units', 'agreement'."""
  import copy
  import torch.nn.functional as F
  from oracle.cpu_step import CpuStep
  from spml_amd import mc_bottleneck, synth
  from spml_amd.train import Trainer, build_models, voc12_scribble_config
  say = verbose or (lambda *a: None)
  dev = 'cuda:0'
  cfg = voc12_scribble_config(batch_size=batch, crop=crop, use_syncbn=False)
  torch.manual_seed(seed)
  emb, pred = build_models(cfg, softmax_head=True)
  pred.semantic_classifier[3].p = 0.0                      # (dropout of the softmax head: off, as in the h01 goldens)
  datas, targets = synth.make_batch(batch, crop, num_classes=cfg.dataset.num_classes, seed=seed, palette=(1, 3))
  stages = ('conv1', 'res2', 'res3', 'res4', 'res5')

  def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp(min=1e-300)).item()

  def hook_stages(model, store):
    hs = []
    for name in stages:
      hs.append(getattr(model.resnet_backbone, name).register_forward_hook(
          lambda m, i, o, n=name: store.__setitem__(n, o.detach().to(torch.float64).cpu())))
    hs.append(model.aspp.register_forward_hook(lambda m, i, o: store.__setitem__('aspp', o.detach().to(torch.float64).cpu())))
    return hs

  def run_trainer(channels_last, ids):
    tr = Trainer(cfg, dev, softmax_head=True, channels_last=channels_last,
                 models=(copy.deepcopy(emb), copy.deepcopy(pred)))
    tr.embedding_model.train(); tr.prediction_model.train()
    acts, rec, seen, calls = {}, {}, {}, {}
    hs = hook_stages(tr.embedding_model, acts)
    real = segsort_common.segment_by_kmeans

    def spy(embeddings, *a, **k):
      seen['emb'] = embeddings
      out = real(embeddings, *a, **k)
      seen['ids'] = out[3].detach().clone()
      return out
    d, t = to_gpu(datas, targets, channels_last)
    if ids is None:                                        # first run: this GPU path's own clustering
      segsort_common.segment_by_kmeans = spy
      try:
        with torch.no_grad():
          tr.forward_losses(d, t)
      finally:
        segsort_common.segment_by_kmeans = real
      ids = seen['ids']
    segsort_common.segment_by_kmeans = spy
    try:
      with given_clustering([ids], rec), count_calls(mc_bottleneck, 'bottleneck_forward', calls):
        loss, out, _ = tr.forward_losses(d, t)
        loss.backward()
    finally:
      segsort_common.segment_by_kmeans = real
    for h in hs:
      h.remove()
    acts['embedding'] = seen['emb'].detach().to(torch.float64).cpu()
    grads = {n: p.grad.detach().to(torch.float64).cpu() for n, p in tr.embedding_model.named_parameters() if p.grad is not None}
    return dict(ids=ids, acts=acts, d_emb=rec['d_embedding'][0].detach(), loss=loss.item(),
                out={k: float(v) for k, v in out.items() if v is not None and torch.is_tensor(v) and v.numel() == 1},
                grads=grads, mc=calls.get('bottleneck_forward', 0), agreement=rec['agreement'], tr=tr)

  a = run_trainer(True, None)
  say('A done: loss %.6f, matrix-core units entered %d times' % (a['loss'], a['mc']))
  b = run_trainer(False, a['ids'])
  say('B done: loss %.6f' % b['loss'])
  # C: the embedding network in fp64 on the GPU, forward + backward from A's upstream gradient
  e64 = copy.deepcopy(emb).double().to(dev).train()
  for name in ('conv1', 'res2'):
    for p in getattr(e64.resnet_backbone, name).parameters():
      p.requires_grad_(False)
  acts_c = {}
  hs = hook_stages(e64, acts_c)
  prev = torch.backends.cudnn.enabled
  torch.backends.cudnn.enabled = False                     # (MIOpen has no fp64 convolutions: framework kernels)
  try:
    out_c = e64.generate_embeddings({'image': datas['image'].double().to(dev)})
    emb_c = out_c['embedding']
    emb_c.backward(a['d_emb'].double())
  finally:
    torch.backends.cudnn.enabled = prev
  for h in hs:
    h.remove()
  acts_c['embedding'] = emb_c.detach().cpu()
  grads_c = {n: p.grad.detach().cpu() for n, p in e64.named_parameters() if p.grad is not None}
  say('C done (fp64 network on the GPU)')

  def net_grads_from(tr_path, channels_last):
    """parameter gradients of a GPU fp32 path from A's upstream gradient (so that A, B, C differentiate the same function)"""
    m = copy.deepcopy(emb).to(dev).train()
    for name in ('conv1', 'res2'):
      for p in getattr(m.resnet_backbone, name).parameters():
        p.requires_grad_(False)
    img = datas['image'].to(dev)
    if channels_last:
      m = m.to(memory_format=torch.channels_last)
      img = img.contiguous(memory_format=torch.channels_last)
    o = m.generate_embeddings({'image': img})['embedding']
    o.backward(a['d_emb'].to(o.dtype))
    return {n: p.grad.detach().to(torch.float64).cpu() for n, p in m.named_parameters() if p.grad is not None}
  ga, gb = net_grads_from(a, True), net_grads_from(b, False)
  # D: the oracle's loss head in fp64 at A's embedding
  given = a['acts']['embedding'].clone().requires_grad_(True)
  lfn64 = copy.deepcopy(emb.lfn).double()

  class AtEmbedding:
    def generate_embeddings(self, d, *args, **kw):
      return {'embedding': given, 'local_feature': lfn64(d['image'], size=given.shape[-2:])}
  cpu = CpuStep(AtEmbedding(), copy.deepcopy(pred).double(), cfg, None, softmax_head=True)
  cpu.given_cluster_index = a['ids'].cpu()
  l64, out64, _ = cpu.forward_losses({'image': datas['image'].double()}, targets)
  l64.backward()
  say('D done: fp64 loss at the GPU embedding %.8f' % l64.item())
  res = {
      'stages': {n: (rel(a['acts'][n], acts_c[n]), rel(b['acts'][n], acts_c[n])) for n in stages + ('aspp', 'embedding')},
      'losses': dict({k: (a['out'][k], float(out64[k])) for k in ('sem_ann_loss', 'sem_occ_loss', 'img_sim_loss')},
                     total=(a['loss'], l64.item())),
      'loss_b': b['loss'],
      'd_embedding': rel(a['d_emb'], given.grad),
      'd_embedding_b': rel(b['d_emb'], given.grad),
      'param_grad': sorted(((n, rel(ga[n], grads_c[n]), rel(gb[n], grads_c[n])) for n in grads_c), key=lambda t: -t[1]),
      'mc_units': a['mc'], 'agreement': a['agreement'],
  }
  return res
