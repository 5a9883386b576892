"""world_size-2 test of the prototype exchange on CPU (gloo): the sharded
all-gather reproduces the reference's single-process global computation
(golden b01_gather) in ordering and value, and its backward sums every rank's
gradient into the owning rank."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def _worker(rank, world, port, out):
  sys.path.insert(0, ROOT)
  sys.path.insert(0, os.path.join(ROOT, 'tests'))
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    from conftest import load_golden
    from oracle import spml_oracle as O
    from spml_amd import parallel
    g = load_golden('b01_gather')
    pre = 's%d_' % rank
    emb = g[pre + 'emb'].clone().requires_grad_(True)
    embloc = g[pre + 'embloc'].clone().requires_grad_(True)
    # rank-local prototypes (the product computes these with the HIP kernels)
    r = O.gather_clustering_and_update_prototypes(
        [emb], [embloc], [g[pre + 'clu']], [g[pre + 'bat']], [g[pre + 'sem']], [g[pre + 'ins']])
    local = [x[0] for x in r]
    protos, protos_loc, p_sem, p_ins, p_bat, clu = parallel.gather_prototypes(*local)
    torch.testing.assert_close(protos.detach(), g.protos, rtol=0, atol=1e-6)
    torch.testing.assert_close(protos_loc.detach(), g.protos_loc, rtol=0, atol=1e-6)
    assert torch.equal(p_sem, g.p_sem) and torch.equal(p_ins, g.p_ins) and torch.equal(p_bat, g.p_bat)
    assert torch.equal(clu, g[pre + 'new_clu'])
    # every rank evaluates the same scalar on ALL prototypes; the backward all-reduce sums
    # the ranks' gradients, i.e. world x the single-process gradient of the golden file
    ((protos * g.wgt).sum() + (protos_loc * g.wgt2).sum()).backward()
    torch.testing.assert_close(emb.grad / world, g[pre + 'd_emb'], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(embloc.grad / world, g[pre + 'd_embloc'], rtol=1e-5, atol=1e-6)
    tags = parallel.gather_tags(torch.full((2, 4), rank, dtype=torch.long))
    assert tags.tolist() == [[0] * 4, [0] * 4, [1] * 4, [1] * 4]
    sizes = parallel._all_sizes(3 + rank, torch.device('cpu'))
    assert sizes == [3, 4]
    # the self-retrieval accuracy, sharded over the ranks' queries, equals the global value
    a = load_golden('a11_topk')
    want, _ = O.top_k_ranking(a.pr, a.prl, a.pr, a.prl, 5)
    got = parallel.sharded_retrieval_accuracy(O.top_k_ranking, a.pr, a.prl, 5)
    assert abs(float(got) - float(want)) < 1e-6 and abs(float(want) - float(a.acc_self)) < 1e-6
    assert [parallel.shard_bounds(7, r, 3) for r in range(3)] == [(0, 2), (2, 4), (4, 7)]
    # the collective counter of bench.py (`collectives_per_step`): wraps torch.distributed while active, restores it
    before = dist.all_reduce
    with parallel.count_collectives() as cc:
      v = torch.ones(3)
      dist.all_reduce(v)
      buf = torch.empty(2 * 3)
      dist.all_gather_into_tensor(buf, v)
      parallel.gather_tags(torch.zeros((1, 4), dtype=torch.long))
    assert dist.all_reduce is before
    assert cc.calls.get('all_reduce') == 1 and cc.calls.get('all_gather_into_tensor', 0) >= 1 and cc.total >= 3
    assert v.tolist() == [2.0, 2.0, 2.0] and buf.tolist() == [2.0] * 6
    lat = parallel.small_collective_latency_us(torch.device('cpu'), channels=8, reps=5)
    assert lat > 0
    out.put((rank, 'ok'))
  except Exception as e:                                    # pragma: no cover
    import traceback
    out.put((rank, traceback.format_exc()))
  finally:
    dist.destroy_process_group()


def test_prototype_all_gather_world2():
  ctx = mp.get_context('spawn')
  out = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
  for p in procs:
    p.start()
  res = [out.get(timeout=180) for _ in procs]
  for p in procs:
    p.join(timeout=60)
  for rank, msg in res:
    assert msg == 'ok', 'rank %d: %s' % (rank, msg)


def _step_worker(rank, world, port, out, recipe):
  """Two steps of the REAL Trainer (DDP + prototype exchange + memory bank)
  on CPU over gloo; the HIP ops are replaced by tests/cpu_ops.py in this process only."""
  sys.path.insert(0, ROOT)
  sys.path.insert(0, os.path.join(ROOT, 'tests'))
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  torch.set_num_threads(2)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    import cpu_ops
    cpu_ops.install()
    from spml_amd import synth
    from spml_amd.train import Trainer, densepose_point_config, voc12_scribble_config
    from tools_synth import reinit_parameters
    n = 2
    if recipe == 'densepose':
      from spml_amd.models.embeddings.resnet_pspnet_densepose import ResnetPspnet
      from spml_amd.models.predictions.segsort_softmax_densepose import segsort
      cfg = densepose_point_config(batch_size=n, crop=97, embedding_dim=16, kmeans=3)
      cfg.train.memory_bank_size = 1
      emb = ResnetPspnet([1, 1, 1, 1], [1, 2, 1, 1], [1, 1, 2, 4], cfg)
    else:
      from spml_amd.models.embeddings.resnet_deeplab import ResnetDeeplab
      from spml_amd.models.predictions.segsort_softmax import segsort
      if recipe == 'tag':                  # BASELINE config 3: image-tag recipe, data-parallel with the prototype all-gather
        from spml_amd.train import voc12_tag_config
        cfg = voc12_tag_config(batch_size=n, crop=97, embedding_dim=16, kmeans=3)
      else:
        cfg = voc12_scribble_config(batch_size=n, crop=97, embedding_dim=16, kmeans=3)
      emb = ResnetDeeplab([1, 1, 1, 1], [1, 2, 1, 1], [1, 1, 2, 4], cfg)
    cfg.network.kmeans_iterations = 3
    cfg.network.use_syncbn = False       # torch's SyncBatchNorm is GPU-only (its RCCL path is
    cfg.gpus = '0,1'                     # covered by the 1-rank test in test_train_step_gpu.py)
    pred = segsort(cfg)
    reinit_parameters(emb, 11)
    reinit_parameters(pred, 12)
    pred.semantic_classifier[3].p = 0.0
    tr = Trainer(cfg, 'cpu', softmax_head=True, models=(emb, pred), recipe='densepose' if recipe == 'densepose' else 'voc')
    assert tr.distributed and tr.world == world
    seen = {}
    orig = tr.pred_fwd.forward if hasattr(tr.pred_fwd, 'forward') else None

    def spy(datas, targets, *a, **kw):
      seen['cluster_batch'] = datas['cluster_batch_index'].clone()
      seen['proto_batch'] = targets['prototype_batch_index'].clone()
      seen['n_tags'] = targets['semantic_tag'].shape[0]
      mem = targets.get('memory_prototype_batch_index', None)
      seen['mem_batch'] = [m.clone() for m in mem] if mem else []
      return orig(datas, targets, *a, **kw)
    tr.pred_fwd.forward = spy
    losses = []
    for it in range(2):
      datas, targets = synth.make_batch(n, 97, num_classes=cfg.dataset.num_classes,
                                        seed=(51 if recipe == 'tag' else 50) + 7 * rank + it,      # (tag blobs at this crop: seed 50 labels no pixel at all)
                                        supervision='tag' if recipe == 'tag' else 'scribble')
      o = tr.step(datas, targets)
      assert torch.isfinite(o['loss'])
      losses.append(float(o['loss']))
      # rank r owns global image ids r*n .. r*n+n-1, whatever the device ordinal
      cb = seen['cluster_batch']
      assert cb.min().item() >= rank * n and cb.max().item() < (rank + 1) * n
      # ... and every rank sees the prototypes and tags of ALL images
      assert set(seen['proto_batch'].tolist()) == set(range(world * n))
      assert seen['n_tags'] == world * n
      if it == 1:       # memory bank of step 0, shifted past every live image id
        for m in seen['mem_batch']:
          assert m.min().item() >= world * n
    # replicas stay bit-identical: same parameters on every rank after 2 DDP steps
    flat = torch.cat([p.detach().reshape(-1) for p in tr.embedding_model.parameters()
                      if p.requires_grad])
    ref = flat.clone()
    dist.broadcast(ref, 0)
    assert torch.equal(flat, ref), 'replicas diverged'
    out.put((rank, 'ok'))
  except Exception:                                         # pragma: no cover
    import traceback
    out.put((rank, traceback.format_exc()))
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize('recipe', ['voc', 'tag', 'densepose'])
def test_full_training_step_world2(recipe):
  """DDP + prototype exchange (all-gather fwd / reduce-scatter bwd) + memory bank + the
  rank-based batch ids in a 2-rank gloo job on CPU, 2 steps, for the VOC and the DensePose recipe (the latter
  has a parameter used only under no_grad: DDP must not wait for its gradient)."""
  ctx = mp.get_context('spawn')
  out = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_step_worker, args=(r, 2, port, out, recipe)) for r in range(2)]
  for p in procs:
    p.start()
  res = [out.get(timeout=600) for _ in procs]
  for p in procs:
    p.join(timeout=60)
  for rank, msg in res:
    assert msg == 'ok', 'rank %d: %s' % (rank, msg)


def _stage2_worker(rank, world, port, out):
  """Two steps of ClassifierTrainer (stage 2) under DDP over gloo: every rank its own image, the classifier's
  gradients averaged, the embedding network frozen."""
  sys.path.insert(0, ROOT)
  sys.path.insert(0, os.path.join(ROOT, 'tests'))
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  torch.set_num_threads(2)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    from conftest import load_golden
    from spml_amd.train import ClassifierTrainer
    from tools_synth import h02_batch, h02_config, h02_models, parameter_checksums
    g = load_golden('h02_classifier_step')
    cfg = h02_config()
    emb, pred = h02_models(cfg)
    tr = ClassifierTrainer(cfg, 'cpu', models=(emb, pred))
    assert tr.distributed and tr.pred_fwd is not tr.prediction_model
    before = parameter_checksums(emb)[1]
    # the gradient DDP leaves on every rank = the mean of the ranks' own gradients: checked against a second,
    # unwrapped copy of the classifier stepped by hand on this rank's image
    import copy
    solo = copy.deepcopy(pred)
    for it in range(2):
      datas, targets = h02_batch(g, it)
      mine = ({'image': datas['image'][rank:rank + 1]}, {'semantic_label': targets['semantic_label'][rank:rank + 1]})
      if it == 0:
        with torch.no_grad():
          e = emb.eval().generate_embeddings(mine[0])['embedding']
        solo.train()
        solo({'embedding': e}, mine[1])['sem_ann_loss'].backward()
        own = torch.cat([p.grad.reshape(-1) for p in solo.parameters()])
        mean = own.clone()
        dist.all_reduce(mean)
        mean /= world
      o = tr.step(*mine)
      assert torch.isfinite(o['loss'])
      if it == 0:
        got = torch.cat([p.grad.reshape(-1) for p in tr.prediction_model.parameters()])
        torch.testing.assert_close(got, mean, rtol=1e-5, atol=1e-7)
    flat = torch.cat([p.detach().reshape(-1) for p in tr.prediction_model.parameters()])
    ref = flat.clone()
    dist.broadcast(ref, 0)
    assert torch.equal(flat, ref), 'replicas diverged'
    assert torch.equal(parameter_checksums(emb)[1], before)
    out.put((rank, 'ok'))
  except Exception:                                         # pragma: no cover
    import traceback
    out.put((rank, traceback.format_exc()))
  finally:
    dist.destroy_process_group()


def test_stage2_classifier_step_world2():
  ctx = mp.get_context('spawn')
  out = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_stage2_worker, args=(r, 2, port, out)) for r in range(2)]
  for p in procs:
    p.start()
  res = [out.get(timeout=600) for _ in procs]
  for p in procs:
    p.join(timeout=60)
  for rank, msg in res:
    assert msg == 'ok', 'rank %d: %s' % (rank, msg)


def test_single_process_is_identity():
  from spml_amd import parallel
  x = torch.randn(3, 4)
  assert parallel.all_gather_rows(x) is x
  assert not parallel.is_distributed()
