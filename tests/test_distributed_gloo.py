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


def test_single_process_is_identity():
  from spml_amd import parallel
  x = torch.randn(3, 4)
  assert parallel.all_gather_rows(x) is x
  assert not parallel.is_distributed()
