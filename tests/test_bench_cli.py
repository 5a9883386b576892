"""bench.py as the driver invokes it: `python bench.py --gpus N ...` must start the N ranks itself
(the reference drives all GPUs from one command, pyscripts/train/train.py:131-139,167,211), print
ONE JSON line with n_gpus = world_size = N, and refuse anything else loudly."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _run(args, env=None, timeout=900):
  e = dict(os.environ)
  for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
    e.pop(k, None)
  e.update(env or {})
  return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=e, timeout=timeout)


@pytest.mark.skipif(torch.cuda.is_available(), reason='the no-GPU refusal is checked where there is no GPU')
def test_bench_without_a_gpu_fails_cleanly():
  for args in (['--gpus', '2'], ['--gpus', '1'], []):
    r = _run(args + ['--steps', '1', '--warmup', '0'], timeout=300)
    assert r.returncode != 0
    assert 'needs an MI355X' in r.stderr and 'Traceback' not in r.stderr, r.stderr[-2000:]
    assert r.stdout.strip() == ''          # no JSON line that could be mistaken for a measurement


def test_bench_rejects_nonsense_gpu_counts():
  r = _run(['--gpus', '0'], timeout=300)
  assert r.returncode != 0 and '--gpus must be >= 1' in r.stderr


@pytest.mark.gpu
def test_bench_gpus_2_starts_two_ranks_itself():
  """Two ranks on this box's one GPU (gloo: RCCL refuses two ranks per device): the self-launch, DDP +
  SyncBatchNorm + prototype exchange, the max-over-ranks clock and the line's bookkeeping."""
  r = _run(['--gpus', '2', '--steps', '2', '--warmup', '1', '--no-kmeans', '--no-cpu-baseline', '--batch', '4',
            '--share-gpus', '--dist-backend', 'gloo'])
  assert r.returncode == 0, r.stderr[-4000:]
  lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
  assert len(lines) == 1, r.stdout[-2000:]
  res = json.loads(lines[0])
  assert res['n_gpus'] == 2 and res['world_size'] == 2 and res['steps'] == 2 and res['warmup'] == 1
  assert res['config']['global_batch'] == 8 and res['config']['parallelism'] == 'dp2'
  assert res['value'] > 0 and res['scaling'] == 'weak' and res['unit'] == 'images/s'
  lo, hi = res['ms_per_step_rank_min_max']
  assert 0 < lo <= hi and abs(hi - res['ms_per_step']) < 1e-6
  assert abs(res['value'] - 8 * 1e3 / res['ms_per_step']) < 0.01 * res['value']
  assert res['loss'] == res['loss']          # finite
  # the collective budget of a step, counted in this very run (what an 8-GPU line will carry): one all_gather per
  # synchronised batch norm forward (a unit's third and downsample batch norm share theirs), one all_reduce per batch
  # norm that has a backward (the stem and res2 are frozen), the prototype exchange and the accuracy counts
  cps = res['collectives_per_step']
  assert cps['python_level_per_step'] == sum(cps['by_call'].values())
  assert 195 <= cps['python_level_per_step'] <= 210, cps
  assert cps['by_call']['all_gather_into_tensor'] <= 106 and cps['by_call']['all_reduce'] <= 96, cps


@pytest.mark.gpu
def test_bench_refuses_more_ranks_than_gpus_and_mismatched_worlds():
  n = torch.cuda.device_count()
  r = _run(['--gpus', str(n + 1), '--steps', '1', '--warmup', '0'], timeout=300)
  assert r.returncode != 0 and 'this node exposes' in r.stderr and r.stdout.strip() == ''
  r = _run(['--gpus', '1', '--steps', '1', '--warmup', '0'],
           env=dict(WORLD_SIZE='2', RANK='0', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='29999'),
           timeout=300)
  assert r.returncode != 0 and 'WORLD_SIZE=2' in r.stderr and r.stdout.strip() == ''


def _load_bench():
  import importlib.util
  spec = importlib.util.spec_from_file_location('spml_bench', BENCH)
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


def test_roofline_reads_the_committed_rocprof_average(tmp_path, monkeypatch):
  """`roofline.frac` is capped by the AverageNs of the pass kernel in the committed rocprofv3 stats of the driver's
  command: the parser finds the kernel by name, and says so when the file or the row is missing."""
  bench = _load_bench()
  stats = tmp_path / 'stats.csv'
  stats.write_text('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n'
                   '"void spml::(anonymous namespace)::kmeans_pass64<3, 8, 1, true>(spml::PassArgs)",3249,172489410,53090.0,59.9,48441,62601,3043.1\n'
                   '"void spml::(anonymous namespace)::kmeans_pass64<3, 8, 1, false>(spml::PassArgs)",47,2143336,45602.8,5.7,44080,48000,945.1\n')
  monkeypatch.setattr(bench, 'ROCPROF_STATS_FILE', str(stats))
  us, calls = bench.rocprof_mean_us('kmeans_pass64<3, 8, 1, true>')
  assert abs(us - 53.09) < 1e-9 and calls == 3249
  # the figure a reader recomputes: algorithmic bytes / that duration / 8 TB/s
  assert abs(273770064 / (us * 1e-6) / 1e9 / bench.HBM_PEAK_GBPS - 0.6446) < 1e-3
  assert bench.rocprof_mean_us('kmeans_pass16<3, 8, 1, true>') == (None, 0)
  monkeypatch.setattr(bench, 'ROCPROF_STATS_FILE', str(tmp_path / 'missing.csv'))
  assert bench.rocprof_mean_us('kmeans_pass64<3, 8, 1, true>') == (None, 0)


def test_collective_counter_counts_and_restores():
  """spml_amd.parallel.count_collectives patches torch.distributed for the duration of a step and puts it back."""
  import torch.distributed as dist
  from spml_amd import parallel
  before = dist.all_reduce
  with parallel.count_collectives() as cc:
    assert dist.all_reduce is not before
  assert dist.all_reduce is before and cc.total == 0 and cc.gpu_ms() == {} if torch.cuda.is_available() else cc.total == 0
