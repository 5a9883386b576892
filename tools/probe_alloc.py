"""Device allocations (hipMalloc calls of the caching allocator) and wall time per training step: which steps
after start-up still grow the memory pool (tuning aid for bench.py's warm-up)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import spml_amd
from spml_amd import synth
from spml_amd.train import Trainer, voc12_scribble_config, stress_config

recipe = sys.argv[1] if len(sys.argv) > 1 else 'voc'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
# SPML_FORCE_DISTRIBUTED=1: the collective code path on a 1-rank RCCL group (DDP, SyncBatchNorm, prototype exchange)
if os.environ.get('SPML_FORCE_DISTRIBUTED') == '1':
  import torch.distributed as dist
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29619', RANK='0', WORLD_SIZE='1')
  torch.cuda.set_device(0)
  dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
dev = torch.device('cuda', 0)
if recipe == 'stress':
  cfg, batch, crop = stress_config(batch_size=2, crop=1025), 2, 1025
else:
  cfg, batch, crop = voc12_scribble_config(batch_size=16, crop=513), 16, 513
torch.manual_seed(235)
tr = Trainer(cfg, dev, softmax_head=True, channels_last=True)
bs = [synth.make_batch(batch, crop, num_classes=cfg.dataset.num_classes, seed=235 + i, device=dev) for i in range(2)]
for d, _ in bs:
  d['image'] = d['image'].contiguous(memory_format=torch.channels_last)
prev = 0
for i in range(steps):
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  tr.step(*bs[i % 2])
  torch.cuda.synchronize()
  st = torch.cuda.memory_stats()
  n = st['num_device_alloc']
  print('step %2d  %7.1f ms  device allocations +%d frees %d retries %d  reserved %.2f GB  protos %s' % (
      i, (time.perf_counter() - t0) * 1e3, n - prev, st['num_device_free'], st['num_alloc_retries'], st['reserved_bytes.all.current'] / 2**30,
      sum(int(t.shape[0]) for t in tr.memory_banks.get('memory_prototype', []))), flush=True)
  prev = n
