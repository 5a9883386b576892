#!/usr/bin/env python3
"""One-off MIOpen solver search for the conv shapes of the training step.

The ROCm image ships no gfx950 find-db, so immediate mode (cudnn.benchmark=False)
falls back to im2col + GEMM / layout-transposing solvers.  This script runs the
training step with cudnn.benchmark=True (miopenFindConvolution*Algorithm) while
MIOPEN_USER_DB_PATH points into the repo, so every search result lands in a user
find-db that later runs (bench.py, Trainer) reuse without searching.

  python tools/miopen_tune.py --db spml_amd/miopen_db --steps 4 [--recipe voc --batch 16 --crop 513]

Progress lines (one per conv module on the first step) go to --log so that a run
cut off by a timeout still shows how far the search got; results are stored by
MIOpen as each search completes, so a second run resumes where the first stopped."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--db', default=os.path.join(ROOT, 'spml_amd', 'miopen_db'))
  ap.add_argument('--cache', default=None, help='MIOPEN_CUSTOM_CACHE_DIR (compiled kernels)')
  ap.add_argument('--find-mode', default='NORMAL')
  ap.add_argument('--steps', type=int, default=4)
  ap.add_argument('--batch', type=int, default=16)
  ap.add_argument('--crop', type=int, default=513)
  ap.add_argument('--recipe', default='voc')
  ap.add_argument('--log', default=None)
  ap.add_argument('--no-benchmark', action='store_true', help='immediate mode (check a stored db)')
  ap.add_argument('--channels-last', action='store_true', help='NHWC activations / weights')
  args = ap.parse_args()
  os.makedirs(args.db, exist_ok=True)
  os.environ['MIOPEN_USER_DB_PATH'] = os.path.abspath(args.db)
  if args.cache:
    os.makedirs(args.cache, exist_ok=True)
    os.environ['MIOPEN_CUSTOM_CACHE_DIR'] = os.path.abspath(args.cache)
  if args.find_mode:
    os.environ['MIOPEN_FIND_MODE'] = args.find_mode

  import torch
  from spml_amd import synth
  from spml_amd.train import (Trainer, densepose_point_config, stress_config, voc12_scribble_config,
                              voc12_tag_config)
  torch.backends.cudnn.benchmark = not args.no_benchmark
  device = torch.device('cuda', 0)
  torch.cuda.set_device(0)
  cfg = {'voc': voc12_scribble_config, 'tag': voc12_tag_config, 'stress': stress_config,
         'densepose': densepose_point_config}[args.recipe](batch_size=args.batch, crop=args.crop)
  torch.manual_seed(235)
  trainer = Trainer(cfg, device, softmax_head=True, channels_last=args.channels_last,
                    recipe='densepose' if args.recipe == 'densepose' else 'voc')
  log = open(args.log, 'a') if args.log else sys.stderr
  t_start = time.time()
  state = {'first': True}

  def hook(name):
    def f(mod, inp, out):
      if state['first']:
        torch.cuda.synchronize()
        print('%7.1f s  fwd %s %s -> %s' % (time.time() - t_start, name, tuple(inp[0].shape),
                                            tuple(out.shape)), file=log, flush=True)
    return f
  for name, m in trainer.embedding_model.named_modules():
    if isinstance(m, torch.nn.Conv2d):
      m.register_forward_hook(hook(name))
  batches = [synth.make_batch(args.batch, args.crop, num_classes=cfg.dataset.num_classes,
                              seed=235 + i, device=device,
                              supervision='tag' if args.recipe == 'tag' else 'scribble')
             for i in range(2)]
  if args.channels_last:
    for d, _ in batches:
      d['image'] = d['image'].contiguous(memory_format=torch.channels_last)
  for i in range(args.steps):
    t0 = time.time()
    out = trainer.step(*batches[i % 2])
    torch.cuda.synchronize()
    state['first'] = False
    print('%7.1f s  step %d: %.1f ms  loss %.4f' % (time.time() - t_start, i, (time.time() - t0) * 1e3,
                                                   float(out['loss'])), file=log, flush=True)
  n = 8
  torch.cuda.synchronize()
  t0 = time.time()
  for i in range(n):
    trainer.step(*batches[i % 2])
  torch.cuda.synchronize()
  print('steady state: %.1f ms/step' % ((time.time() - t0) / n * 1e3), file=log, flush=True)
  print('steady state: %.1f ms/step' % ((time.time() - t0) / n * 1e3))


if __name__ == '__main__':
  main()
