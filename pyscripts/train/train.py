#!/usr/bin/env python3
"""Training entry point with the reference's command line and config surface
(`pyscripts/train/train.py:41-152` of twke18/SPML):

  python3 pyscripts/train/train.py --data_dir D --data_list L --snapshot_dir S --cfg_path C.yaml
  torchrun --nproc-per-node 8 pyscripts/train/train.py ...        (one process per GPU, RCCL)

Same YAML keys (spml_amd/config/default.py mirrors spml/config/default.py), same model /
predictor selection (`backbone_types`, `prediction_types`), same optimizer groups, lr policy,
memory bank and step order (spml_amd/train.py::Trainer.step), same snapshot files
(`model-{iter}.pth` with `embedding_model` / `prediction_model`, `model-{iter}.state.pth`).
Differences: one process per GPU under torch.distributed instead of nn.DataParallel over
`config.gpus` (the global batch is `train.batch_size` per process x world size, as the
reference's per-GPU batch), no tensorboard, and the data source: the reference's
ListTagDataset (cv2 augmentation, file lists) is outside the scope of this repository --
`--data_list synthetic` (the default when none is given) feeds seeded synthetic batches of
the loader's shape (spml_amd/synth.py); a real loader only has to yield the same
`(datas, targets)` dicts."""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def synthetic_batches(config, device, rank, supervision, first=0):
  """Seeded per (rank, iteration): a resumed run continues the sequence where it stopped."""
  from spml_amd import synth
  it = first
  while True:
    yield synth.make_batch(config.train.batch_size, config.train.crop_size[0],
                           num_classes=config.dataset.num_classes, seed=235 + 1009 * rank + it,
                           device=device, supervision=supervision, palette=(1, 3))
    it += 1


def main(argv=None, default_recipe='voc', description='Training for pixel-wise embeddings.'):
  """`default_recipe`: which model / predictor modules the entry point binds -- 'voc' as
  pyscripts/train/train.py:27-32 (DeepLab / PSPNet embedding + SegsortSoftmax), 'densepose' as
  pyscripts/train/train_densepose.py:27-29 (colour + location local features, nearest-neighbour
  propagated tags).  `--recipe` / `--supervision` override it explicitly; nothing is guessed from
  file names."""
  from spml_amd.config.default import config
  from spml_amd.config.parse_args import parse_args
  args = parse_args(description, argv)
  if not torch.cuda.is_available():
    raise SystemExit('training needs an MI355X (the HIP path has no CPU fallback)')
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local = int(os.environ.get('LOCAL_RANK', '0'))
  torch.cuda.set_device(local)
  device = torch.device('cuda', local)
  if world > 1:
    dist.init_process_group('nccl', device_id=device)
  from spml_amd.train import Trainer
  if args.kmeans_num_clusters:
    config.network.kmeans_num_clusters = [int(v) for v in args.kmeans_num_clusters.split(',')]
  if args.label_divisor:
    config.network.label_divisor = args.label_divisor
  config.gpus = ','.join(str(i) for i in range(world))
  os.makedirs(args.snapshot_dir, exist_ok=True)
  model_path = os.path.join(args.snapshot_dir, 'model-{:d}.pth')
  state_path = os.path.join(args.snapshot_dir, 'model-{:d}.state.pth')

  if config.network.prediction_types not in ('segsort',):
    raise ValueError('Not support ' + str(config.network.prediction_types))
  recipe = args.recipe or default_recipe
  torch.manual_seed(235)                    # train.py:34-35
  trainer = Trainer(config, device, softmax_head=True, recipe=recipe, channels_last=True)
  if config.train.resume:
    it0 = config.train.begin_iteration
    state = torch.load(model_path.format(it0), map_location=device, weights_only=True)
    # (plain containers + tensors: loads with the safe unpickler; generator states stay on the host)
    extra = torch.load(state_path.format(it0), map_location="cpu", weights_only=True)
    # model-{iter}.state.pth = the optimizer state dict (the reference's file, train.py:303) + what a
    # faithful resume also needs: memory bank, iteration counter, generator states
    state['optimizer'] = {k: extra[k] for k in ('state', 'param_groups')}
    state['memory_banks'] = extra.get('spml_memory_banks', {})
    state['iteration'] = extra.get('spml_iteration', it0 + 1)
    trainer.load_state_dict(state)
    if 'spml_rng' in extra:
      torch.set_rng_state(extra['spml_rng']['cpu'])
      torch.cuda.set_rng_state(extra['spml_rng']['cuda'], device)
    print('Resume training from {:s}'.format(model_path.format(it0)))
  elif config.network.pretrained:
    print('Loading pre-trained model: {:s}'.format(config.network.pretrained))
    trainer.embedding_model.load_state_dict(torch.load(config.network.pretrained, map_location=device, weights_only=True))
  else:
    print('Training from scratch')

  if args.data_list in (None, 'synthetic'):
    batches = synthetic_batches(config, device, rank, args.supervision, first=trainer.curr_iter)
  else:
    raise SystemExit('file-list data loading (ListTagDataset) is outside the scope of this repository; '
                     'use --data_list synthetic or plug a loader that yields (datas, targets) dicts')

  t0 = time.time()
  for curr_iter in range(trainer.curr_iter, config.train.max_iteration):
    datas, targets = next(batches)
    out = trainer.step(datas, targets)
    if rank == 0 and (curr_iter % 10 == 0 or curr_iter == config.train.max_iteration - 1):
      print('iter {:d}: loss = {:.3f}, acc = {:.3f}, lr = {:.6f}  ({:.2f} s)'.format(
          curr_iter, float(out['loss']), float(out.get('accuracy', 0.0)), out['lr'], time.time() - t0),
          flush=True)
    if rank == 0 and config.train.snapshot_step and (
        (curr_iter + 1) % config.train.snapshot_step == 0 or curr_iter == config.train.max_iteration - 1):
      state = trainer.state_dict()
      torch.save({'embedding_model': state['embedding_model'],
                  'prediction_model': state['prediction_model']}, model_path.format(curr_iter))
      extra = dict(state['optimizer'])
      extra['spml_memory_banks'] = state['memory_banks']
      extra['spml_iteration'] = state['iteration']
      extra['spml_rng'] = {'cpu': torch.get_rng_state(), 'cuda': torch.cuda.get_rng_state(device)}
      torch.save(extra, state_path.format(curr_iter))
  if world > 1:
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
