#!/usr/bin/env python3
"""Stage 2: training the softmax classifier on a frozen embedding network, with the reference's command
line and config surface (`pyscripts/train/train_classifier.py:33-185` of twke18/SPML):

  python3 pyscripts/train/train_classifier.py --data_dir D --data_list L --snapshot_dir S --cfg_path config_classifier.yaml
  torchrun --nproc-per-node 8 pyscripts/train/train_classifier.py ...      (one process per GPU, RCCL)

`config.network.pretrained` must name stage 1's snapshot (`model-{iter}.pth`; its `embedding_model` entry
is loaded, :97-104 -- "Pre-trained model is required."); `prediction_types` must be `softmax_classifier`
(:84-87), `backbone_types` `panoptic_deeplab_101` / `panoptic_pspnet_101` (:78-82).  The step itself is
`spml_amd.train.ClassifierTrainer.step`; snapshots are the reference's two files (:172-180).  As in the
stage-1 entry point the file-list loader (ListTagClassifierDataset) is outside this repository:
`--data_list synthetic` feeds seeded synthetic batches of the loader's shape."""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main(argv=None):
  from spml_amd.config.default import config
  from spml_amd.config.parse_args import parse_args
  args = parse_args('Training for softmax classifier only.', argv)
  if not torch.cuda.is_available():
    raise SystemExit('training needs an MI355X (the HIP path has no CPU fallback)')
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local = int(os.environ.get('LOCAL_RANK', '0'))
  torch.cuda.set_device(local)
  device = torch.device('cuda', local)
  if world > 1:
    dist.init_process_group('nccl', device_id=device)
  from spml_amd import synth
  from spml_amd.train import ClassifierTrainer
  os.makedirs(args.snapshot_dir, exist_ok=True)
  model_path = os.path.join(args.snapshot_dir, 'model-{:d}.pth')
  state_path = os.path.join(args.snapshot_dir, 'model-{:d}.state.pth')
  torch.manual_seed(235)                    # train_classifier.py:26-27
  trainer = ClassifierTrainer(config, device, channels_last=True)
  if not config.network.pretrained:
    raise ValueError('Pre-trained model is required.')
  print('Loading pre-trained model: {:s}'.format(config.network.pretrained))
  trainer.load_pretrained(config.network.pretrained)
  if args.data_list not in (None, 'synthetic'):
    raise SystemExit('file-list data loading (ListTagClassifierDataset) is outside the scope of this repository; '
                     'use --data_list synthetic or plug a loader that yields (datas, targets) dicts')
  t0 = time.time()
  for curr_iter in range(trainer.curr_iter, config.train.max_iteration):
    datas, targets = synth.make_batch(config.train.batch_size, config.train.crop_size[0],
                                      num_classes=config.dataset.num_classes, seed=235 + 1009 * rank + curr_iter,
                                      device=device, supervision=args.supervision, palette=(1, 3))
    datas['image'] = datas['image'].contiguous(memory_format=torch.channels_last)
    out = trainer.step(datas, targets)
    if rank == 0 and (curr_iter % 10 == 0 or curr_iter == config.train.max_iteration - 1):
      print('iter {:d}: loss = {:.3f}, acc = {:.3f}, lr = {:.6f}  ({:.2f} s)'.format(
          curr_iter, float(out['loss']), float(out['accuracy']), out['lr'], time.time() - t0), flush=True)
    if rank == 0 and config.train.snapshot_step and (
        (curr_iter + 1) % config.train.snapshot_step == 0 or curr_iter == config.train.max_iteration - 1):
      state = trainer.state_dict()
      torch.save({'embedding_model': state['embedding_model'],
                  'prediction_model': state['prediction_model']}, model_path.format(curr_iter))
      torch.save(state['optimizer'], state_path.format(curr_iter))
  if world > 1:
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
