#!/usr/bin/env python3
"""Run-to-run reproducibility of whole training steps: two Trainers built from the same seed take the
same N steps; reports which outputs / parameters differ.  `--deterministic` switches the library's deterministic mode
on (SPML_DETERMINISTIC=1 does the same) together with the framework's own switches."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch


def run(args, tag):
  from spml_amd import synth
  from spml_amd.train import Trainer, voc12_scribble_config
  torch.manual_seed(235)
  cfg = voc12_scribble_config(batch_size=args.batch, crop=args.crop, use_syncbn=False)
  if args.small:
    cfg.network.backbone_types = 'panoptic_deeplab_50'
  tr = Trainer(cfg, 'cuda:0', softmax_head=True, channels_last=True)
  outs = []
  for it in range(args.steps):
    datas, targets = synth.make_batch(args.batch, args.crop, seed=100 + it, device='cuda:0')
    datas['image'] = datas['image'].contiguous(memory_format=torch.channels_last)
    torch.manual_seed(1000 + it)                 # the head's dropout mask
    o = tr.step(datas, targets)
    outs.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items()})
  params = {'emb.' + k: v.detach().clone() for k, v in tr.embedding_model.named_parameters()}
  params.update({'pred.' + k: v.detach().clone() for k, v in tr.prediction_model.named_parameters()})
  return outs, params


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--deterministic', action='store_true')
  ap.add_argument('--batch', type=int, default=4)
  ap.add_argument('--crop', type=int, default=257)
  ap.add_argument('--steps', type=int, default=2)
  ap.add_argument('--small', action='store_true')
  ap.add_argument('--framework-flags', action='store_true', help='also torch.backends.cudnn.deterministic (MIOpen then runs its naive reference convolutions: 13x slower)')
  args = ap.parse_args()
  from spml_amd import _ffi
  if args.deterministic:
    _ffi.set_deterministic(True)
    os.environ['SPML_DETERMINISTIC_FRAMEWORK'] = '1' if args.framework_flags else '0'
  a_out, a_par = run(args, 'a')
  b_out, b_par = run(args, 'b')
  print('deterministic mode:', _ffi.deterministic())
  for it, (oa, ob) in enumerate(zip(a_out, b_out)):
    for k in oa:
      if torch.is_tensor(oa[k]):
        print('step %d %-14s %s  %.9g  %.9g' % (it, k, 'same' if torch.equal(oa[k], ob[k]) else 'DIFFERS', float(oa[k]), float(ob[k])))
  diff = [(k, (a_par[k] - b_par[k]).abs().max().item() / max(a_par[k].abs().max().item(), 1e-30)) for k in a_par
          if not torch.equal(a_par[k], b_par[k])]
  print('%d of %d parameter tensors differ after %d steps' % (len(diff), len(a_par), args.steps))
  for k, e in sorted(diff, key=lambda kv: -kv[1])[:12]:
    print('  %-60s rel %.2e' % (k, e))


if __name__ == '__main__':
  main()
