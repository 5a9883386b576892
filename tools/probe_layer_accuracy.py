#!/usr/bin/env python3
"""Forward error of every module of the embedding network on the GPU against an fp64 CPU copy
(relative L2 of the module outputs, in execution order) -- finds the op that loses accuracy.
  python tools/probe_layer_accuracy.py [--nhwc] [--config h01|small|densepose]"""
import copy, os, sys
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from spml_amd import synth
from tools_synth import h01_config, h01_models, reinit_parameters


def build(which):
  if which == 'h01':
    cfg = h01_config()
    emb, _ = h01_models(cfg)
    return emb, 161, 21
  if which == 'small':
    from spml_amd.train import voc12_scribble_config, build_models
    cfg = voc12_scribble_config(batch_size=2, crop=97, embedding_dim=32, kmeans=4, use_syncbn=False)
    cfg.network.backbone_types = 'panoptic_deeplab_50'
    torch.manual_seed(0)
    emb, _ = build_models(cfg, False)
    return emb, 97, 21
  from spml_amd.train import densepose_point_config
  from spml_amd.models.embeddings.resnet_pspnet_densepose import ResnetPspnetDensepose
  cfg = densepose_point_config(batch_size=2, crop=129, embedding_dim=32, kmeans=4, use_syncbn=False)
  return reinit_parameters(ResnetPspnetDensepose([1, 1, 1, 1], [1, 2, 1, 1], [1, 1, 2, 4], cfg), 41), 129, 15


def main():
  nhwc = '--nhwc' in sys.argv
  which = sys.argv[sys.argv.index('--config') + 1] if '--config' in sys.argv else 'h01'
  emb, crop, classes = build(which)
  emb.train()
  e64 = copy.deepcopy(emb).double()
  datas, _ = synth.make_batch(2, crop, num_classes=classes, seed=77)
  o64, og, order = {}, {}, []

  def hook(store, name, track):
    def f(m, i, o):
      if torch.is_tensor(o) and name not in store:
        store[name] = o.detach().double().cpu()
        if track:
          order.append(name)
    return f
  for n, m in e64.named_modules():
    m.register_forward_hook(hook(o64, n, True))
  eg = copy.deepcopy(emb).cuda()
  img = datas['image'].cuda()
  if nhwc:
    eg = eg.to(memory_format=torch.channels_last)
    img = img.contiguous(memory_format=torch.channels_last)
  for n, m in eg.named_modules():
    m.register_forward_hook(hook(og, n, False))
  e64.generate_embeddings({'image': datas['image'].double()})
  img.requires_grad_(False)
  eg.generate_embeddings({'image': img})
  prev = 0.0
  for n in order:
    if n in og:
      a, b = og[n], o64[n]
      r = ((a - b).norm() / b.norm().clamp(min=1e-300)).item()
      flag = '  <==' if r > 4 * max(prev, 1e-7) and r > 2e-6 else ''
      print('%-52s %.3e  %s%s' % (n or '(model)', r, tuple(b.shape), flag))
      prev = r


if __name__ == '__main__':
  main()
