#!/usr/bin/env python3
"""Timing of the full-resolution memory-bank pass (prototype.py:107-211)
on one synthetic image: sliding-window ResNet-101 DeepLab-v2 embedding, overlap
averaging, k-means over the whole image, prototypes + majority labels -- HIP path on the
GPU next to the oracle's restatement on the host cores."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--pad', type=int, nargs=2, default=[769, 1025])
  ap.add_argument('--crop', type=int, default=513)
  ap.add_argument('--stride', type=int, default=342)
  ap.add_argument('--clusters', type=int, nargs=2, default=[12, 12])
  ap.add_argument('--no-cpu', action='store_true')
  a = ap.parse_args()
  from spml_amd import inference
  from spml_amd.train import build_models, voc12_scribble_config
  dev = torch.device('cuda', 0)
  cfg = voc12_scribble_config(batch_size=1, use_syncbn=False)
  cfg.network.kmeans_num_clusters = list(a.clusters)
  torch.manual_seed(235)
  emb_model, _ = build_models(cfg, softmax_head=False)
  emb_model.eval()
  g = torch.Generator().manual_seed(1)
  image = torch.randn(1, 3, a.pad[0], a.pad[1], generator=g)
  valid = (a.pad[0] - 20, a.pad[1] - 30)
  sem = torch.randint(0, 21, (valid[0] // 64 + 1, valid[1] // 64 + 1), generator=g)
  sem = sem.repeat_interleave(64, 0).repeat_interleave(64, 1)[:valid[0], :valid[1]].contiguous()
  crop, stride = (a.crop, a.crop), (a.stride, a.stride)

  gmodel = emb_model.to(dev)
  gimage = image.to(dev)

  def gpu_pass():
    t = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    emb = inference.embed_full_resolution(gmodel, gimage, crop, stride)
    torch.cuda.synchronize(); t['embed_ms'] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    fake = torch.full((1,) + tuple(a.pad), 255, dtype=torch.long, device=dev)
    fake[:, :valid[0], :valid[1]] = 0
    with torch.no_grad():
      out = gmodel.generate_clusters(emb, fake, fake)
    torch.cuda.synchronize(); t['kmeans_ms'] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    import spml_amd.utils.segsort.common as sc
    pr = sc.calculate_prototypes_from_labels(out['cluster_embedding'], out['cluster_index'])
    _, lab = sc.find_majority_label_index(sem.to(dev), out['cluster_index'])
    torch.cuda.synchronize(); t['prototypes_ms'] = (time.perf_counter() - t0) * 1e3
    t['prototypes'] = int(pr.shape[0])
    return t

  gpu_pass()
  t = gpu_pass()
  res = {'image': list(a.pad), 'crop': a.crop, 'stride': a.stride, 'clusters': a.clusters,
         'gpu': {k: (round(v, 2) if isinstance(v, float) else v) for k, v in t.items()}}
  res['gpu']['total_ms'] = round(t['embed_ms'] + t['kmeans_ms'] + t['prototypes_ms'], 2)
  if not a.no_cpu:
    from oracle import spml_oracle as O
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    cmodel = emb_model.cpu()
    fn = lambda c: cmodel.generate_embeddings({'image': c}, resize_as_input=True)['embedding']
    t0 = time.perf_counter()
    O.full_resolution_prototypes(fn, image, sem, crop, stride, a.clusters, 2048)
    res['cpu_oracle_total_ms'] = round((time.perf_counter() - t0) * 1e3, 1)
    res['cpu_threads'] = torch.get_num_threads()
  print(json.dumps(res))


if __name__ == '__main__':
  main()
