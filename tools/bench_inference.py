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


def walk(a):
  from spml_amd import _ffi, inference
  dev = torch.device('cuda', 0)
  h, w = a.walk
  g = torch.Generator().manual_seed(2)
  base = torch.randn(1, 64, h // 4 + 2, w // 4 + 2, generator=g)
  embs = [torch.nn.functional.interpolate(base, size=(h, w), mode='bilinear', align_corners=False)
          + 0.2 * torch.randn(1, 64, h, w, generator=g) for _ in range(2)]
  cam = torch.rand(21, h, w, generator=g)
  gembs, gcam = [e.to(dev) for e in embs], cam.to(dev)
  stacked = torch.stack([(e / torch.norm(e, dim=1)).reshape(64, -1) for e in gembs], 0).contiguous()

  def timed(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
      out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out
  t_aff, _ = timed(lambda: _ffi.affinity_transition(stacked))
  t_all, got = timed(lambda: inference.affinity_random_walk(gembs, gcam))
  res = {'map': [h, w], 'n': h * w, 'views': 2, 'gpu': {'affinity_kernel_ms': round(t_aff, 3),
                                                      'total_ms': round(t_all, 3)}}
  if not a.no_cpu:
    from oracle import spml_oracle as O
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    t0 = time.perf_counter()
    want = O.affinity_random_walk(embs, cam)
    res['cpu_oracle_total_ms'] = round((time.perf_counter() - t0) * 1e3, 1)
    res['max_rel_err'] = float(((got.cpu() - want).abs() / want.abs().clamp_min(1e-6)).max())
  print(json.dumps(res))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--pad', type=int, nargs=2, default=[769, 1025])
  ap.add_argument('--crop', type=int, default=513)
  ap.add_argument('--stride', type=int, default=342)
  ap.add_argument('--clusters', type=int, nargs=2, default=[12, 12])
  ap.add_argument('--no-cpu', action='store_true')
  ap.add_argument('--nchw', action='store_true', help='keep the model in NCHW (library convolutions everywhere)')
  ap.add_argument('--walk', type=int, nargs=2, default=None, metavar=('H8', 'W8'),
                  help='time the affinity random walk (N3) on an H8 x W8 map instead')
  a = ap.parse_args()
  if a.walk:
    return walk(a)
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
  if not a.nchw:                                  # channels-last: matrix-core units + NHWC library kernels
    gmodel = gmodel.to(memory_format=torch.channels_last)
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
