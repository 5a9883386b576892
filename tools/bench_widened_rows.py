#!/usr/bin/env python3
"""Timing of the two rows round 6 added: the stage-2 classifier step
(`pyscripts/train/train_classifier.py:139-169`; ResNet-101 DeepLab-v2 frozen, batch 16, 513 x 513) and the
full-resolution kNN label inference of one image (`pyscripts/inference/inference.py:145-237`; 769 x 1025 padded image,
513 crops, 12 x 12 clusters, a memory bank of 20 000 prototypes) -- HIP path on the GPU next to the oracle's
restatement on the host cores (a bounded sample: batch 2 for the step)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--no-cpu', action='store_true')
  ap.add_argument('--bank', type=int, default=20000)
  a = ap.parse_args()
  from spml_amd import inference, synth
  from spml_amd.models.predictions.segsort import segsort
  from spml_amd.models.predictions.softmax_classifier import softmax_classifier
  from spml_amd.train import ClassifierTrainer, build_models, voc12_scribble_config
  dev = torch.device('cuda', 0)
  res = {}

  # ---- stage 2 ----
  cfg = voc12_scribble_config(batch_size=16, crop=513, max_iteration=4000, use_syncbn=False)
  torch.manual_seed(235)
  emb, _ = build_models(cfg, softmax_head=False)
  pred = softmax_classifier(cfg)
  tr = ClassifierTrainer(cfg, dev, channels_last=True, models=(emb, pred))
  batches = [synth.make_batch(16, 513, seed=300 + i, device=dev, palette=(1, 3)) for i in range(2)]
  for d, _ in batches:
    d['image'] = d['image'].contiguous(memory_format=torch.channels_last)
  for i in range(4):
    tr.step(*batches[i % 2])
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  n = 10
  for i in range(n):
    out = tr.step(*batches[i % 2])
  torch.cuda.synchronize()
  ms = (time.perf_counter() - t0) / n * 1e3
  res['stage2_step'] = {'gpu_ms_per_step': round(ms, 2), 'gpu_images_per_s': round(16e3 / ms, 1), 'batch': 16,
                        'loss': round(float(out['loss']), 4)}
  if not a.no_cpu:
    from oracle.cpu_step import CpuClassifierStep
    from spml_amd.nn.optimizer import SGD
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    cfg2 = voc12_scribble_config(batch_size=2, crop=513, max_iteration=4000, use_syncbn=False)
    torch.manual_seed(235)
    emb_c, _ = build_models(cfg2, softmax_head=False)
    pred_c = softmax_classifier(cfg2)
    opt = SGD(emb_c.get_params_lr() + pred_c.get_params_lr(), lr=1, momentum=0.9, weight_decay=5e-4)
    step = CpuClassifierStep(emb_c, pred_c, cfg2, opt)
    d, t = synth.make_batch(2, 513, seed=300)
    t0 = time.perf_counter()
    step.step(d, t, 3e-4)
    dt = time.perf_counter() - t0
    res['stage2_step']['cpu_oracle_images_per_s'] = round(2.0 / dt, 3)
    res['stage2_step']['cpu_sample'] = 'one step of batch 2 on %d host threads: %.1f s' % (torch.get_num_threads(), dt)
  del tr, emb, pred

  # ---- full-resolution kNN inference ----
  cfg = voc12_scribble_config(batch_size=1, use_syncbn=False)
  cfg.network.kmeans_num_clusters = [12, 12]
  torch.manual_seed(235)
  emb_model, _ = build_models(cfg, softmax_head=False)
  emb_model.eval()
  predictor = segsort(cfg).eval()
  g = torch.Generator().manual_seed(1)
  pad, valid, crop, stride = (769, 1025), (749, 995), (513, 513), (342, 342)
  image = torch.randn(1, 3, pad[0], pad[1], generator=g)
  bank = torch.nn.functional.normalize(torch.randn(a.bank, 64, generator=g), dim=1)
  bank_lab = torch.randint(0, 21, (a.bank,), generator=g)
  gmodel = emb_model.to(dev).to(memory_format=torch.channels_last)
  gpred = predictor.to(dev)
  gimage, gbank, gbank_lab = image.to(dev), bank.to(dev), bank_lab.to(dev)
  run = lambda: inference.predict_full_resolution(gmodel, gpred, gimage, valid, crop, stride, gbank, gbank_lab)
  run()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(3):
    out = run()
  torch.cuda.synchronize()
  ms = (time.perf_counter() - t0) / 3 * 1e3
  res['full_resolution_knn_inference'] = {'gpu_ms_per_image': round(ms, 2), 'image': list(pad), 'valid': list(valid),
                                          'memory_bank': a.bank, 'segments': int(out['cluster_index'].max()) + 1}
  if not a.no_cpu:
    from oracle import spml_oracle as O
    cmodel = emb_model.cpu().to(memory_format=torch.contiguous_format)
    fn = lambda c: cmodel.generate_embeddings({'image': c}, resize_as_input=True)['embedding']
    t0 = time.perf_counter()
    O.predict_full_resolution(fn, image, valid, crop, stride, [12, 12], 2048, bank, bank_lab)
    res['full_resolution_knn_inference']['cpu_oracle_ms_per_image'] = round((time.perf_counter() - t0) * 1e3, 1)
    res['full_resolution_knn_inference']['cpu_threads'] = torch.get_num_threads()
  print(json.dumps(res))


if __name__ == '__main__':
  main()
