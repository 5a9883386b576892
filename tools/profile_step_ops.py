"""Attribute the GPU time of one training step to framework ops (torch.profiler), to see what is
left outside libspml_hip and the convolution library."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import spml_amd
from spml_amd import synth
from spml_amd.train import Trainer, voc12_scribble_config

dev = torch.device('cuda', 0)
cfg = voc12_scribble_config(batch_size=16, crop=513)
torch.manual_seed(235)
tr = Trainer(cfg, dev, softmax_head=True, channels_last=True)
bs = [synth.make_batch(16, 513, num_classes=cfg.dataset.num_classes, seed=235 + i, device=dev) for i in range(2)]
for d, _ in bs:
  d['image'] = d['image'].contiguous(memory_format=torch.channels_last)
for i in range(3):
  tr.step(*bs[i % 2])
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
  for i in range(2):
    tr.step(*bs[i % 2])
  torch.cuda.synchronize()
import sys as _s
print(prof.key_averages().table(sort_by=(_s.argv[1] if len(_s.argv) > 1 else 'cuda_time_total'), row_limit=45, max_name_column_width=60))
