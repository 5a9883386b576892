#!/usr/bin/env python3
"""Framework convolution ops of a bench recipe with their shapes and device time (torch profiler):
what still runs on the library.   python tools/profile_conv_ops.py voc|tag|stress|densepose [bench flags]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
steps = 2
recipe = sys.argv[1] if len(sys.argv) > 1 else 'voc'
sys.argv = ['bench.py', '--recipe', recipe, '--steps', str(steps), '--warmup', '2', '--no-cpu-baseline', '--no-kmeans'] + sys.argv[2:]
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
  bench.main()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
  if e.key in ('aten::convolution', 'aten::convolution_backward') and e.device_time_total > 0:
    rows.append((e.device_time_total / 1e3 / (steps + 2), e.count // (steps + 2), e.key, str(e.input_shapes)[:110]))
tot = sum(r[0] for r in rows)
print('framework convolutions: %.2f ms per step' % tot)
for r in sorted(rows, reverse=True)[:24]:
  print('%7.3f ms/step %2d x %-28s %s' % r)
