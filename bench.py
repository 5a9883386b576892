#!/usr/bin/env python3
"""Benchmark of the SPML pixel-to-segment contrastive hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  N > 1: one rank per GPU over RCCL.  Either form works: under
  `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (the ranks read
  RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*) or plainly as `python bench.py --gpus N ...`, which
  starts the N ranks itself (the counterpart of the reference's one command driving all GPUs,
  pyscripts/train/train.py:131-139,167,211).  The world size must equal --gpus; anything else aborts.

Metric (BASELINE.json): images/sec at 513x513 of the full training step --
ResNet-101 DeepLab-v2 forward/backward in fp32 (the reference trains in fp32),
per-image spherical k-means, prototypes, the three contrastive losses + softmax
head, SGD step -- on the "VOC12 scribble, ResNet-101, batch 16 per GPU" config,
synthetic 21-class batches, weak scaling.  Also reported: k-means iterations/sec
on the 513x513x(256+2) roofline configuration, the HBM roofline fraction of the
fused k-means pass kernel and the CPU oracle timed on this node's host cores.
Prints ONE JSON line on rank 0.

How the roofline kernel is timed: HIP events on the launch stream around ONE library call that issues 60
back-to-back launches of the fused pass kernel (`roofline.us_per_launch`: the mean launch period, what a rocprofv3
kernel trace of this command lists for the kernel plus the kernel boundary); beside it the workgroups' own
s_memrealtime stamps over all fused launches of five whole k-means calls (`us_per_launch_device_stamps`), the shader
clock a spinning wave saw meanwhile, and the HBM bytes per launch from profiles/kmeans_pass_pmc_traffic.json
(rocprofv3 --pmc passes, tools/refresh_profiles.py).  `frac_iteration` prices the WHOLE call (HIP events around
spml_kmeans_run_f32: seed pass, finalize kernels, label conversions included) against the same algorithmic bytes
per iteration; `kmeans_iters_per_s_coherent` is the same call on spatially coherent rows.

Other recipes (not the headline): --recipe tag (BASELINE config 3), densepose
(config 4: --batch 8 --crop 769), stress (config 5: 1025 crop, 512-d embedding,
32x32 = 1024 centroids; --batch 2; its roofline is the MFMA-bound E-step)."""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GBPS = 6300.0    # what a streaming kernel reaches on this part (same guide): reported next to `frac`
MFMA_F16_PEAK_TFLOPS = 2500.0   # dense f16/bf16 MFMA peak (MI355X_MICROARCH.md)
# HBM bytes per launch of the fused k-means pass at the roofline configuration: counters cannot be read from
# inside this process; tools/pmc_kmeans.sh collects the rocprofv3 PMC passes (FETCH_SIZE x2, the gfx950
# correction, + WRITE_SIZE) and tools/refresh_profiles.py writes them here
PMC_TRAFFIC_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'kmeans_pass_pmc_traffic.json')


# rocprofv3 --kernel-trace --stats of exactly the driver's command (`python bench.py --gpus 1 --steps 20 --warmup 5`),
# committed per round by tools/run_round_checks.sh + tools/refresh_profiles.py: the average duration of the roofline
# kernel over ALL its launches of that process (clock ramps, launches inside whole k-means calls, profiler overhead
# included) is the figure a reader recomputes from profiles/ -- `roofline.frac` never exceeds it
ROCPROF_STATS_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r06_bench_driver_cmd_kernel_stats.csv')


def rocprof_mean_us(kernel_prefix):
  """(average us, calls) of the kernel whose name contains `kernel_prefix` in the committed stats, or (None, 0)."""
  import csv
  try:
    with open(ROCPROF_STATS_FILE, newline='') as f:
      for row in csv.DictReader(f):
        if kernel_prefix in row['Name']:
          return float(row['AverageNs']) * 1e-3, int(row['Calls'])
  except (OSError, ValueError, KeyError):
    pass
  return None, 0


def pmc_traffic():
  try:
    with open(PMC_TRAFFIC_FILE) as f:
      t = json.load(f)
    return int(t['bytes_per_launch']), t.get('source', PMC_TRAFFIC_FILE)
  except (OSError, ValueError, KeyError):
    return None, 'no PMC record (profiles/kmeans_pass_pmc_traffic.json)'


def parse():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=8)
  ap.add_argument('--warmup', type=int, default=3)
  ap.add_argument('--batch', type=int, default=None, help='images per GPU (default: 16; stress: 2)')
  ap.add_argument('--crop', type=int, default=None, help='default: 513; stress: 1025')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--dist-backend', default='nccl', choices=['nccl', 'gloo'],
                  help="collective backend ('nccl' = RCCL over xGMI; 'gloo' only exists so that a test can run "
                       "two ranks on ONE device, which RCCL refuses)")
  ap.add_argument('--share-gpus', action='store_true',
                  help='testing only: allow more ranks than devices (rank r uses device r %% device_count)')
  ap.add_argument('--no-kmeans', action='store_true')
  ap.add_argument('--dense-tags', action='store_true',
                  help='synthetic batch of rounds 1-4: every region draws from all classes (image tag sets nearly '
                       'always intersect: the co-occurrence term is identically zero); default: 1-3 object classes '
                       'per image')
  ap.add_argument('--miopen-find', action='store_true', help='cudnn.benchmark (MIOpen find mode)')
  ap.add_argument('--no-miopen-db', action='store_true',
                  help='ignore the tuned MIOpen find-db shipped in spml_amd/miopen_db')
  ap.add_argument('--no-mc-conv', action='store_true',
                  help='res4 / res5 bottleneck units on the framework (MIOpen fp32) convolutions instead of '
                       'this repository\'s matrix-core kernels (csrc/conv.hip)')
  ap.add_argument('--channels-last', dest='channels_last', action='store_true', default=None,
                  help='NHWC activations / weights: what the matrix-core units and MIOpen\'s tuned NHWC solvers '
                       'take (default; the shipped find-db holds the NHWC search results of all four recipes)')
  ap.add_argument('--nchw', dest='channels_last', action='store_false', help='NCHW activations / weights')
  ap.add_argument('--recipe', default='voc', choices=['voc', 'tag', 'densepose', 'stress'],
                  help="'voc': headline VOC12 scribble config; 'tag': BASELINE config 3 (image-tag "
                       "recipe); 'densepose': config 4 (use --batch 8 --crop 769); 'stress': config 5 "
                       "(1025 crop, 512-d embedding, 1024 centroids)")
  return ap.parse_args()


def _event_time_ms(fn, reps):
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps


def conv_roofline(device, batch=16, side=65, cin=256, cout=256, taps=9, dil=2, reps=10):
  """The step's largest kernel since the backbone's hot units moved to the matrix cores: the
  3x3 (dilation 2) convolution of a res4 unit, `conv_gemm` of csrc/conv.hip, timed with HIP
  events on the stream it is launched on.  MFMA-bound: achieved = the f16 matrix work (3 exact
  products per multiply-add of the fp32-class split) / time, peak = dense f16 MFMA."""
  from spml_amd import _ffi
  g = torch.Generator(device=device).manual_seed(7)
  x = torch.randn(batch, cin, side, side, device=device, generator=g).clamp_min(0)
  x = x.contiguous(memory_format=torch.channels_last)
  w = torch.randn(cout, cin, 3 if taps == 9 else 1, 3 if taps == 9 else 1, device=device, generator=g) * 0.02
  xa = _ffi.hl8_from_f32(x)
  wf, _ = _ffi.hl8_weight(w)
  fn = lambda: _ffi.conv_hl8(xa, wf, batch, side, side, taps, dil)
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  ms = _event_time_ms(fn, reps)
  madds = float(batch * side * side) * cin * cout * taps
  return {'bound': 'mfma', 'kernel': 'conv_gemm (3x3 d%d, %d->%d, %dx%dx%d pixels, split-f16 x3)' % (
              dil, cin, cout, batch, side, side),
          'achieved': round(3 * 2 * madds / ms / 1e9, 1), 'peak': 2500.0, 'unit': 'TFLOP/s',
          'frac': round(3 * 2 * madds / ms / 1e9 / 2500.0, 4), 'traffic': None,
          'us_per_launch': round(ms * 1e3, 1), 'fp32_equivalent_tflops': round(2 * madds / ms / 1e9, 1),
          'note': 'power-bound: the bare MFMA loop sustains 1.87 PFLOP/s (0.75 of the peak) on this part, '
                  'DESIGN 5c'}


def coherent_rows(p_side, d, device, g):
  """Unit rows with spatial structure (what embeddings are; tools/bench_kmeans.py): 0.3 noise + a smooth field."""
  p = p_side * p_side
  x = torch.randn(p, d, device=device, generator=g)
  yy = torch.linspace(0, 1, p_side, device=device).view(-1, 1).expand(p_side, p_side).reshape(-1)
  xx = torch.linspace(0, 1, p_side, device=device).view(1, -1).expand(p_side, p_side).reshape(-1)
  base = torch.randn(8, d, device=device, generator=g)
  w = torch.stack([torch.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / 0.03)
                   for cy, cx in torch.rand(8, 2, generator=torch.Generator().manual_seed(1)).tolist()], 1)
  x = 0.3 * x + w @ base
  return x / x.norm(dim=1, keepdim=True)


def kmeans_roofline(device, side=513, c=256, k=6, iters=10, reps=20):
  """Config R of SURVEY 8d: one 513x513 map, D = 256 + 2, K = 36, 10 iterations."""
  from spml_amd import _ffi
  d = c + 2
  p = side * side
  g = torch.Generator(device=device).manual_seed(235)
  x = torch.randn(p, d, device=device, generator=g)
  x = x / x.norm(dim=1, keepdim=True)
  init = _ffi.kmeans_init_grid(side, side, k, k, device).view(-1)
  off = torch.tensor([0, p], dtype=torch.int64, device=device)
  kk = k * k
  # ... and the same call on spatially coherent rows.  The two are timed in ALTERNATING blocks (A B A B, each block
  # = reps / 2 back-to-back calls on one data set, HIP events around the block, no host synchronisation in between):
  # the pass kernels are data-independent, but the shader clock drifts by 10-15 % over a second of k-means calls,
  # and timing one data set after the other charged that drift to whichever came second (noise 85.6 / coherent 82.7 us
  # per iteration cold, 74.9 / 74.2 warm, in one process).  (Call-by-call alternation reads 4-5 us more per iteration
  # for both: two 272-MB tensors evict each other's tiles from the 256-MB MALL.)
  xc = coherent_rows(side, d, device, g)
  path = _ffi.kmeans_path_name(p, d, kk, 1, p, iters)
  half = max(reps // 2, 1)
  # warm-up straight in front of the timed blocks, no idle gap in between: 6 x half = 60 calls = 45 ms -- the firmware
  # takes tens of ms of uninterrupted launches to settle the shader clock (with 10 calls the first block paid the ramp:
  # 13.4-13.5 k iterations/s on noise against 13.8 k on the coherent rows measured second; BENCH_KM_WARM_CALLS)
  for _ in range(int(os.environ.get('BENCH_KM_WARM_CALLS', 6 * half))):
    _ffi.kmeans_run(xc, off, p, kk, init, iters)
  ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
  ev[0].record()
  for blk in range(4):
    data = x if blk % 2 == 0 else xc
    for _ in range(half):
      _ffi.kmeans_run(data, off, p, kk, init, iters)
    ev[blk + 1].record()
  torch.cuda.synchronize()
  run_ms = (ev[0].elapsed_time(ev[1]) + ev[2].elapsed_time(ev[3])) / (2 * half)
  coherent_ms = (ev[1].elapsed_time(ev[2]) + ev[3].elapsed_time(ev[4])) / (2 * half)
  del xc
  bytes_pass = p * d * 4 + p * 8 + 2 * kk * d * 4     # B_iter
  # (a) the roofline kernel alone, HIP events on its stream: the fused pass (E-step + M-step accumulation, X read
  # once) launched back to back through the exported entry point with SPML_KMEANS_PASS_ONLY -- the mean launch
  # period = what `rocprofv3 --kernel-trace --stats` lists for the kernel, plus the ~1.5-us kernel boundary.  A
  # one-wave probe on a second stream reads the shader clock meanwhile (cycles against the 100-MHz wall clock).
  ws = _ffi.kmeans_workspace(x, off, p, kk)
  _ffi.kmeans_preconvert(x, off, p, kk, ws)
  cent = torch.nn.functional.normalize(torch.randn(1, kk, d, device=device, generator=g), dim=-1)
  out = _ffi.kmeans_fused_pass(x, off, p, cent, ws=ws, preconverted=True)
  n_launch = 60                                                 # (issued by ONE library call: no host gaps)
  pass_only = lambda: _ffi.kmeans_fused_pass(x, off, p, cent, ws=ws, preconverted=True, out=out,
                                             flags=256 | (n_launch << 16))
  pass_only()
  torch.cuda.synchronize()
  side_stream = torch.cuda.Stream(device=device)

  warm_us = []                                                  # launch period of every untimed warm-up burst

  def timed_block():
    """One block of n_launch back-to-back launches: (us per launch, shader clock in MHz during the block)."""
    # untimed warm-up straight in front (no host synchronisation in between): the firmware needs ~10 ms of these
    # launches to settle the shader clock of a part that was idle or ran other kernels -- behind ONE burst (3.3 ms) the
    # five blocks of a run read 50-64 us at 1.8-2.25 GHz, always the same blocks slow; behind six (20 ms) 48.4-50.8 us
    # (BENCH_KM_WARM_BURSTS; profiles/r05_kmeans_clock.md)
    # (the warm-up bursts are timed too, burst by burst: they do not enter the settled-clock figure, but the mean over
    # ALL launches of this process -- ramps included -- is reported next to it and caps `frac`)
    n_warm = int(os.environ.get('BENCH_KM_WARM_BURSTS', '6'))
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_warm + 1)]
    evs[0].record()
    for i in range(n_warm):
      pass_only()
      evs[i + 1].record()
    probe = _ffi.clock_probe(device, 4000, side_stream)        # 4 ms: covers the timed launches
    ms = _event_time_ms(pass_only, 1) / n_launch
    torch.cuda.synchronize()
    warm_us.extend(evs[i].elapsed_time(evs[i + 1]) * 1e3 / n_launch for i in range(n_warm))
    cyc, ticks = [int(v) for v in probe.tolist()]
    return ms * 1e3, 100.0 * cyc / max(ticks, 1)

  # five blocks spread over the k-means section (the launch period drifts inside one process; every
  # block carries the clock it ran at, and duration x clock says whether the kernel follows the shader clock)
  blocks = [timed_block()]
  # (b) the passes of whole runs from their per-workgroup device time stamps: every fused launch of `reps` runs
  durs = []
  for r in range(reps):
    _, dur = _ffi.kmeans_run_profiled(x, off, p, kk, init, iters)
    durs.append(dur)
    if r < 4:
      blocks.append(timed_block())
  while len(blocks) < 5:
    blocks.append(timed_block())
  dur = torch.stack(durs)
  fused = dur[:, 1:-1]
  us_iter = run_ms * 1e3 / iters
  b_us = sorted(b[0] for b in blocks)
  # the MEDIAN of the five blocks: a block that catches a stall from outside the kernel (under rocprofv3 one block of a
  # round-5 run read 654 us per launch -- a buffer flush -- next to 52-59 us for the other four) must not decide the
  # roofline figure; the mean of the blocks is reported next to it
  pass_us = b_us[len(b_us) // 2]
  pass_us_mean = sum(b_us) / len(b_us)
  clock_mhz = sum(b[1] for b in blocks) / len(blocks)
  # the mean over EVERY launch of the kernel this process made and could time: all bursts (warm-up ones included: the
  # clock ramp) by HIP events, the fused launches inside the whole runs by their device stamps + the ~3 us of dispatch
  # ramp and end-of-kernel write-back the stamps do not see
  in_call_us = fused.mean().item() + 3.0
  n_burst, n_call = (len(warm_us) + len(b_us)) * n_launch, fused.numel()
  all_us = (sum(warm_us) + sum(b_us)) * n_launch / max(n_burst, 1)
  mean_all_us = (all_us * n_burst + in_call_us * n_call) / (n_burst + n_call)
  fused_kernel = 'kmeans_pass64<3, 8, 1, true>' if path.endswith('v4p') else 'kmeans_pass16<3, 8, 1, true>'
  prof_us, prof_calls = rocprof_mean_us(fused_kernel)
  settled_us = pass_us
  # `frac`: the LOWEST of the three -- settled-clock median, mean over all launches of this process, and the average
  # duration in the committed rocprofv3 stats of the driver's command
  pass_us = max([settled_us, mean_all_us] + ([prof_us] if prof_us else []))
  achieved = bytes_pass / (pass_us * 1e-6) / 1e9
  # (c) the exported single pass as a caller uses it (centroid split + pass + slab reduction + label widening)
  export_ms = _event_time_ms(
      lambda: _ffi.kmeans_fused_pass(x, off, p, cent, ws=ws, preconverted=True, out=out), 8)
  traffic, traffic_source = pmc_traffic()
  return {
      'iters_per_s': iters / (run_ms * 1e-3),
      'iters_per_s_coherent': iters / (coherent_ms * 1e-3),
      'path': path,
      'roofline': {'bound': 'hbm', 'kernel': '%s (fused E+M pass, 513x513x258, K=36; the E-only final pass of a call is the '
                                             'same kernel without the accumulation)' % fused_kernel,
                   'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                   'frac': round(achieved / HBM_PEAK_GBPS, 4), 'traffic': traffic,
                   # the launch moves `traffic` bytes, not `algorithmic_bytes`: its HBM rate against the ~6.3 TB/s a
                   # streaming kernel reaches on this part (information; `frac` is the roofline figure)
                   'hbm_rate_of_achievable': None if traffic is None else
                                             round(traffic / (settled_us * 1e-6) / 1e9 / HBM_ACHIEVABLE_GBPS, 4),
                   'traffic_source': traffic_source,
                   'timing': 'HIP events on the launch stream around bursts of %d back-to-back launches of the pass kernel, five '
                             'blocks of seven bursts spread over the k-means section.  `frac` = algorithmic bytes / the LARGEST of: '
                             '(a) the median of the five settled-clock bursts (each behind six warm-up bursts = 20 ms), (b) the mean '
                             'over all launches this process made (every burst, warm-up ones included, + the fused launches inside '
                             'the whole k-means runs), (c) the AverageNs of the kernel in the committed rocprofv3 --kernel-trace '
                             '--stats of the driver command (profiles/r06_bench_driver_cmd_kernel_stats.csv)' % n_launch,
                   'us_per_launch': round(pass_us, 2),
                   'us_per_launch_settled_median': round(settled_us, 2),
                   'frac_settled_clock': round(bytes_pass / (settled_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
                   'us_per_launch_mean_all_launches': round(mean_all_us, 2),
                   'launches_in_that_mean': {'bursts': n_burst, 'inside_whole_runs': n_call},
                   'us_per_launch_rocprof_mean': None if prof_us is None else round(prof_us, 2),
                   'frac_rocprof_mean': None if prof_us is None else round(bytes_pass / (prof_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
                   'rocprof_calls': prof_calls,
                   'us_per_launch_mean_of_blocks': round(pass_us_mean, 2),
                   'us_per_launch_min_median_max': [round(b_us[0], 2), round(b_us[len(b_us) // 2], 2), round(b_us[-1], 2)],
                   'blocks_us_mhz_kcycles': [[round(u, 2), round(m), round(u * m / 1e3, 1)] for u, m in blocks],
                   'shader_clock_mhz_during_the_launches': round(clock_mhz, 0),
                   'us_per_launch_device_stamps': round(fused.mean().item(), 2),
                   'us_per_launch_device_stamps_note': 'all %d fused launches of %d whole runs, max end - min start of '
                                                       'the workgroups (no dispatch ramp, no end-of-kernel '
                                                       'write-back: ~3 us below the profiler\'s figure)' % (
                                                           fused.numel(), reps),
                   'us_seed_pass': round(dur[:, 0].mean().item(), 1), 'us_final_pass': round(dur[:, -1].mean().item(), 1),
                   'us_per_iteration': round(us_iter, 2),
                   'frac_iteration': round(bytes_pass / (us_iter * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
                   'us_per_iteration_coherent_rows': round(coherent_ms * 1e3 / iters, 2),
                   'us_per_call_fused_pass_export': round(export_ms * 1e3, 2),
                   'algorithmic_bytes': bytes_pass},
      'x': x, 'init': init, 'k': kk, 'iters': iters,
  }


def kmeans_stress_roofline(device, side=258, c=512, k=32, iters=10, reps=5, imgs=1):
  """BASELINE config 5: one 258x258 map (1025 crop), D = 512 + 2, K = 32x32 = 1024: the E-step
  is MFMA-bound (split-f16: 3 f16 MFMA passes per product); reported against the f16 peak."""
  from spml_amd import _ffi
  d = c + 2
  p = side * side
  g = torch.Generator(device=device).manual_seed(235)
  x = torch.randn(imgs * p, d, device=device, generator=g)
  x = x / x.norm(dim=1, keepdim=True)
  init = _ffi.kmeans_init_grid(side, side, k, k, device).view(-1).repeat(imgs)
  off = (torch.arange(imgs + 1, device=device) * p).to(torch.int64)
  kk = k * k
  for _ in range(2):
    _ffi.kmeans_run(x, off, p, kk, init, iters)
  run_ms = _event_time_ms(lambda: _ffi.kmeans_run(x, off, p, kk, init, iters), reps)
  path = _ffi.kmeans_last_path()
  cent = torch.nn.functional.normalize(torch.randn(imgs, kk, d, device=device, generator=g), dim=-1)
  # the MFMA roofline kernel: the exact E-step over ALL pixels (flag 64 = no screening pass)
  _ffi.kmeans_assign(x, off, p, cent, flags=64)
  assign_ms = _event_time_ms(lambda: _ffi.kmeans_assign(x, off, p, cent, flags=64), 8)
  # what a run actually does: hi-half screening + the exact kernel over the ambiguous pixels
  _ffi.kmeans_assign(x, off, p, cent)
  screened_ms = _event_time_ms(lambda: _ffi.kmeans_assign(x, off, p, cent), 8)
  flops = 2.0 * imgs * p * d * kk * 3                 # f16 MFMA flops of one exact E-step (3 passes)
  achieved = flops / (assign_ms * 1e-3) / 1e12
  return {
      'iters_per_s': iters / (run_ms * 1e-3), 'path': path,
      'roofline': {'bound': 'mfma', 'kernel': 'bigk_assign<33,1> (exact E-step, %d x 258x258x514, K=1024; the '
                                              'timed call also splits the prototypes and decodes the labels)' % imgs,
                   'achieved': round(achieved, 1), 'peak': MFMA_F16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                   'frac': round(achieved / MFMA_F16_PEAK_TFLOPS, 4), 'traffic': None,
                   'us_per_launch': round(assign_ms * 1e3, 1),
                   'us_per_estep_with_screening': round(screened_ms * 1e3, 1),
                   'us_per_iteration': round(run_ms * 1e3 / iters, 1),
                   'algorithmic_flops': flops},
      'x': x[:p], 'init': init[:p], 'k': kk, 'iters': iters,
  }


def cpu_baseline(km, quick_kmeans_iters=None):
  """The oracle (a CPU restatement of the reference, kind 'port') on this node's
  host cores: one training step of config 1 (batch 2, 513x513, ResNet-101) and
  the k-means of the roofline configuration."""
  from oracle import spml_oracle as O
  from oracle.cpu_step import CpuStep
  from spml_amd import synth
  from spml_amd.nn.optimizer import SGD
  from spml_amd.train import build_models, voc12_scribble_config
  # torch's CPU kernels stop scaling (and then collapse) far below the 256 logical
  # cores of the GPU node: use at most 32 threads and say so.
  cores = min(os.cpu_count() or 1, 32)
  torch.set_num_threads(cores)
  out = {'cores': cores, 'host_logical_cpus': os.cpu_count(), 'kind': 'port'}
  if km is not None:
    x, init = km['x'].cpu(), km['init'].cpu()
    its = quick_kmeans_iters or km['iters']
    t0 = time.perf_counter()
    O.kmeans_with_initial_labels(x, init, km['k'], its)
    out['kmeans_iters_per_s'] = its / (time.perf_counter() - t0)
  cfg = voc12_scribble_config(batch_size=2, use_syncbn=False)
  torch.manual_seed(235)
  emb, pred = build_models(cfg, softmax_head=True)
  opt = SGD(emb.get_params_lr() + pred.get_params_lr(), lr=1, momentum=0.9, weight_decay=5e-4)
  step = CpuStep(emb, pred, cfg, opt, softmax_head=True)
  emb.train(); pred.train()
  batches = [synth.make_batch(2, 513, seed=235 + i) for i in range(2)]
  t0 = time.perf_counter()
  for datas, targets in batches:            # the second step also runs the memory bank
    step.step(datas, targets, 3e-4)
  dt = time.perf_counter() - t0
  out.update({'value': round(4.0 / dt, 4), 'unit': 'images/s',
              'sample': '2 training steps of config 1 (batch 2, 513x513, ResNet-101 DeepLab-v2, '
                        '3 contrastive losses + softmax head, memory bank, fwd+bwd+SGD): %.1f s' % dt})
  return out


WORKLOADS = {
    'voc': ('VOC12 scribble recipe (train_spml_scribble.sh; PREDICTION_TYPES=segsort, which train.py:31 '
            'binds to the softmax-head variant), ResNet-101 DeepLab-v2, %dx%d crop, 21 classes, batch %d '
            'per GPU, dim 64, K=6x6, 10 k-means iters, memory bank 2, fp32 train step (fwd+bwd+SGD)'),
    'tag': ('VOC12 image-tag recipe (train_spml_tag.sh: concentrations 6/8/16, weights 0.3/0.3/0.1; CAM-like '
            'blob supervision), ResNet-101 DeepLab-v2 + softmax head, %dx%d crop, 21 classes, batch %d per '
            'GPU, dim 64, K=6x6, 10 k-means iters, memory bank 2, fp32 train step (fwd+bwd+SGD)'),
    'densepose': ('DensePose point recipe, ResNet-101 PSPNet, %dx%d crop, 15 classes, batch %d per GPU, '
                  'dim 32 (+5 local), K=12x12, 10 k-means iters, no memory bank, fp32 train step '
                  '(fwd+bwd+SGD)'),
    'stress': ('stress / roofline recipe (VOC12 scribble with a %dx%d crop, 512-d embedding, 32x32 = 1024 '
               'k-means centroids per image), ResNet-101 DeepLab-v2 + softmax head, 21 classes, batch %d per '
               'GPU, 10 k-means iters, memory bank 2, fp32 train step (fwd+bwd+SGD)'),
}


def _flush_c_stdio():
  try:
    import ctypes
    ctypes.CDLL(None).fflush(None)
  except (OSError, AttributeError):
    pass


def _free_port():
  import socket
  with socket.socket() as sk:
    sk.bind(('127.0.0.1', 0))
    return sk.getsockname()[1]


def launch_ranks(args):
  """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks of this node
  (one per GPU, rendezvous on 127.0.0.1) and hand back their exit status; rank 0's JSON line goes to
  this process's stdout."""
  import subprocess
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
         '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
         os.path.abspath(__file__)] + sys.argv[1:]
  env = dict(os.environ)
  env.setdefault('OMP_NUM_THREADS', '8')           # (torchrun would otherwise pin it to 1 with a warning)
  env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC: what RCCL needs on this host driver
  return subprocess.call(cmd, env=env)


def main():
  args = parse()
  if args.gpus < 1:
    raise SystemExit('bench.py: --gpus must be >= 1')
  if not torch.cuda.is_available():
    raise SystemExit('bench.py needs an MI355X (no CPU fallback for the HIP path)')
  ndev = torch.cuda.device_count()
  if args.gpus > ndev and not args.share_gpus:
    raise SystemExit('bench.py: --gpus %d but this node exposes %d GPU(s)' % (args.gpus, ndev))
  if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
    raise SystemExit(launch_ranks(args))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local = int(os.environ.get('LOCAL_RANK', '0')) % ndev
  if world != args.gpus:
    raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d -- start as many ranks as GPUs (or let '
                     '`python bench.py --gpus N` start them)' % (args.gpus, world))
  if args.no_miopen_db:
    os.environ['MIOPEN_USER_DB_PATH'] = os.path.join('/tmp', 'spml_miopen_db_unused')
    os.environ['MIOPEN_CUSTOM_CACHE_DIR'] = os.path.join('/tmp', 'spml_miopen_cache_unused')
  torch.cuda.set_device(local)
  device = torch.device('cuda', local)
  # SPML_FORCE_DISTRIBUTED=1: take the collective code path (DDP, SyncBN, prototype exchange
  # over RCCL) in a 1-rank group too -- what a single-GPU box can check of the N > 1 path
  forced = os.environ.get('SPML_FORCE_DISTRIBUTED') == '1'
  if world > 1 or forced:
    if 'MASTER_ADDR' not in os.environ:
      os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29618', RANK='0', WORLD_SIZE='1')
    # a rendezvous / RCCL bootstrap that stalls (a rank that never started, a dead link, IPC refused) must not hang the
    # driver's clock silently: a watchdog thread names the phase and ends the process (BENCH_INIT_TIMEOUT_S, default 300)
    import datetime
    import threading
    limit = float(os.environ.get('BENCH_INIT_TIMEOUT_S', '300'))
    phase = {'name': 'process-group rendezvous (MASTER_ADDR=%s:%s)' % (os.environ.get('MASTER_ADDR'), os.environ.get('MASTER_PORT')),
             'done': False}

    def watchdog():
      t_end = time.time() + limit
      while time.time() < t_end:
        if phase['done']:
          return
        time.sleep(0.5)
      print('bench.py: rank %d of %d stalled for %.0f s in: %s -- giving up (check that all %d ranks started, '
            'HSA_ENABLE_IPC_MODE_LEGACY=0 is exported, and `rocm-smi --showtopo` lists every GPU)' % (
                rank, world, limit, phase['name'], world), file=sys.stderr, flush=True)
      os._exit(3)
    threading.Thread(target=watchdog, daemon=True).start()
    if args.dist_backend == 'nccl':
      dist.init_process_group('nccl', device_id=device, timeout=datetime.timedelta(seconds=max(limit, 60.0)))
      phase['name'] = 'first RCCL collective (communicator bootstrap over xGMI)'
      probe = torch.ones(1, device=device)
      dist.all_reduce(probe)
      torch.cuda.synchronize()
      if int(probe.item()) != dist.get_world_size():
        raise SystemExit('bench.py: first all-reduce returned %s for %d ranks' % (probe.item(), dist.get_world_size()))
    else:
      dist.init_process_group('gloo', timeout=datetime.timedelta(seconds=max(limit, 60.0)))
    phase['done'] = True
    _flush_c_stdio()          # (RCCL's version banner leaves every rank's C stdio buffer NOW, not behind rank 0's JSON line)
    if dist.get_world_size() != args.gpus:
      raise SystemExit('bench.py: process group of %d ranks for --gpus %d' % (dist.get_world_size(), args.gpus))
  torch.backends.cudnn.benchmark = bool(args.miopen_find)

  if args.no_mc_conv:
    os.environ['SPML_NO_MC_CONV'] = '1'
  import spml_amd                       # (also points MIOpen at the tuned find-db)
  from spml_amd import synth
  from spml_amd.train import (Trainer, densepose_point_config, stress_config, voc12_scribble_config,
                              voc12_tag_config)
  if args.channels_last is None:
    args.channels_last = True
  batch = args.batch or (2 if args.recipe == 'stress' else 16)
  crop = args.crop or (1025 if args.recipe == 'stress' else 513)
  make = {'voc': voc12_scribble_config, 'tag': voc12_tag_config, 'densepose': densepose_point_config,
          'stress': stress_config}[args.recipe]
  cfg = make(batch_size=batch, crop=crop)
  cfg.gpus = ','.join(str(i) for i in range(world))
  torch.manual_seed(235)
  trainer = Trainer(cfg, device, softmax_head=True, channels_last=args.channels_last,
                    recipe='densepose' if args.recipe == 'densepose' else 'voc')
  batches = [synth.make_batch(batch, crop, num_classes=cfg.dataset.num_classes,
                              seed=235 + 17 * rank + i, device=device,
                              supervision='tag' if args.recipe == 'tag' else 'scribble',
                              palette=None if args.dense_tags else (1, 3))
             for i in range(2)]
  if args.channels_last:
    for d, _ in batches:
      d['image'] = d['image'].contiguous(memory_format=torch.channels_last)

  def sync():
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()

  last = None
  for i in range(args.warmup):
    last = trainer.step(*batches[i % 2])
  sync()
  # the collective budget of one step (counted on an untimed extra step) and what ONE small collective costs on
  # this node: a multi-GPU line then explains itself (DESIGN 7 has the table of what each latency does to 8 GPUs)
  coll = None
  if dist.is_initialized() and (world > 1 or forced):
    from spml_amd import parallel
    with parallel.count_collectives(timed=device.type == 'cuda') as cc:
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      trainer.step(*batches[0])
      e1.record()
    sync()
    in_coll = cc.gpu_ms()
    step_ms = e0.elapsed_time(e1)
    coll = {'python_level_per_step': cc.total, 'by_call': dict(cc.calls),
            # where this rank's step went (one untimed step with stream events around every Python-level collective):
            # GPU time of the launch stream inside them = transfer + waiting for the slowest peer; the rest is this
            # rank's own compute at the W-rank prototype count.  A sub-linear multi-GPU line reads off here which of
            # the two grew.
            'phase_split_ms': {'step': round(step_ms, 2), 'inside_collectives': round(sum(in_coll.values()), 2),
                               'compute_and_launch': round(step_ms - sum(in_coll.values()), 2),
                               'inside_by_call': {k: round(v, 2) for k, v in in_coll.items()}},
            'note': 'SyncBatchNorm statistics (one all-gather per batch norm forward, one all-reduce per backward), '
                    'prototype exchange, accuracy counts; the bucketed gradient all-reduces of DistributedDataParallel '
                    'overlap with the backward pass and are not in this count'}
    if device.type == 'cuda' and args.dist_backend == 'nccl':
      coll['us_per_small_all_gather'] = round(parallel.small_collective_latency_us(device), 1)
    sync()
  t0 = time.perf_counter()
  marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
  marks[0].record()
  mem_trace = os.environ.get('BENCH_MEM_TRACE') == '1'      # (diagnosis: device allocations per timed step, stderr)
  for i in range(args.steps):
    last = trainer.step(*batches[i % 2])
    marks[i + 1].record()               # (a stamp on the stream, no synchronisation)
    if mem_trace:
      st = torch.cuda.memory_stats(device)
      print('step %d: device allocs %d frees %d reserved %.2f GB active peak %.2f GB' % (
          i, st['num_device_alloc'], st['num_device_free'], st['reserved_bytes.all.current'] / 2**30,
          st['active_bytes.all.peak'] / 2**30), file=sys.stderr, flush=True)
  sync()
  elapsed = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
  per_rank = [elapsed.item()]
  if world > 1:
    every = torch.zeros((world,), device=device, dtype=torch.float64)
    dist.all_gather_into_tensor(every, elapsed)
    per_rank = every.tolist()
  elapsed = max(per_rank)                   # the slowest rank's clock prices the job

  km = None
  km_total = 0.0
  if not args.no_kmeans:
    km = kmeans_stress_roofline(device) if args.recipe == 'stress' else kmeans_roofline(device)
    tot = torch.tensor([km['iters_per_s']], device=device, dtype=torch.float64)
    if world > 1:
      dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    km_total = tot.item()

  if rank == 0:
    images = batch * world * args.steps
    res = {
        'metric': 'images/sec (%dx%d) + k-means iters/sec' % (crop, crop),
        'value': round(images / elapsed, 3),
        'unit': 'images/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': round(elapsed / args.steps * 1e3, 2),
        'ms_per_step_rank_min_max': [round(min(per_rank) / args.steps * 1e3, 2),
                                     round(max(per_rank) / args.steps * 1e3, 2)],
        'ms_per_step_each_rank': [round(v / args.steps * 1e3, 2) for v in per_rank],
        'ms_each_step': [round(marks[i].elapsed_time(marks[i + 1]), 1) for i in range(args.steps)],
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f32',
        'data': 'synthetic',
        'config': {'workload': WORKLOADS[args.recipe] % (crop, crop, batch),
                   'global_batch': batch * world, 'parallelism': 'dp%d' % world,
                   'layout': 'channels_last (NHWC)' if args.channels_last else 'NCHW',
                   'deterministic_mode': bool(__import__('spml_amd._ffi', fromlist=['_ffi']).deterministic()),     # (SPML_DETERMINISTIC=1: DESIGN 11)
                   'convolutions': ('framework (MIOpen fp32) everywhere' if (args.no_mc_conv or
                                                                            not args.channels_last) else
                                    'stride-1 bottleneck units of res3/res4/res5 and the ASPP head: own split-f16 matrix-core kernels '
                                    '(fp32 in/out, 22-bit operands, 3 exact f16 products per term, fp32 accumulation '
                                    'chunked for K >= 4096; per convolution 4-8e-7 of max|out| vs fp64 against 1.5-5e-7 '
                                    'for the fp32 library, per unit <= the library: profiles/r03_conv_accuracy.md) + '
                                    'fused batch norm; rest: MIOpen fp32'),
                   'miopen': ('find mode (cudnn.benchmark)' if args.miopen_find else 'immediate mode') +
                             (', no tuned db' if args.no_miopen_db else
                              ', tuned find-db from spml_amd/miopen_db (tools/miopen_tune.py)')},
        'loss': round(float(last['loss']), 5),
        # what actually ran: ranks in the process group and the collective library (N > 1: RCCL over xGMI)
        'world_size': dist.get_world_size() if dist.is_initialized() else 1,
        'collectives': 'none (single process)' if not dist.is_initialized() else
                       ('RCCL %s' % '.'.join(str(v) for v in torch.cuda.nccl.version())) if args.dist_backend == 'nccl'
                       else 'gloo (testing: ranks share a device)',
        'devices': ndev,
    }
    if coll is not None:
      res['collectives_per_step'] = coll
    if km is not None:
      res['kmeans_iters_per_s'] = round(km_total, 1)
      res['kmeans_path'] = km['path']
      if 'iters_per_s_coherent' in km:                  # the same call on spatially coherent rows, this rank
        res['kmeans_iters_per_s_coherent'] = round(km['iters_per_s_coherent'], 1)
      res['roofline'] = km['roofline']
      if args.recipe in ('voc', 'tag') and args.channels_last and not args.no_mc_conv:
        res['roofline_backbone'] = conv_roofline(device)
    if world == 1 and not args.no_cpu_baseline:
      res['cpu_baseline'] = cpu_baseline(km, quick_kmeans_iters=2 if args.recipe == 'stress' else None)
    line = json.dumps(res)
  else:
    line = None
  if world > 1 or forced:
    dist.destroy_process_group()
  # rank 0's JSON line is the LAST thing on stdout: RCCL writes its version banner through C stdio, which is block
  # buffered on a pipe and would otherwise land behind the line when the process exits
  _flush_c_stdio()
  if line is not None:
    print(line, flush=True)


if __name__ == '__main__':
  main()
