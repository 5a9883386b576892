"""Copy the outputs of tools/run_round_checks.sh (gpurun_out/final/) into profiles/r06_* and regenerate the
markdown summaries that quote them (hand-written analyses -- r06_nll.md, r06_step_accuracy.md,
r06_conv_accuracy.md -- are not touched), and write profiles/kmeans_pass_pmc_traffic.json, the HBM bytes per launch
of the roofline kernel that bench.py reports as `roofline.traffic`."""
import json
import os
import shutil
import subprocess

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(R, 'gpurun_out', 'final')
P = os.path.join(R, 'profiles')


def j(name):
  return json.loads(open(os.path.join(F, name)).read())


def txt(name):
  return open(os.path.join(F, name)).read().strip()


# ---- HBM traffic counters of the k-means pass kernels (separate --pmc passes of run_round_checks.sh) ----
def _pmc(path):
  import collections, csv
  agg = collections.defaultdict(list)
  if not os.path.exists(path):
    return agg
  for r in csv.DictReader(open(path)):
    if 'kmeans_pass' in r['Kernel_Name']:
      agg[r['Kernel_Name'].replace('void spml::(anonymous namespace)::', '').split('(spml')[0]].append(float(r['Counter_Value']))
  return agg

fetch = _pmc(os.path.join(F, 'pmc_fetch', 'f_counter_collection.csv'))
write = _pmc(os.path.join(F, 'pmc_write', 'w_counter_collection.csv'))
if fetch and write:
  lines = ['# Round 6 -- HBM traffic of the k-means pass kernels (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)',
           '', 'Command: `rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python tools/bench_kmeans.py --reps 2` (and `WRITE_SIZE`);',
           '513x513x258, K = 36.  Counter values in KiB as reported; the gfx950 correction of',
           '`MI355X_MICROARCH.md` doubles it (64-B requests counted as 32 B).  Algorithmic bytes of a fused pass: 273.8 MB.', '',
           '| kernel | launches | FETCH_SIZE median (KiB) | x2 (MB) | WRITE_SIZE median (KiB) | HBM bytes per launch (MB) |', '|---|---|---|---|---|---|']
  for k in sorted(fetch):
    fv, wv = sorted(fetch[k]), sorted(write.get(k, [0.0]))
    fm, wm = fv[len(fv) // 2], wv[len(wv) // 2]
    lines.append('| `%s` | %d | %.0f | %.1f | %.0f | %.1f |' % (k, len(fv), fm, fm * 2 * 1024 / 1e6, wm,
                                                              (fm * 2 + wm) * 1024 / 1e6))
  lines += ['', '`kmeans_pass64<3, 8, 1, true>` = the fused E + M pass on pre-converted 64-pixel tiles (the roofline kernel; the value',
            '`bench.py` reports as `roofline.traffic`); `kmeans_pass64<3, 8, 1, false>` = the E-only final pass;',
            '`kmeans_pass16<3, 8, 1, false>` = the seed pass (reads fp32 X, writes the tiles).', '']
  open(os.path.join(P, 'r06_kmeans_pmc.md'), 'w').write('\n'.join(lines))
  fused = [k for k in fetch if 'kmeans_pass64<3, 8, 1, true>' in k]
  if fused:
    fv, wv = sorted(fetch[fused[0]]), sorted(write.get(fused[0], [0.0]))
    rec = {'kernel': fused[0], 'bytes_per_launch': int((fv[len(fv) // 2] * 2 + wv[len(wv) // 2]) * 1024),
           'fetch_size_kib_median': fv[len(fv) // 2], 'write_size_kib_median': wv[len(wv) // 2], 'launches': len(fv),
           'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/bench_kmeans.py --reps 2; '
                     'FETCH_SIZE x 2 (gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE, KiB; profiles/r06_kmeans_pmc.md'}
    json.dump(rec, open(os.path.join(P, 'kmeans_pass_pmc_traffic.json'), 'w'), indent=1)


import sys
if '--pmc-only' in sys.argv:
  print('PMC record refreshed')
  sys.exit(0)


shutil.copy(os.path.join(F, 'bench_default.json'), os.path.join(P, 'r06_bench_default.json'))
shutil.copy(os.path.join(F, 'prof_step', 'step_kernel_stats.csv'), os.path.join(P, 'r06_bench_kernel_stats.csv'))
shutil.copy(os.path.join(F, 'prof_driver', 'drv_kernel_stats.csv'), os.path.join(P, 'r06_bench_driver_cmd_kernel_stats.csv'))
shutil.copy(os.path.join(F, 'prof_km', 'km_kernel_stats.csv'), os.path.join(P, 'r06_kmeans_bench_kernel_stats.csv'))
shutil.copy(os.path.join(F, 'prof_km5', 'km5_kernel_stats.csv'), os.path.join(P, 'r06_kmeans_config5_kernel_stats.csv'))
d, nomc = j('bench_default.json'), j('bench_no_mc_conv.json')
tab = subprocess.run(['python', os.path.join(R, 'tools', 'summarize_trace.py'),
                      os.path.join(F, 'prof_step', 'step_kernel_trace.csv'), '--steps', '3', '--top', '45'],
                     capture_output=True, text=True).stdout
open(os.path.join(P, 'r06_train_step_steady_state.md'), 'w').write('''# Round 6 -- steady-state kernel time per training step (1x MI355X)

Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline`
(batch 16, 513x513, ResNet-101 DeepLab-v2, fp32 in / out, channels-last; stride-1 bottleneck units of res3 / res4 / res5
and the ASPP head (forward as one 1x1 convolution with 36 x 64 columns + `conv_tap_gather`, data gradient as one 36-tap
launch, weight gradients as one launch of `conv_wgrad<256, 256, 4, true, true>` on tiles of four taps x 64 channels) on
the matrix-core convolutions of `csrc/conv.hip` with fused batch norm, the rest (stem, res2, the stride-2 unit) on MIOpen with the tuned find-db of `spml_amd/miopen_db`).  Default `python bench.py` of the same build without
the profiler: %.1f images/s, %.1f ms/step (`r06_bench_default.json`); `python bench.py --no-mc-conv` (library
convolutions everywhere): %.1f images/s, %.1f ms.  Aggregated with `tools/summarize_trace.py` over the last 3 timed
steps (the whole-run `--stats` file is `r06_bench_kernel_stats.csv`); `tools/run_round_checks.sh` +
`tools/refresh_profiles.py` regenerate everything.  Phases of a step from stream events and the host
synchronisations of one step (`tools/probe_step_phases.py`):

```
%s
```

''' % (d['value'], d['ms_per_step'], nomc['value'], nomc['ms_per_step'], txt('step_phases.txt')) + tab +
  '\nThe narrow ASPP head stand-alone (`tools/bench_conv.py --narrow`):\n\n```\n' +
  ''.join(l for l in open(os.path.join(F, 'bench_conv_narrow.txt')) if l.startswith('ASPP')) + '```\n')

# k-means / K1 / other recipes
km = [json.loads(l) for l in open(os.path.join(F, 'bench_kmeans_configs.txt')) if l.startswith('{')]
names = ['config R: 513^2 x 258, K=36', 'configs 2/3: 16 x 130^2 x 66, K=36', 'config 4: 8 x 194^2 x 34, K=144',
         'config R, 12x12: 513^2 x 258, K=144', 'config 5: 258^2 x 514, K=1024', 'config 5 x4 images']
t = ('| shape | path | us / iteration | iterations / s | fused pass us (mean) | HBM frac of the pass | f16 MFMA TFLOP/s of the pass | '
     'seed / final pass us |\n|---|---|---|---|---|---|---|---|\n')
for n, r in zip(names, km):
  t += '| %s | `%s` | %.1f | %.0f | %s | %s | %s | %s |\n' % (
      n, r['path'], r['us_per_iter'], r['iters_per_s'], r.get('fused_pass_us_mean', '-'), r.get('frac_8TB', '-'),
      r.get('mfma_f16_tflops', '-'),
      ('%s / %s' % (r['seed_pass_us'], r['final_pass_us'])) if 'seed_pass_us' in r else '-')
k1 = [json.loads(l) for l in open(os.path.join(F, 'bench_k1.txt')) if l.startswith('{')]
t1 = '| shape | layout of the map | fwd us | fwd GB/s (frac of 8 TB/s) | bwd us | bwd GB/s (frac) |\n|---|---|---|---|---|---|\n'
for r in k1:
  t1 += '| %s | %s | %.1f | %.0f (%.2f) | %.1f | %.0f (%.2f) |\n' % ('x'.join(str(v) for v in r['shape']), r.get('layout', 'nchw'), r['fwd_us'],
                                                              r['fwd_GBps'], r['fwd_frac_8TB'], r['bwd_us'],
                                                              r['bwd_GBps'], r['bwd_frac_8TB'])
rec = ''
for name in ('tag', 'stress', 'densepose'):
  b = j('bench_%s.json' % name)
  extra = ''
  if 'roofline' in b:
    extra = '; roofline %s %.1f %s = %.3f of peak; k-means %.0f it/s (%s)' % (
        b['roofline']['bound'], b['roofline']['achieved'], b['roofline']['unit'], b['roofline']['frac'],
        b.get('kmeans_iters_per_s', 0), b.get('kmeans_path', ''))
  rec += '* `%s`: **%.2f images/s** (%.1f ms/step), %s%s\n' % (name, b['value'], b['ms_per_step'],
                                                               b['config']['workload'][:150], extra)
open(os.path.join(P, 'r06_other_configs.md'), 'w').write('''# Round 6 -- k-means on every BASELINE shape, K1, label algebra, other recipes (1x MI355X, `tools/run_round_checks.sh`)

## k-means (`tools/bench_kmeans.py`, 10 iterations behind 50 ms of untimed calls -- settled shader clock, `r06_kmeans_clock.md`; pass durations = per-workgroup device clocks of one run)

%s
Binding roofline per row: config R at K = 36 -- HBM (fused pass 0.7 of 8 TB/s by device stamps; whole iteration incl. the seed / final
passes and the two small kernels: see `us / iteration`); the training shape (16 images of 130^2 x 66) -- per-tile fixed
costs at D = 66 (0.46 of HBM; 0.42 behind the three-call warm-up of the earlier runs, 0.29 in round 4); `pass16k` at K = 144 / D = 34 and the `bigk` rows -- matrix-core work on padded tiles
(TFLOP/s column against the 2 500 TFLOP/s dense f16 peak; counters in `r03_mfma_counters.md`); config R with the 12 x 12
grid (K = 144): `mfma_f16x2_v4k`, an assign and an accumulate kernel per iteration (`fused pass us` = their sum; phase
breakdown in `r05_kmeans_k144.md`; round 4: 244.5 us per iteration on `mfma_f16x2_bigk`).

Why the k-means ITERATION rate stays at ~0.50 of the HBM roofline (round 6: the finalize kernels' serialised loads fixed,
72 -> 67-70 us per iteration; `r06_kmeans_iteration.md` has what else was measured): round 3 built the decomposition VERDICT r2 asked for
(hi-half screened E-step + exact incremental M-step, `csrc/kmeans_inc.hip`): parity-green, 13.2 k instead of 12.7 k
iterations / s on noise-like rows and SLOWER on spatially coherent ones (`r03_kmeans_screened.md`); round 4 removed it
(1 100 opt-in lines with spills, VERDICT r3 weak 4); round 5 rebuilt the fused pass itself (`kmeans_pass64`).  Rates that
priced the decomposition
(513^2 x 258, K = 36, labels after iteration i against i - 1, exact top-2 margin of every pixel):

| iteration | 1 | 2 | 3 | 4 | 5 | 6 | 7 | 8 | 9 | 10 |
|---|---|---|---|---|---|---|---|---|---|---|
| pixels that change cluster | 61 %% | 14.5 %% | 9.9 %% | 8.1 %% | 6.5 %% | 5.1 %% | 4.0 %% | 3.2 %% | 2.6 %% | 2.2 %% |
| top-2 margin < 1e-3 (worst-case bound of the dropped l terms, 2 x 2^-11) | 12.4 %% | 6.7 %% | 6.2 %% | 5.3 %% | 4.8 %% | 4.4 %% | 4.2 %% | 4.0 %% | 4.0 %% | 3.8 %% |

A hipGraph replay of the whole (fused-pass) call measured 0 %% (83.5 vs 82.7 us per iteration).

## K1 (`tools/bench_k1.py`)

%s
NCHW backward rewritten in round 4 (all loads of a tile up front, one reduction round, g1 rows in registers; round 2:
127.6 us = 0.27 and 1 210 us = 0.11).  Channels-last rows: the backbone of the benchmarked configuration runs NHWC, so
the embedding map arrives with contiguous pixel rows; `k1_nhwc_kernel` streams them (LPR lanes per row, float4 per
lane, shuffles inside the row group, no LDS, no transposition) -- and the NHWC -> NCHW copy in front of K1 and the
NCHW -> NHWC copy behind its backward are gone from the step.

## Label algebra (`tools/bench_relabel.py`): `spml_relabel_unique_i64` against `torch.unique(return_inverse=True)`

```
%s
```

Since round 4 the distinct keys are sorted in 2048-key LDS bitonic tiles and a key's rank is the sum of its lower bounds
in the tiles (binary searches), instead of the O(U^2) count of round 3: 1 273 -> 315 us at U = 139 k distinct keys (the
8-GPU segment count), unchanged at the step's own sizes.  What the kernel buys is the host: 23 -> 3 synchronisations per
training step.

## Other recipes (`bench.py --recipe ...`, 6 timed steps after 3 warm-up steps; densepose: 3 after 2)

%s
N2 / N3 (`tools/bench_inference.py`): %s

%s
''' % (t, t1, txt('bench_relabel.txt'), rec, txt('bench_inference_n2.json'), txt('bench_inference_n3.json')))

# ---- the 60-launch block of the roofline kernel, five times ----
rf = d['roofline']
rows = rf.get('blocks_us_mhz_kcycles') or []
lines = ['# Round 6 -- the roofline kernel under the clock: five blocks of 60 launches inside the default `python bench.py`', '',
         '`roofline.achieved` = algorithmic bytes / the MEDIAN launch duration of the five blocks (HIP events on the launch stream;',
         'every block runs straight behind SIX untimed bursts of 60 launches = 20 ms of the same kernel: see below);',
         'the shader clock of a block is read by a one-wave probe kernel right behind it (`spml_clock_probe`: `s_memtime` against the',
         '100-MHz `s_memrealtime`).', '',
         '| block | us per launch | shader MHz | duration x clock (k cycles) | HBM fraction |', '|---|---|---|---|---|']
for i, r in enumerate(rows):
  lines.append('| %d | %.2f | %.0f | %.1f | %.3f |' % (i + 1, r[0], r[1], r[2], rf['algorithmic_bytes'] / (r[0] * 1e-6) / 8e12))
mm = rf.get('us_per_launch_min_median_max')
lines += ['', 'median %.2f us per launch -> %.1f GB/s = **%.3f** of 8 TB/s; min / median / max of the five blocks: %s us.' % (
              rf.get('us_per_launch', float('nan')), rf['achieved'], rf['frac'], mm),
          '', 'duration x clock is constant to ~2 %: a launch takes 110-117 k shader cycles in whatever clock state the firmware',
          'has the part in.  What decides the clock state (measured in round 5, `BENCH_KM_WARM_BURSTS`, three boxes):',
          '', '| untimed launches in front of a timed block | the five blocks of one run, us per launch (shader MHz) |', '|---|---|',
          '| one burst of 60 (3.3 ms) | 52.2 (2157), 59.4 (1900), 53.4 (2127), 55.3 (2080), 59.8 (1926); another run: 51.9, 62.9, 60.2, 61.2, 63.6 (1765-2100) |',
          '| six bursts (20 ms) | 49.4, 50.8, 50.1, 49.4, 50.8; another run: 50.1, 50.7, 50.2, 48.8, 50.8 |',
          '| twelve bursts (40 ms) | 49.7, 50.1, 50.5, 49.1, 48.4 |',
          '', 'Behind an idle gap or other kernels (the blocks sit between whole k-means runs with their host synchronisations) the',
          'firmware runs the part at 1.8-2.1 GHz and needs ~10 ms of uninterrupted launches to raise it to the 2.2-2.35 GHz this',
          'HBM-bound kernel sustains; with one burst in front, the SAME blocks of every run were the slow ones (the second',
          'and the fifth).  That ramp is the "identical launches take 52-75 us inside one process" of VERDICT r4, not a property',
          'of the kernel: settled, the launch period is 48.4-51.1 us = 0.67-0.71 of 8 TB/s on every block of every run.  The',
          'whole-call figure (`kmeans_iters_per_s`) gets the same treatment: 60 untimed calls (45 ms) in front of its A B A B',
          'blocks instead of 10 (10: 13.4-13.5 k / 13.8 k iterations/s on noise / coherent rows -- the first block paid the ramp;',
          '40: 13.8 / 13.8 k; 80: 14.0 / 14.0 k).  What the cycles are spent on: DESIGN 5e.  The same kernel under',
          '`rocprofv3 --kernel-trace --stats` of the driver command: `r06_bench_driver_cmd_kernel_stats.csv` (all launches of',
          'the process, ramps included).', '']
open(os.path.join(P, 'r06_kmeans_clock.md'), 'w').write('\n'.join(lines))

print('profiles refreshed: %.1f images/s, %.1f ms/step' % (d['value'], d['ms_per_step']))
