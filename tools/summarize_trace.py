#!/usr/bin/env python3
"""Steady-state per-step kernel table from a rocprofv3 kernel trace of bench.py.

  rocprofv3 --kernel-trace --stats --output-format csv -d DIR -o step -- \
      python bench.py --steps 3 --warmup 2 --no-cpu-baseline
  python tools/summarize_trace.py DIR/step_kernel_trace.csv --steps 3 > profiles/rNN_....md

The whole-run --stats file is dominated by MIOpen's first-iteration solver search,
so the table is built from the trace instead: training steps are delimited by the
max-pool forward kernel (launched exactly once per step, at the start of the
backbone) and only the last `--steps` of them are aggregated."""
import argparse
import collections
import csv


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('trace')
  ap.add_argument('--steps', type=int, default=3)
  ap.add_argument('--marker', default='maxpool3x3s2_nhwc,max_pool_forward_n',
                  help="substrings (comma separated) of the kernel launched once per step: the stem's max pool "
                       "(own channels-last kernel; the framework's max_pool_forward_nhwc / _nchw on other paths)")
  ap.add_argument('--top', type=int, default=45)
  ap.add_argument('--tail-marker', default='kmeans_pass16<3, 8',
                  help='first kernel of what bench.py runs after the timed steps')
  args = ap.parse_args()

  rows = []
  with open(args.trace) as f:
    for r in csv.DictReader(f):
      rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
  rows.sort()
  marks = [i for i, r in enumerate(rows) if any(m in r[2] for m in args.marker.split(','))]
  if len(marks) < args.steps + 1:
    raise SystemExit('need at least %d marker kernels, found %d' % (args.steps + 1, len(marks)))
  # the last step ends where the k-means roofline runs begin (or, without them, one
  # mean step after its marker)
  first = marks[-args.steps]
  per_step = (rows[marks[-1]][0] - rows[marks[-args.steps]][0]) / max(args.steps - 1, 1)
  end_time = rows[marks[-1]][0] + per_step
  tail = [r[0] for r in rows[marks[-1]:] if args.tail_marker in r[2]]
  if tail:
    end_time = min(end_time, tail[0])
  sel = [r for r in rows[first:] if r[0] < end_time]
  wall = (end_time - rows[first][0]) / args.steps
  agg = collections.OrderedDict()
  for s, e, n in sel:
    a = agg.setdefault(n, [0, 0])
    a[0] += e - s
    a[1] += 1
  busy = sum(a[0] for a in agg.values()) / args.steps
  print('GPU busy: %.1f ms of %.1f ms per step (%d kernels/step).\n' %
        (busy / 1e6, wall / 1e6, len(sel) // args.steps))
  print('| kernel | ms/step | calls/step | avg us |')
  print('|---|---|---|---|')
  for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:args.top]:
    print('| `%s` | %.2f | %.1f | %.1f |' % (n[:100], t / args.steps / 1e6, c / args.steps, t / c / 1e3))
  groups = [('own convolutions, forward + data gradient (conv_gemm)', ('conv_gemm',)),
            ('own weight gradients (conv_wgrad + reduce)', ('conv_wgrad',)),
            ('own batch norm (bn_*), split conversions (hl8_*, weightset_*)', ('bn_', 'hl8_', 'weightset_', 'absmax')),
            ('library convolutions (MIOpen igemm / CK / transposes)', ('igemm', 'ck16', '_ZN2ck', 'batched_transpose', 'Im2d2Col', 'Cijk', 'naive_conv', 'SubTensor')),
            ('contrastive losses (nll_*), top-k, relabel', ('nll_', 'topk', 'relabel')),
            ('k-means, prototypes, K1, cross-entropy head (kmeans_*, segsum, k1_*, uce_*)', ('kmeans', 'segsum', 'k1_', 'uce_', 'normalize')),
            ]
  left = dict((n, t) for n, (t, c) in agg.items())
  print('\n| group | ms/step |')
  print('|---|---|')
  for title, keys in groups:
    tot = sum(t for n, t in list(left.items()) if any(k in n for k in keys))
    for n in [n for n in left if any(k in n for k in keys)]:
      del left[n]
    print('| %s | %.1f |' % (title, tot / args.steps / 1e6))
  print('| framework element-wise / reduction / optimizer / copy kernels | %.1f |' % (sum(left.values()) / args.steps / 1e6))
  ours = sum(t for n, (t, c) in agg.items() if 'spml' in n) / args.steps
  print('\nlibspml_hip.so kernels: %.2f ms/step (%.1f %% of GPU busy time).' %
        (ours / 1e6, 100.0 * ours / busy))
  km = [(e - s) for s, e, n in rows if 'kmeans_pass16' in n and 'false>' not in n and s >= end_time]
  if km:
    km.sort()
    # per run: 1 accumulate-only seed pass, iterations-1 fused passes, 1 assign-only pass
    fused = [t for t in km if t > 0.85 * km[len(km) // 2]]
    print('\nk-means roofline configuration (same run): %d pass launches after the timed steps, '
          'fused E+M launches avg %.1f us (min %.1f, max %.1f).' %
          (len(km), sum(fused) / len(fused) / 1e3, fused[0] / 1e3, fused[-1] / 1e3))


if __name__ == '__main__':
  main()
