#!/usr/bin/env python3
"""Kernels of a rocprofv3 kernel trace in launch order: name, duration, gap to the previous one.
  python tools/trace_kernels.py TRACE.csv [substring] [--last N]"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
  for r in csv.DictReader(f):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
pat = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith('--') else ''
last = int(sys.argv[sys.argv.index('--last') + 1]) if '--last' in sys.argv else 60
sel = [(s, e, n) for s, e, n in rows if pat in n][-last:]
prev = None
for s, e, n in sel:
  short = n.replace('void spml::(anonymous namespace)::', '').replace('spml::(anonymous namespace)::', '').replace('_ZN4spml12_GLOBAL__N_1', '').split('(')[0][:60]
  print('%-60s %8.1f us   gap %7.1f us' % (short, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
  prev = e
