#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes tools/pmc_kernel.sh left in a directory for the kernels whose (demangled) name
contains a pattern:  pmc_summary.py <dir> <pattern>"""
import csv, glob, sys, collections
out, pat = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(float)
for f in glob.glob(out + '/*counter_collection.csv'):
  for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if pat not in k: continue
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for f in sorted(glob.glob(out + '/*kernel_trace.csv'))[:1]:
  for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if pat in k:
      n[k] += 1; dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k in agg:
  print(k[:120]); print('  launches %d  avg %.1f us (under the profiler)' % (n[k], dur[k] / max(n[k], 1)))
  for c, v in sorted(agg[k].items()): print('  %-28s per launch %.5g' % (c, v / max(n[k], 1)))
