"""Per (kernel, grid size) averages of the counters of rocprofv3 --pmc passes:  pmc_by_grid.py <dir> <kernel-substring>"""
import collections
import csv
import glob
import sys

d, pat = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
  for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if pat not in k:
      continue
    key = (k.split('(')[0][-60:], int(r['Grid_Size']) // max(int(r['Workgroup_Size']), 1))
    agg[key][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[key][r['Counter_Name']].add((f, r['Dispatch_Id']))
for key in sorted(agg):
  print('%s  workgroups %d' % key)
  for c, v in sorted(agg[key].items()):
    print('   %-28s per launch %.5g   (%d launches)' % (c, v / len(cnt[key][c]), len(cnt[key][c])))
