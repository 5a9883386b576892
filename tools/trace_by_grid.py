"""Group a rocprofv3 kernel trace by (kernel, grid size): calls, mean and total duration."""
import collections
import csv
import sys

acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
  name = r['Kernel_Name']
  if len(sys.argv) > 2 and sys.argv[2] not in name:
    continue
  acc[(name[:70], r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Grid_Size_Y', ''))].append(
      int(r['End_Timestamp']) - int(r['Start_Timestamp']))
rows = sorted(acc.items(), key=lambda kv: -sum(kv[1]))
for (name, gx, gy), v in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
  print('%-70s grid %8s x %5s  calls %4d  mean %8.1f us  total %8.2f ms' % (name, gx, gy, len(v), sum(v) / len(v) / 1e3,
                                                                            sum(v) / 1e6))
