"""Worst observed error per assertion of tests/test_conv_gpu.py over the soak files
(gpurun_out/soak/margins_*.tsv, written by tools/soak_margins.sh) -> a markdown table."""
import collections
import glob
import sys

rows = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 1e9, 0])
files = sorted(glob.glob(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/soak/margins_*.tsv'))
for f in files:
  for line in open(f):
    what, e, bound, lib = line.rstrip('\n').split('\t')
    r = rows[what]
    r[0] = max(r[0], float(e)); r[1] = float(bound); r[2] = max(r[2], float(lib)); r[3] = min(r[3], float(lib)); r[4] += 1
print('| assertion | samples | worst own error | bound | bound / worst | fp32 library (min .. max) |')
print('|---|---|---|---|---|---|')
worst = 1e9
for what, (e, b, lmax, lmin, n) in sorted(rows.items(), key=lambda kv: kv[1][1] / max(kv[1][0], 1e-30)):
  ratio = b / max(e, 1e-30)
  worst = min(worst, ratio)
  print('| %s | %d | %.2e | %.1e | %.1f | %.2e .. %.2e |' % (what, n, e, b, ratio, lmin, lmax))
print('\n%d files, smallest bound / worst-error ratio: %.2f' % (len(files), worst))
