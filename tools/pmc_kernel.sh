# PMC passes (separate runs, kernel trace only) over one command, summarised for the kernels whose
# name contains $1:   tools/pmc_kernel.sh <kernel-substring> <out-tag> <command...>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
pat=$1; tag=$2; shift 2
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_WAVES SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU"; do
  t=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o $t -- "$@" > /dev/null 2>&1
done
python - "$pat" "$out" <<'PY'
import csv, glob, sys, collections
pat, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(float)
for f in glob.glob(out + '/*counter_collection.csv'):
  seen = set()
  for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if pat not in k: continue
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    did = (f, r['Dispatch_Id'])
    if did not in seen:
      seen.add(did)
for f in glob.glob(out + '/*kernel_trace.csv')[:1]:
  for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if pat in k:
      n[k] += 1; dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k in agg:
  print(k[:120]); print('  launches %d  avg %.1f us (under the profiler)' % (n[k], dur[k] / max(n[k], 1)))
  for c, v in sorted(agg[k].items()): print('  %-28s %.4g  per launch %.4g' % (c, v, v / max(n[k], 1)))
PY
