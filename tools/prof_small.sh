cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_small
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_small -o s -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kmeans > /dev/null 2>&1
python - <<'PY'
import csv,os,glob
R=os.environ['GRAFT_REPO_ROOT']
f=glob.glob(R+'/gpurun_out/prof_small/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if any(k in n for k in ('bn_merge','conv_wgrad_reduce','bn_partial','bn_apply','bn_bwd_apply','nll_')):
        print('%-90s calls %6s avg %9.1f us'%(n[:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
