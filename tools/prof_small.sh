# steady-state kernel list of the training step: framework kernels by total time (tuning aid)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_small
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_small -o s -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kmeans > /dev/null 2>&1
python - <<'PY'
import csv,os,glob,collections
R=os.environ['GRAFT_REPO_ROOT']
f=glob.glob(R+'/gpurun_out/prof_small/*kernel_trace.csv')[0]
rows=[]
for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name']))
rows.sort()
marks=[i for i,r in enumerate(rows) if 'max_pool_forward_n' in r[2]]
a,b=marks[-3],marks[-1]
tot=collections.Counter(); cnt=collections.Counter()
for s,e,n in rows[a:b]:
    if 'spml' in n: continue
    tot[n[:110]]+=(e-s)/2e3; cnt[n[:110]]+=0.5
print('step ms', (rows[b][0]-rows[a][0])/2e6)
for k,v in tot.most_common(28): print('%8.1f us  x%5.1f  %s'%(v,cnt[k],k))
print('framework total ms', sum(tot.values())/1e3)
PY
