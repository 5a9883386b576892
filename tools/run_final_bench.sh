#!/bin/bash
# The subset of tools/run_round_checks.sh that carries the judged numbers (bench line + rocprofv3 kernel statistics of
# exactly the driver's command + the steady-state step trace + the k-means profile), for a last run on the final build;
# tools/refresh_profiles.py then rewrites profiles/r06_* from gpurun_out/final/.
set -x
OUT=gpurun_out/final
mkdir -p $OUT
python bench.py 2>$OUT/bench_default.err | tail -1 > $OUT/bench_default.json; cut -c1-200 $OUT/bench_default.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_step -o step -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline 2>&1 | grep -v "^[WE]2" | tail -1 | cut -c1-200
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_driver -o drv -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep -v "^[WE]2" | tail -1 > $R/$OUT/bench_driver_cmd_under_rocprof.json; cut -c1-300 $R/$OUT/bench_driver_cmd_under_rocprof.json
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_km -o km -- python $R/tools/bench_kmeans.py --reps 5 2>&1 | grep path | tail -1 | cut -c1-300
cd $R
for cfg in "--side 513 --d 258 --k 6 --imgs 1" "--side 130 --d 66 --k 6 --imgs 16" "--side 194 --d 34 --k 12 --imgs 8" "--side 513 --d 258 --k 12 --imgs 1" "--side 258 --d 514 --k 32 --imgs 1" "--side 258 --d 514 --k 32 --imgs 4"; do python tools/bench_kmeans.py $cfg --reps 5 2>&1 | grep path | tail -1 | cut -c1-700; done > $OUT/bench_kmeans_configs.txt
python tools/probe_step_phases.py 8 2>&1 | grep -v "MIOpen\|amdgpu\|prototype feature\|set_sync_debug" | tail -12 > $OUT/step_phases.txt; cat $OUT/step_phases.txt
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_driver_style.json; cut -c1-200 $OUT/bench_driver_style.json
