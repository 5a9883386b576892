#!/bin/bash
# Round checks on an MI355X box: GPU tests, smoke, the default bench line, rocprofv3 kernel
# traces / PMC passes of the same commands, the other BASELINE configs and the micro-benchmarks.
# Everything lands under gpurun_out/final/; the summaries to be judged are copied to profiles/.
set -x
OUT=gpurun_out/final
mkdir -p $OUT
# hardware probes are built artefacts (git-ignored): build the ones that are missing
for p in tools/hw_probes/global_atomics.hip; do [ -x ${p%.hip}.bin ] || hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -Wno-inline-asm -mllvm -amdgpu-mfma-vgpr-form $p -o ${p%.hip}.bin; done
python -m pytest tests -m gpu -q -rf 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
python __graft_entry__.py --smoke 2>&1 | tail -1
python bench.py 2>$OUT/bench_default.err | tail -1 > $OUT/bench_default.json; cat $OUT/bench_default.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_step -o step -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline 2>&1 | grep -v "^[WE]2" | tail -1
# exactly the driver's command: its kernel stats are profiles/r06_bench_driver_cmd_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_driver -o drv -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep -v "^[WE]2" | tail -1 > $R/$OUT/bench_driver_cmd_under_rocprof.json; cut -c1-300 $R/$OUT/bench_driver_cmd_under_rocprof.json
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_km -o km -- python $R/tools/bench_kmeans.py --reps 5 2>&1 | grep path | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_km5 -o km5 -- python $R/tools/bench_kmeans.py --side 258 --d 514 --k 32 --reps 5 2>&1 | grep path | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_km144 -o km144 -- python $R/tools/bench_kmeans.py --k 12 --reps 5 2>&1 | grep path | tail -1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$OUT/pmc_fetch -o f -- python $R/tools/bench_kmeans.py --reps 2 2>&1 | grep path | tail -1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$OUT/pmc_write -o w -- python $R/tools/bench_kmeans.py --reps 2 2>&1 | grep path | tail -1
cd $R
for cfg in "--side 513 --d 258 --k 6 --imgs 1" "--side 130 --d 66 --k 6 --imgs 16" "--side 194 --d 34 --k 12 --imgs 8" "--side 513 --d 258 --k 12 --imgs 1" "--side 258 --d 514 --k 32 --imgs 1" "--side 258 --d 514 --k 32 --imgs 4"; do python tools/bench_kmeans.py $cfg --reps 5 2>&1 | grep path | tail -1 | cut -c1-700; done > $OUT/bench_kmeans_configs.txt; cat $OUT/bench_kmeans_configs.txt
python tools/bench_nll.py > $OUT/bench_nll.txt 2>&1; cat $OUT/bench_nll.txt
python tools/bench_nll.py 66564 3000 9000 --d 514 > $OUT/bench_nll_d514.txt 2>&1; cat $OUT/bench_nll_d514.txt
python tools/bench_k1.py > $OUT/bench_k1.txt 2>&1; cat $OUT/bench_k1.txt
for r in tag stress; do python bench.py --recipe $r --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_$r.json; cut -c1-400 $OUT/bench_$r.json; done
python bench.py --recipe densepose --batch 8 --crop 769 --steps 3 --warmup 2 --no-cpu-baseline --no-kmeans 2>/dev/null | tail -1 > $OUT/bench_densepose.json; cut -c1-300 $OUT/bench_densepose.json
python tools/bench_inference.py 2>&1 | tail -1 > $OUT/bench_inference_n2.json; cat $OUT/bench_inference_n2.json
python tools/bench_inference.py --walk 64 64 2>&1 | tail -1 > $OUT/bench_inference_n3.json; cat $OUT/bench_inference_n3.json
python tools/bench_conv.py > $OUT/bench_conv.txt 2>&1; grep fwd $OUT/bench_conv.txt
python tools/bench_conv.py --narrow > $OUT/bench_conv_narrow.txt 2>&1; grep "fwd\|ASPP" $OUT/bench_conv_narrow.txt
python tools/bench_upsample_ce.py 2>&1 | grep -v amdgpu > $OUT/bench_upsample_ce.txt; cat $OUT/bench_upsample_ce.txt
python tools/probe_step_phases.py 8 2>&1 | grep -v "MIOpen\|amdgpu\|prototype feature\|set_sync_debug" | tail -12 > $OUT/step_phases.txt; cat $OUT/step_phases.txt
python tools/bench_relabel.py 2>&1 | grep "^P " > $OUT/bench_relabel.txt; cat $OUT/bench_relabel.txt
for l in "" "--nhwc" "--config headline"; do python tools/probe_step_accuracy.py $l 2>&1 | grep -v "MIOpen\|Warn\|amdgpu\|detach"; done > $OUT/probe_step_accuracy.txt; cat $OUT/probe_step_accuracy.txt
python tools/probe_mc_unit.py 2>&1 | grep -v "^MIOpen\|amdgpu" > $OUT/probe_mc_unit.txt; cat $OUT/probe_mc_unit.txt
python bench.py --no-mc-conv --steps 4 --warmup 2 --no-cpu-baseline --no-kmeans 2>/dev/null | tail -1 > $OUT/bench_no_mc_conv.json; cut -c1-220 $OUT/bench_no_mc_conv.json
# a rank's compute at the W-rank prototype count (profiles/r06_scaling_emulation.md is written from this by hand)
python tools/emulate_world.py 1 2 4 8 2>&1 | grep "^W = " > $OUT/emulate_world.txt; cat $OUT/emulate_world.txt
./tools/hw_probes/global_atomics.bin > $OUT/global_atomics.txt 2>&1; cat $OUT/global_atomics.txt
python tools/probe_determinism.py > $OUT/determinism_off.txt 2>&1; python tools/probe_determinism.py --deterministic > $OUT/determinism_on.txt 2>&1; tail -3 $OUT/determinism_on.txt
# what the deterministic mode costs: the default bench step with it switched on
SPML_DETERMINISTIC=1 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kmeans 2>/dev/null | tail -1 > $OUT/bench_deterministic.json; cut -c1-200 $OUT/bench_deterministic.json
python tools/bench_widened_rows.py 2>/dev/null | tail -1 > $OUT/bench_widened_rows.json; cat $OUT/bench_widened_rows.json
