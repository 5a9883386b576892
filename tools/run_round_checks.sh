set -x
mkdir -p gpurun_out/final
python -m pytest tests -m gpu -q -rf 2>&1 | tail -6
python __graft_entry__.py --smoke 2>&1 | tail -1
python bench.py 2>&1 | grep -v "^[WE]2" | tail -1 > gpurun_out/final/bench_default.json; cat gpurun_out/final/bench_default.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/final/prof_step -o step -- python /root/repo/bench.py --steps 3 --warmup 2 --no-cpu-baseline 2>&1 | grep -v "^[WE]2" | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/final/prof_km -o km -- python /root/repo/tools/bench_kmeans.py --reps 5 2>&1 | tail -1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /root/repo/gpurun_out/final/pmc_fetch -o f -- python /root/repo/tools/bench_kmeans.py --reps 2 2>&1 | tail -1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /root/repo/gpurun_out/final/pmc_write -o w -- python /root/repo/tools/bench_kmeans.py --reps 2 2>&1 | tail -1
cd /root/repo
python tools/bench_inference.py 2>&1 | tail -1 > gpurun_out/final/bench_inference_n2.json; cat gpurun_out/final/bench_inference_n2.json
python tools/bench_inference.py --walk 64 64 2>&1 | tail -1 > gpurun_out/final/bench_inference_n3.json; cat gpurun_out/final/bench_inference_n3.json
for cfg in "--side 130 --d 66 --k 6 --imgs 16" "--side 194 --d 34 --k 12 --imgs 8" "--side 258 --d 514 --k 32 --imgs 1"; do python tools/bench_kmeans.py $cfg --reps 5 2>&1 | tail -1 | cut -c1-330; done > gpurun_out/final/bench_kmeans_other_configs.txt; cat gpurun_out/final/bench_kmeans_other_configs.txt
