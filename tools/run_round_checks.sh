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
