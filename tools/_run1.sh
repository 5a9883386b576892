mkdir -p gpurun_out/r5a
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_edge_cases_gpu.py tests/test_full_size_properties_gpu.py -m gpu -q -x -k "kmeans" 2>&1 | tail -3 | tee gpurun_out/r5a/pytest_kmeans.txt
for f in 0 512; do timeout 300 python tools/bench_kmeans.py --reps 10 --flags $f 2>&1 | grep path | tail -1 | tee -a gpurun_out/r5a/bench_km.txt; done
for f in 0 512; do timeout 300 python tools/bench_kmeans.py --side 130 --d 66 --k 6 --imgs 16 --reps 10 --flags $f 2>&1 | grep path | tail -1 | tee -a gpurun_out/r5a/bench_km.txt; done
SPML_TRACE=1 python -m spml_amd._build --force > /dev/null 2>&1
SPML_KM_TRACE=1 timeout 300 python tools/bench_kmeans.py --reps 1 --iters 3 2>&1 | grep "wave" | tail -4 | tee gpurun_out/r5a/trace64.txt
