mkdir -p gpurun_out/r5b
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r5b/pytest_gpu.txt; cat gpurun_out/r5b/pytest_gpu.txt
timeout 900 python bench.py --steps 10 --warmup 4 2>gpurun_out/r5b/bench.err | tail -1 > gpurun_out/r5b/bench_default.json; cat gpurun_out/r5b/bench_default.json | cut -c1-1500
timeout 600 python bench.py --steps 10 --warmup 4 --dense-tags --no-cpu-baseline --no-kmeans 2>/dev/null | tail -1 > gpurun_out/r5b/bench_dense_tags.json; cut -c1-400 gpurun_out/r5b/bench_dense_tags.json
