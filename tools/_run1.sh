mkdir -p gpurun_out/r5a
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_edge_cases_gpu.py tests/test_full_size_properties_gpu.py -m gpu -q -x -k "kmeans" 2>&1 | tail -2
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); f = r['roofline']
print({k: f[k] for k in ('frac', 'us_per_launch', 'us_per_launch_min_median_max', 'us_per_launch_device_stamps', 'us_per_iteration', 'blocks_us_mhz_kcycles')}, r['kmeans_iters_per_s'])"
