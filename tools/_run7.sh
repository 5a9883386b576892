timeout 900 python -m pytest tests/test_train_step_gpu.py -m gpu -q -x -s -k "collective_code_path or headline" 2>&1 | grep -E "collectives per step|passed|failed|Error|assert" | head
SPML_FORCE_DISTRIBUTED=1 timeout 600 python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-kmeans 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print(r['ms_per_step'], r.get('collectives_per_step'))"
timeout 600 python tools/emulate_world.py 1 8 2>&1 | grep "^W = "
