SPML_TRACE=1 python -m spml_amd._build --force > /dev/null 2>&1
for e in 1 17; do
  touch spml_amd/csrc/kmeans64.hip
  SPML_TRACE=1 SPML_P64_EXP=$e python -m spml_amd._build > /dev/null 2>&1
  echo "=== EXP $e"
  SPML_KM_TRACE=1 timeout 300 python tools/bench_kmeans.py --reps 1 --iters 3 2>&1 | grep "wave0\|path" | tail -2 | cut -c1-420
done
