for e in 0 1 2 3 4 7; do
  touch spml_amd/csrc/conv.hip
  SPML_CONV_EXP=$e python -m spml_amd._build > /dev/null 2>&1
  echo "=== EXP $e"
  timeout 300 python tools/bench_conv.py --no-lib --reps 5 2>&1 | grep "^fwd" | cut -c1-75
done
