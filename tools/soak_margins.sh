#!/bin/bash
# Soak of the -m gpu suite on one box: the whole suite in the driver's order with -x,
# then tests/test_conv_gpu.py under three shifted seeds with every (error, bound, fp32-library error)
# triple appended to gpurun_out/soak/margins_<host>.tsv; tools/summarize_margins.py turns the files of
# several boxes into profiles/r06_test_margins.md.
OUT=gpurun_out/soak
mkdir -p $OUT
TAG=${1:-$(hostname)-$(date +%H%M%S)}
python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $OUT/suite_$TAG.txt
for s in 0 1000 2000 3000; do
  SPML_TEST_SEED=$s SPML_TEST_MARGINS=$OUT/margins_$TAG.tsv python -m pytest tests/test_conv_gpu.py -q -m gpu 2>&1 | tail -3 | tee -a $OUT/conv_seeds_$TAG.txt
done
