#!/bin/bash
# power / clock / phase probe of the matrix-core convolutions over the SPML_CONV_EXP profiling builds (GPU box):
#   0 the shipped kernel, 48 + clock and phase stamps, 52 without MFMAs, 51 operands from the zero page, 55 both
out=${1:-gpurun_out/conv_power.txt}
mkdir -p "$(dirname "$out")"
: > "$out"
for e in 0 48 52 51 55; do
  touch spml_amd/csrc/conv.hip
  SPML_CONV_EXP=$e python -m spml_amd._build > /dev/null 2>&1
  SPML_CONV_EXP=$e timeout 600 python tools/probe_conv_power.py 2>&1 | grep -v amdgpu.ids >> "$out"
done
touch spml_amd/csrc/conv.hip
python -m spml_amd._build > /dev/null 2>&1
cat "$out"
