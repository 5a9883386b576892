cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_nll
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_WAVES SQ_ACTIVE_INST_LDS SQ_WAIT_ANY"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_nll -o $tag -- python $R/tools/bench_nll.py 270400 70000 > /dev/null 2>&1
done
ls $R/gpurun_out/pmc_nll
