cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT" "SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_c4_$i -- python3 bench.py --workload c4 --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_c4_$i.log 2>&1 || echo "pass $i failed"
done
python3 profiles/tools/pmc_sum.py gpurun_out/pmc_c4_1 gpurun_out/pmc_c4_2 gpurun_out/pmc_c4_3 gpurun_out/pmc_c4_4 > gpurun_out/pmc_c4_summary.txt
cat gpurun_out/pmc_c4_summary.txt
