# SQ counters of the three C4 kernels (k_cbf_nominal, k_cbf_filter_gi, k_lowlevel_step): four --pmc passes with --kernel-trace
# only (8 SQ slots per pass), per-launch means -> gpurun_out/r02_pmc_c4_summary.txt.  bash profiles/tools/pmc_c4.sh [scene]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
scene=${1:-level}
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT" "SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_c4_${scene}_$i -- python3 bench.py --workload c4 --c4-scene $scene --steps 200 --warmup 20 --rollout-streams 1 --no-cpu-baseline --no-extras > gpurun_out/pmc_c4_${scene}_$i.log 2>&1 || echo "pass $i failed"
done
python3 profiles/tools/pmc_sum.py gpurun_out/pmc_c4_${scene}_1 gpurun_out/pmc_c4_${scene}_2 gpurun_out/pmc_c4_${scene}_3 gpurun_out/pmc_c4_${scene}_4 > gpurun_out/r02_pmc_c4_${scene}_summary.txt
cat gpurun_out/r02_pmc_c4_${scene}_summary.txt
