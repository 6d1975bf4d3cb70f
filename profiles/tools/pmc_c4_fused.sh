# SQ counters of the persistent CBF rollout kernel (k_cbf_rollout): four --pmc passes with --kernel-trace only (8 SQ slots per pass),
# per-launch means -> gpurun_out/r03_pmc_c4_<scene>_fused_summary.txt.  bash profiles/tools/pmc_c4_fused.sh [scene] [steps per launch]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
scene=${1:-under}
T=${2:-50}
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT" "SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_c4f_${scene}_$i -- python3 bench.py --workload c4 --c4-scene $scene --steps 200 --warmup 50 --fused-rollout $T --no-cpu-baseline --no-extras > gpurun_out/pmc_c4f_${scene}_$i.log 2>&1 || echo "pass $i failed"
done
python3 profiles/tools/pmc_sum.py gpurun_out/pmc_c4f_${scene}_1 gpurun_out/pmc_c4f_${scene}_2 gpurun_out/pmc_c4f_${scene}_3 gpurun_out/pmc_c4f_${scene}_4 > gpurun_out/${MDS_ROUND:-r04}_pmc_c4_${scene}_fused_summary.txt
cat gpurun_out/${MDS_ROUND:-r04}_pmc_c4_${scene}_fused_summary.txt
