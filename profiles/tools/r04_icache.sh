# instruction-cache counters of k_cbf_rollout for a set of library builds (one --pmc pass each, --kernel-trace only):
# bash profiles/tools/r04_icache.sh <scene> <lib.so> [lib.so ...]   -> per-launch means of SQC_ICACHE_REQ / HITS / MISSES, SQ_WAIT_INST_ANY, SQ_WAVE_CYCLES
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
scene=$1; shift
for l in "$@"; do
  tag=$(basename $l .so)
  MDS_LIB_PATH=$PWD/$l rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH --kernel-trace --output-format csv -d gpurun_out/ic_${scene}_$tag -- python3 bench.py --workload c4 --c4-scene $scene --steps 200 --warmup 50 --fused-rollout 50 --no-cpu-baseline --no-extras > gpurun_out/ic_${scene}_$tag.log 2>&1 || { echo "pass $tag failed"; tail -3 gpurun_out/ic_${scene}_$tag.log; }
  echo "== $tag"; python3 profiles/tools/pmc_sum.py gpurun_out/ic_${scene}_$tag | grep cbf_rollout
done
