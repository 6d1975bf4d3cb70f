# two-chain rollouts: LDS bytes per half-shard workgroup (= workgroups per CU) against call length.  bash profiles/tools/split_lds.sh
cd $GRAFT_REPO_ROOT
for lds in 20480 23296 27264 32768 40960 54528; do
  echo "== MDS_TUNE_SPLIT_LDS=$lds"
  MDS_TUNE_SPLIT_LDS=$lds python3 profiles/tools/short_calls.py 15 20,48,200,2000 | grep " 2 2 "
done
