# round 4, GPU call 1: new launch-form tests, the strong-scaling shard sweep, the 20-step call overhead probe, C5 profile evidence
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "launch_form or two_stream or rollout_equals_stepwise or baseline_configs_2_and_3 or captured" > gpurun_out/r4a/pytest.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r4a/pytest.log
python3 profiles/tools/r04_shard_sweep.py gpurun_out/r4a/r04_shard_sweep.json > gpurun_out/r4a/sweep.log 2>&1; echo "sweep rc $?"; tail -25 gpurun_out/r4a/sweep.log
python3 profiles/tools/r04_call_overhead.py 20 15 > gpurun_out/r4a/call_overhead.log 2>&1; echo "overhead rc $?"; cat gpurun_out/r4a/call_overhead.log
bash profiles/tools/r04_profile_c5.sh > gpurun_out/r4a/c5_profile.log 2>&1; echo "c5 rc $?"; tail -30 gpurun_out/r4a/c5_profile.log
