# rocprofv3 --kernel-trace --stats of the driver's own command and of the HBM-resident shard -> gpurun_out/r02_trace_*.
# Run on the GPU box from the repo root:  bash profiles/tools/trace_c3.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_trace_driver -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r02_trace_driver.json 2> gpurun_out/r02_trace_driver.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_trace_c3 -- python3 bench.py --gpus 1 --steps 2000 --warmup 200 --no-cpu-baseline --no-extras > gpurun_out/r02_trace_c3.json 2> gpurun_out/r02_trace_c3.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_trace_c3_one_stream -- python3 bench.py --gpus 1 --steps 2000 --warmup 200 --rollout-streams 1 --no-cpu-baseline --no-extras > gpurun_out/r02_trace_c3_one_stream.json 2> gpurun_out/r02_trace_c3_one_stream.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_trace_c3big -- python3 bench.py --gpus 1 --workload c3big --steps 50 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r02_trace_c3big.json 2> gpurun_out/r02_trace_c3big.err || exit 1
for d in driver c3 c3_one_stream c3big; do f=$(find gpurun_out/r02_trace_$d -name '*kernel_stats.csv' | head -1); echo "== $d"; head -4 "$f"; cat gpurun_out/r02_trace_$d.json | python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['roofline']['us_per_step'], r['roofline']['frac'], r['roofline']['streams'])"; done
