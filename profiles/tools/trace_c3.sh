# rocprofv3 --kernel-trace --stats of the driver's own command, of the default (long, two-chain) run, of the same on one stream and
# of the HBM-resident shard -> gpurun_out/r02_trace_*.  Run on the GPU box from the repo root:  bash profiles/tools/trace_c3.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { # name, bench args
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_trace_$name -- python3 bench.py "$@" --no-cpu-baseline --no-extras > gpurun_out/r02_trace_$name.json 2> gpurun_out/r02_trace_$name.err || exit 1
}
run driver --gpus 1 --steps 20 --warmup 5
run c3 --gpus 1 --steps 2000 --warmup 200
run c3_one_stream --gpus 1 --steps 2000 --warmup 200 --rollout-streams 1
run c3big --gpus 1 --workload c3big --steps 50 --warmup 5
run c4 --gpus 1 --workload c4 --steps 200 --warmup 20
for d in driver c3 c3_one_stream c3big c4; do f=$(find gpurun_out/r02_trace_$d -name '*kernel_stats.csv' | head -1); echo "== $d"; head -5 "$f"; tail -1 gpurun_out/r02_trace_$d.json | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['us_per_step'], r['roofline']['frac'], r['roofline']['streams'])"; done
