# priority of the handle's internal chain stream (MDS_SPLIT_STREAM_PRIORITY: 0 default priority, 1 / high, 2 / low) against the three bench shapes, 3 runs each
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for pr in 0 1 2; do
  export MDS_SPLIT_STREAM_PRIORITY=$pr
  a=$(python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('%.2f' % (r['ms_per_step']*1e3))")
  b=$(python3 bench.py --steps 2000 --warmup 200 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('%.2f' % r['roofline']['us_per_step'])")
  c=$(python3 bench.py --workload c4 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('%.2f' % (r['ms_per_step']*1e3))")
  d=$(python3 bench.py --workload c5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('%.2f' % r['roofline']['us_per_step'])")
  echo "prio $pr: c3 20-step wall $a | c3 2000-step $b | c4 $c | c5 $d"
done
done
