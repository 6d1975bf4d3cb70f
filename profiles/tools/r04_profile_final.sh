# kernel traces of the default bench command and of a strong-scaling shard with the round's final library.  bash profiles/tools/r04_profile_final.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04pf
mkdir -p $O
trace() { name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -- python3 bench.py "$@" --no-cpu-baseline --no-extras > $O/trace_$name.json 2> $O/trace_$name.err || { echo "trace $name failed"; tail -3 $O/trace_$name.err; }
  f=$(find $O/trace_$name -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_$name.csv
  echo "== $name"; head -2 $O/kernel_stats_$name.csv | cut -c1-60,225-330
  tail -1 $O/trace_$name.json | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('$name', '%.4g' % r['value'], 'us/step %.2f' % r['roofline']['us_per_step'], 'frac %.3f' % r['roofline']['frac'], r['config'].get('launch_form'))"
}
trace default_form2 --gpus 1
trace shard_of_8_h2 --gpus 1 --steps 2000 --warmup 200 --shard-of 8
