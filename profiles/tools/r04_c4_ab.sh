# round 4: C4 persistent kernel A/B -- CBF GPU tests, then bench.py --workload c4 --fused-rollout 50 on the three scenes (HIP events, us per control step)
# usage: bash profiles/tools/r04_c4_ab.sh <tag> [pytest -k expression]
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c4_$1
mkdir -p $O
if [ -n "$2" ]; then
  timeout -k 10 900 python3 -m pytest tests/test_gpu_cbf.py -x -q -m gpu -k "$2" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -15 $O/pytest.log
fi
for sc in under level far; do
  timeout -k 10 300 python3 bench.py --workload c4 --c4-scene $sc --steps 200 --warmup 20 --fused-rollout 50 --no-cpu-baseline --no-extras > $O/bench_$sc.json 2> $O/bench_$sc.err || { echo "bench $sc failed"; tail -5 $O/bench_$sc.err; }
  python3 -c "
import json,sys
r=json.loads(open('$O/bench_$sc.json').read().strip().splitlines()[-1])
print('$sc', 'us/step %.2f' % r['roofline']['us_per_step'], 'G %.2f' % (r['value']/1e9), 'frac %.3f' % r['roofline']['frac'], 'fallback_last %.3f' % r['cbf_fallback_frac_last_step'], 'iters_last', r['cbf_iterations_last_step'].get('mean'), 'sane', r['state_sane'])"
done
