#!/bin/bash
# the persistent C4 kernel by control steps per launch (1 = a launch per step, like the step-by-step API)
for T in 1 2 5 10 50; do
  timeout -k 10 300 python bench.py --workload c4 --c4-scene ${1:-under} --no-cpu-baseline --no-extras --fused-rollout $T --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${1:-under} steps per launch $T: us/step %.2f' % d['roofline']['us_per_step'])"
done
