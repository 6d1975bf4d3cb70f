#!/bin/bash
# occupancy experiment on the persistent C4 kernel: unused dynamic LDS forces 1 workgroup per CU (8 waves) instead of 2
for sc in ${1:-under}; do
  for x in 0 65536; do
    MDS_TUNE_ROLL_EXTRA_LDS=$x timeout -k 10 300 python bench.py --workload c4 --c4-scene $sc --no-cpu-baseline --no-extras --fused-rollout 50 --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sc extra_lds=$x us/step %.2f' % d['roofline']['us_per_step'])"
  done
done
