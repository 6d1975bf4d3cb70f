#!/bin/bash
# A/B of library builds on the persistent C4 kernel: r03_ab.sh "<scene ...>" <lib.so> [lib.so ...]
scenes=$1; shift
for sc in $scenes; do
  for l in "$@"; do
    MDS_LIB_PATH=$PWD/$l timeout -k 10 300 python bench.py --workload c4 --c4-scene $sc --no-cpu-baseline --no-extras --fused-rollout 50 --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sc $l us/step %.2f' % d['roofline']['us_per_step'])"
  done
done
