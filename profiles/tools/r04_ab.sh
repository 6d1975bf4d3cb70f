#!/bin/bash
# A/B of library builds on the persistent C4 kernel, same box, interleaved, `reps` rounds: r04_ab.sh <reps> "<scene ...>" <lib.so> [lib.so ...]
cd $GRAFT_REPO_ROOT
reps=$1; scenes=$2; shift 2
for r in $(seq $reps); do
for sc in $scenes; do
  for l in "$@"; do
    MDS_LIB_PATH=$PWD/$l timeout -k 10 300 python3 bench.py --workload c4 --c4-scene $sc --no-cpu-baseline --no-extras --fused-rollout 50 --steps 200 --warmup 20 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sc $l us/step %.2f' % d['roofline']['us_per_step'], flush=True)"
  done
done
done
