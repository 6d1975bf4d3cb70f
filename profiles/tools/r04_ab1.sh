#!/bin/bash
# A/B of part-1 library builds on the C3 rollout, same box, interleaved: r04_ab1.sh <reps> "<bench args>" <lib.so> [lib.so ...]
cd $GRAFT_REPO_ROOT
reps=$1; bargs=$2; shift 2
for r in $(seq $reps); do
  for l in "$@"; do
    MDS_LIB_PATH=$PWD/$l timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras $bargs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l form %s us/step %.2f' % (d['config'].get('launch_form'), d['roofline']['us_per_step']), flush=True)"
  done
done
