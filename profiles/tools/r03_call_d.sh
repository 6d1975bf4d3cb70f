#!/bin/bash
# CBF test file (all kernels use gi_solve) + persistent-kernel timings + stamps
set -o pipefail
mkdir -p gpurun_out/r3d
timeout -k 10 900 python -m pytest tests/test_gpu_cbf.py -m gpu -x -q > gpurun_out/r3d/tests.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -4 gpurun_out/r3d/tests.log
[ $rc -ne 0 ] && { tail -40 gpurun_out/r3d/tests.log; exit 1; }
for sc in under level far; do
  for T in 0 50; do
    timeout -k 10 300 python bench.py --workload c4 --c4-scene $sc --no-cpu-baseline --no-extras --fused-rollout $T --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sc T=$T us/step %.2f G %.2f frac %.3f sane %s' % (d['roofline']['us_per_step'], d['value']/1e9, d['roofline']['frac'], d['state_sane']))"
  done
done
bash profiles/tools/r03_stamps.sh 50 | grep -v "max.*wait\|xx"
