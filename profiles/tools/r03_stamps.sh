#!/bin/bash
# per-stage shader-clock stamps of k_cbf_rollout on the three C4 scenes (MDS_TUNE_ROLL_STAMPS=1: a tuning aid, synchronises the stream)
mkdir -p gpurun_out/r3c
for sc in far under level; do
  echo "== scene $sc"
  MDS_TUNE_ROLL_STAMPS=1 timeout -k 10 300 python bench.py --workload c4 --c4-scene $sc --no-cpu-baseline --no-extras --fused-rollout ${1:-50} --steps 200 --warmup 50 2>&1 >/dev/null | grep "roll stamps" | tail -11
done
