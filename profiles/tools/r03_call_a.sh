#!/bin/bash
# round 3, call a: scene validation at full size + regression of the GPU suite + the default bench
set -o pipefail
mkdir -p gpurun_out/r3a
for sc in under level far; do
  timeout -k 10 300 python bench.py --workload c4 --c4-scene $sc --no-cpu-baseline > gpurun_out/r3a/c4_$sc.json 2> gpurun_out/r3a/c4_$sc.err || echo "c4 $sc failed"
done
for z in -2.0 -3.0; do
  timeout -k 10 300 python bench.py --workload c4 --c4-scene under --c4-z $z --no-cpu-baseline > gpurun_out/r3a/c4_under_z$z.json 2> gpurun_out/r3a/c4_under_z$z.err || echo "c4 z $z failed"
done
echo "scenes done"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/gpu_tests.log 2>&1; echo "pytest rc $?"
tail -3 gpurun_out/r3a/gpu_tests.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3a/driver.json 2> gpurun_out/r3a/driver.err; echo "driver bench rc $?"
