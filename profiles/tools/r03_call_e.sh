#!/bin/bash
# CBF test file, then an A/B of library builds: r03_call_e.sh "<scenes>" <lib.so ...>
set -o pipefail
mkdir -p gpurun_out/r3e
timeout -k 10 900 python -m pytest tests/test_gpu_cbf.py -m gpu -x -q > gpurun_out/r3e/tests.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -4 gpurun_out/r3e/tests.log
[ $rc -ne 0 ] && { tail -60 gpurun_out/r3e/tests.log; exit 1; }
bash profiles/tools/r03_ab.sh "$@"
