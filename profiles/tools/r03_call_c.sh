#!/bin/bash
# persistent CBF rollout: parity tests, timings on three scenes (T = 50), per-stage stamps
set -o pipefail
mkdir -p gpurun_out/r3c
timeout -k 10 600 python -m pytest tests/test_gpu_cbf.py -m gpu -x -q -k "persistent" > gpurun_out/r3c/tests.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -5 gpurun_out/r3c/tests.log
[ $rc -ne 0 ] && { tail -40 gpurun_out/r3c/tests.log; exit 1; }
for sc in under level far; do
  for T in ${TS:-50}; do
    timeout -k 10 300 python bench.py --workload c4 --c4-scene $sc --no-cpu-baseline --no-extras --fused-rollout $T --steps 200 --warmup 20 > gpurun_out/r3c/c4_${sc}_T$T.json 2> gpurun_out/r3c/c4_${sc}_T$T.err || { echo "c4 $sc T=$T failed"; tail -5 gpurun_out/r3c/c4_${sc}_T$T.err; }
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3c/c4_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "us/step %.2f" % d["roofline"]["us_per_step"], "G %.2f" % (d["value"] / 1e9), "frac %.3f" % d["roofline"]["frac"], "sane", d["state_sane"], "fb", d.get("cbf_fallback_frac_last_step"))
    except Exception as e:
        print(f, "ERR", e)
PY
bash profiles/tools/r03_stamps.sh 50
