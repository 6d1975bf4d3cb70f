#!/bin/bash
# the bench lines of round 3: driver's command, defaults, C4 (three scenes, both forms), C5, C2
mkdir -p gpurun_out/r3b_all
b() { name=$1; shift; t0=$(date +%s.%N); timeout -k 10 900 python bench.py "$@" > gpurun_out/r3b_all/$name.json 2> gpurun_out/r3b_all/$name.err; rc=$?; echo "$name rc $rc $(python -c "import time; print('%.1f s wall' % (time.time() - $t0))")"; }
b driver --gpus 1 --steps 20 --warmup 5
b c3
b c4_under --workload c4
b c4_under_fused --workload c4 --fused-rollout 50
b c4_level --workload c4 --c4-scene level
b c4_level_fused --workload c4 --c4-scene level --fused-rollout 50
b c4_far_fused --workload c4 --c4-scene far --fused-rollout 50
b c5 --workload c5
b c2 --workload c2
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3b_all/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "value %.3g us/step %.2f frac %.3f sane %s" % (d["value"], d["roofline"]["us_per_step"], d["roofline"]["frac"], d["state_sane"]), "cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "ERR", e)
PY
