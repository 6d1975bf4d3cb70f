#!/bin/bash
# full GPU suite + the driver-shaped bench
mkdir -p gpurun_out/r3f
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r3f/gpu_tests.log 2>&1; echo "pytest rc $?"
tail -15 gpurun_out/r3f/gpu_tests.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3f/driver.json 2> gpurun_out/r3f/driver.err; echo "driver bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3f/driver.json").read().strip().splitlines()[-1])
print("value %.3g frac %.3f ms/step %.5f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"]))
for k in ("rk4", "f64", "f32c"):
    print(k, {kk: (round(v, 4) if isinstance(v, float) else v) for kk, v in d[k].items() if kk in ("us_per_step", "frac", "bytes_per_drone_step", "state_sane", "error")})
c4 = d["configs_4_c4"]
for k in ("feasible_active", "survey_8d", "feasible_active_fused"):
    v = c4.get(k, {})
    print(k, v.get("us_per_step"), v.get("value"), v.get("roofline", {}).get("frac"), v.get("window", {}).get("fallback_frac"), v.get("error"))
print("c5", d["configs_5_c5"].get("us_per_step"), d["configs_5_c5"].get("roofline", {}).get("frac"))
PY
