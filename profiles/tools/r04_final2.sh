cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f2
mkdir -p $O
t0=$(date +%s); timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc $? in $(( $(date +%s) - t0 )) s"; tail -2 $O/pytest_gpu.log
t0=$(date +%s); timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $? in $(( $(date +%s) - t0 )) s"; tail -3 $O/smoke.log
t0=$(date +%s); timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "driver bench rc $? in $(( $(date +%s) - t0 )) s"
t0=$(date +%s); timeout -k 10 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc $? in $(( $(date +%s) - t0 )) s"
python3 - <<'PY'
import json
for n in ("driver", "default"):
    r = json.loads(open(f"gpurun_out/r04f2/bench_{n}.json").read().strip().splitlines()[-1])
    print(n, "value %.4g" % r["value"], "ms/step %.5f" % r["ms_per_step"], "frac %.3f" % r["roofline"]["frac"])
    c = r.get("configs_4_c4", {})
    for kk, v in c.items():
        if isinstance(v, dict) and "us_per_step" in v: print("   c4", kk, "us/step %.2f" % v["us_per_step"], "frac %.3f" % v["roofline"]["frac"])
    print("   c2", json.dumps(r.get("configs_1_c2"))[:400])
    print("   c5 %.2f %.3f" % (r["configs_5_c5"]["us_per_step"], r["configs_5_c5"]["roofline"]["frac"]))
PY
