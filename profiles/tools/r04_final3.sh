# round 4: the driver's own sequence on one box -- pytest -m gpu, smoke(), the driver's bench command, the default bench line, a strong-scaling shard
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f3
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest_gpu.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -4 $O/smoke.log
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "driver bench rc $?"
timeout -k 10 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc $?"
timeout -k 10 300 python3 bench.py --scaling strong --shard-of 8 --no-cpu-baseline --no-extras > $O/bench_shard8.json 2> $O/bench_shard8.err; echo "shard bench rc $?"
python3 - <<'PY'
import json
for n in ("driver", "default", "shard8"):
    try:
        r = json.loads(open(f"gpurun_out/r04f3/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, "value %.4g" % r["value"], "ms/step %.5f" % r["ms_per_step"], "frac %.3f" % r["roofline"]["frac"], "streams", r["roofline"].get("streams"), "form", r["config"].get("launch_form"), "scaling", r["scaling"])
        for k in ("configs_4_c4",):
            if k in r:
                for kk, v in r[k].items():
                    if isinstance(v, dict) and "us_per_step" in v:
                        print("   c4", kk, "us/step %.2f" % v["us_per_step"], "frac %.3f" % v["roofline"]["frac"])
        if "configs_5_c5" in r: print("   c5 us/step %.2f frac %.3f traffic %s" % (r["configs_5_c5"]["us_per_step"], r["configs_5_c5"]["roofline"]["frac"], r["configs_5_c5"]["roofline"].get("traffic_over_algorithmic")))
        if "fused_rollout" in r: print("   fused_rollout us/step %.2f" % r["fused_rollout"]["us_per_step"])
        if r.get("cpu_baseline"): print("   cpu_baseline %.4g on %s cores" % (r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"]))
    except Exception as e:
        print(n, "ERR", e)
PY
