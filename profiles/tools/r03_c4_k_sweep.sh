# C4 `under` scene, persistent kernel: microseconds per control step by the number of control steps per launch (bash profiles/tools/r03_c4_k_sweep.sh)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s9
for k in 10 25 50 100 200; do
  python bench.py --gpus 1 --workload c4 --c4-scene under --steps 200 --warmup 200 --fused-rollout $k --no-cpu-baseline --no-extras > gpurun_out/s9/k$k.json 2> gpurun_out/s9/k$k.err
  python - $k <<'PY'
import sys, json
k = sys.argv[1]
d = json.loads(open(f"gpurun_out/s9/k{k}.json").read().strip().splitlines()[-1])
print("steps per launch", k, "us per control step %.2f" % (d["ms_per_step"] * 1e3), "G drone-steps/s %.2f" % (d["value"] / 1e9), "frac %.3f" % d["roofline"]["frac"])
PY
done
