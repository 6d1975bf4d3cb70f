# Is the C4 bench window (20 warm-up steps, 200 timed) measured on a GPU that has not reached its clocks?  The `far` scene costs the same at
# every step (no env ever iterates): us per control step by the number of warm-up steps, persistent kernel and step by step.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s10
for w in 20 200 1000; do
  for mode in "--fused-rollout 50" ""; do
    tag=$(echo "$mode" | tr -d ' -')
    python bench.py --gpus 1 --workload c4 --c4-scene far --steps 200 --warmup $w $mode --no-cpu-baseline --no-extras > gpurun_out/s10/w${w}_${tag}.json 2> gpurun_out/s10/w${w}_${tag}.err
    python - $w "$tag" <<'PY'
import sys, json
w, tag = sys.argv[1], sys.argv[2]
d = json.loads(open(f"gpurun_out/s10/w{w}_{tag}.json").read().strip().splitlines()[-1])
print("far scene, warm-up steps", w, tag or "stepwise", "us per control step %.2f" % (d["ms_per_step"] * 1e3), "frac %.3f" % d["roofline"]["frac"])
PY
  done
done
