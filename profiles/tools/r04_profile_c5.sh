# Round-4 rocprofv3 evidence for BASELINE config 5 (262 144 envs x 2 drones, fp16 state storage, env.step with random RPM, obs -> log ring).
# Run on the GPU box from the repo root:  bash profiles/tools/r04_profile_c5.sh [tag]
#   kernel traces (--kernel-trace --stats) of `bench.py --workload c5` step by step (one stream: full-shard launches) and with 40 steps per launch;
#   HBM-side traffic (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes, --kernel-trace only) of both.
# Outputs under gpurun_out/r04c5$tag/; the summaries that are cited get copied into profiles/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04c5$1
mkdir -p $O
COMMON="--gpus 1 --workload c5 --no-cpu-baseline --no-extras --c5-log-gb 40"
trace() { name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -- python3 bench.py $COMMON "$@" > $O/trace_$name.json 2> $O/trace_$name.err || { echo "trace $name failed"; tail -3 $O/trace_$name.err; }
  f=$(find $O/trace_$name -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_$name.csv
  echo "== $name"; head -4 $O/kernel_stats_$name.csv | cut -c1-220
}
trace c5_step --steps 2000 --warmup 200 --rollout-streams 1
trace c5_step_two_chains --steps 2000 --warmup 200
trace c5_fused40 --steps 2000 --warmup 200 --fused-rollout 40
pmc() { name=$1; c=$2; shift 2
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${name}_$c -- python3 bench.py $COMMON "$@" > $O/pmc_${name}_$c.log 2>&1 || { echo "pmc $name $c failed"; tail -3 $O/pmc_${name}_$c.log; }
}
for c in FETCH_SIZE WRITE_SIZE; do
  pmc c5_step $c --steps 400 --warmup 40 --rollout-streams 1
  pmc c5_fused40 $c --steps 400 --warmup 40 --fused-rollout 40
done
python3 - "$O" <<'PY'
import glob, csv, json, sys
O = sys.argv[1]
n = 524288
def totals(name, c, pick):
    tot, k = 0.0, 0
    for f in glob.glob(f"{O}/pmc_{name}_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and pick(r["Kernel_Name"]):
                tot += float(r["Counter_Value"]); k += 1
    return tot, k
for name, pick, steps_per_launch, alg in (("c5_step", lambda kn: "6k_stepI" in kn or "k_step<" in kn, 1, 100.0), ("c5_fused40", lambda kn: "14k_rollout_stepI" in kn or "k_rollout_step<" in kn, 40, 48.0 + 52.0 / 40)):
    rec = {"source": f"profiles/tools/r04_profile_c5.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), bench.py --workload c5 "
                     + ("--rollout-streams 1 (k_step<float, _Float16, true, ...>, one full-shard launch per env.step)" if steps_per_launch == 1 else "--fused-rollout 40 (k_rollout_step<float, _Float16, ...>, 40 env.step per launch)"),
           "unit_note": "counter unit = KiB; FETCH_SIZE x2 on gfx950 (128-B requests tallied at 64 B, MI355X_MICROARCH.md section HBM); calibrated on 16-byte-per-lane streaming reads -- "
                        "this kernel reads its fp16 state with 8-byte-per-lane loads (one 2-byte load for the last component), for which the factor is an assumption",
           "drones_per_launch_counted": n, "control_steps_per_launch": steps_per_launch}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        tot, k = totals(name, c, pick)
        rec[c + "_KiB_mean"], rec[c + "_launches"] = tot / max(k, 1), k
    rec["read_bytes_corrected"] = rec["FETCH_SIZE_KiB_mean"] * 1024 * 2
    rec["read_bytes_raw"] = rec["FETCH_SIZE_KiB_mean"] * 1024
    rec["write_bytes"] = rec["WRITE_SIZE_KiB_mean"] * 1024
    rec["traffic_bytes_per_launch"] = rec["read_bytes_corrected"] + rec["write_bytes"]
    rec["algorithmic_bytes_per_launch"] = alg * n * steps_per_launch
    rec["traffic_over_algorithmic"] = rec["traffic_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"]
    json.dump(rec, open(f"{O}/r04_pmc_traffic_{name}.json", "w"), indent=1)
    print(name, "traffic MB %.2f (read raw %.2f x2, write %.2f)" % (rec["traffic_bytes_per_launch"] / 1e6, rec["read_bytes_raw"] / 1e6, rec["write_bytes"] / 1e6), "algorithmic MB %.2f" % (rec["algorithmic_bytes_per_launch"] / 1e6), "ratio %.3f" % rec["traffic_over_algorithmic"])
PY
for n in c5_step c5_step_two_chains c5_fused40; do tail -1 $O/trace_$n.json | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('$n', r['value'], r['roofline']['us_per_step'], r['roofline']['frac'], r['roofline'].get('streams'))"; done
