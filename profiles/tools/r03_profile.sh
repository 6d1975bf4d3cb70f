# Round-3 rocprofv3 evidence (run on the GPU box from the repo root:  bash profiles/tools/r03_profile.sh):
#   kernel traces (--kernel-trace --stats) of the driver's command (two chains, and one stream), of the C4 step-by-step rollout and of
#   the persistent C4 rollout kernel on the 'under' and 'level' scenes;
#   HBM-side traffic (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes, --kernel-trace only) of the C3 kernel and of one C4 control
#   step (all its launches), both forms, both scenes.
# Outputs under gpurun_out/r03p/; the summaries that are cited get copied into profiles/ by hand.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03p
mkdir -p $O
trace() { # name, bench args
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -- python3 bench.py "$@" --no-cpu-baseline --no-extras > $O/trace_$name.json 2> $O/trace_$name.err || { echo "trace $name failed"; tail -3 $O/trace_$name.err; }
  f=$(find $O/trace_$name -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_$name.csv
  echo "== $name"; head -4 $O/kernel_stats_$name.csv | cut -c1-200
}
trace driver --gpus 1 --steps 20 --warmup 5
trace driver_one_stream --gpus 1 --steps 20 --warmup 5 --rollout-streams 1
trace c4_under --gpus 1 --workload c4 --c4-scene under --steps 200 --warmup 20
trace c4_under_fused --gpus 1 --workload c4 --c4-scene under --steps 200 --warmup 20 --fused-rollout 50
trace c4_level_fused --gpus 1 --workload c4 --c4-scene level --steps 200 --warmup 20 --fused-rollout 50
pmc() { # name, counter, bench args
  name=$1; c=$2; shift 2
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${name}_$c -- python3 bench.py "$@" --no-cpu-baseline --no-extras > $O/pmc_${name}_$c.log 2>&1 || { echo "pmc $name $c failed"; tail -3 $O/pmc_${name}_$c.log; }
}
for c in FETCH_SIZE WRITE_SIZE; do
  pmc c3 $c --workload c3 --steps 200 --warmup 20 --rollout-streams 1
  for sc in under level; do
    pmc c4_$sc $c --workload c4 --c4-scene $sc --steps 200 --warmup 20 --rollout-streams 1
    pmc c4_${sc}_fused $c --workload c4 --c4-scene $sc --steps 200 --warmup 20 --fused-rollout 50
  done
done
python3 - <<'PY'
import glob, csv, json, collections
O = "gpurun_out/r03p"
def totals(name, c, pick):
    tot, k, per = 0.0, 0, collections.Counter()
    for f in glob.glob(f"{O}/pmc_{name}_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and pick(r["Kernel_Name"]):
                tot += float(r["Counter_Value"]); k += 1; per[r["Kernel_Name"].split("(")[0][:60]] += 1
    return tot, k, dict(per)
# C3: per full-shard launch
n = 524288
out = {"source": "profiles/tools/r03_profile.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), bench.py --workload c3 --rollout-streams 1 --no-extras; kernel k_step_geometric<float, float, true, false, false, false, false>, one full-shard launch per control step",
       "unit_note": "counter unit = KiB; FETCH_SIZE x2 on gfx950 (128-B requests tallied at 64 B, MI355X_MICROARCH.md section HBM)", "drones_per_launch_counted": n}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    tot, k, _ = totals("c3", c, lambda kn: "k_step_geometric" in kn)
    out[c + "_KiB_mean"], out[c + "_launches"] = tot / max(k, 1), k
out["read_bytes_corrected"] = out["FETCH_SIZE_KiB_mean"] * 1024 * 2
out["write_bytes"] = out["WRITE_SIZE_KiB_mean"] * 1024
out["traffic_bytes_per_launch"] = out["read_bytes_corrected"] + out["write_bytes"]
out["algorithmic_bytes_per_launch"] = 212 * n
json.dump(out, open(f"{O}/r03_pmc_traffic_c3.json", "w"), indent=1); print("c3", out["traffic_bytes_per_launch"], out["algorithmic_bytes_per_launch"])
# C4: per control step, all launches of the CBF step
n4, steps = 262144, 220
for sc in ("under", "level"):
    for form in ("", "_fused"):
        pick = (lambda kn: "k_cbf_rollout" in kn) if form else (lambda kn: "k_cbf_filter" in kn or "k_lowlevel_step" in kn or "k_cbf_order" in kn or "k_cbf_step" in kn)
        rec = {"source": f"profiles/tools/r03_profile.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), bench.py --workload c4 --c4-scene {sc} --steps 200 --warmup 20 " + ("--fused-rollout 50 (k_cbf_rollout, every step's observation into a 50-slot ring)" if form else "--rollout-streams 1 (k_cbf_filter_gi + k_lowlevel_step [+ k_cbf_order every 8th step], one stream)"),
               "unit_note": "counter unit = KiB; FETCH_SIZE x2 as for 16-byte-per-lane streaming reads (MI355X_MICROARCH.md section HBM); these kernels also issue 4-byte and scalar loads, for which the factor is uncalibrated: an upper bound on the reads",
               "drones_per_step_counted": n4, "control_steps_counted": steps}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            tot, k, per = totals(f"c4_{sc}{form}", c, pick)
            rec[c + "_KiB_total"], rec[c + "_launches"], rec[c + "_launches_by_kernel"] = tot, k, per
        rec["read_bytes_per_step_corrected"] = rec["FETCH_SIZE_KiB_total"] * 1024 * 2 / steps
        rec["write_bytes_per_step"] = rec["WRITE_SIZE_KiB_total"] * 1024 / steps
        rec["traffic_bytes_per_step"] = rec["read_bytes_per_step_corrected"] + rec["write_bytes_per_step"]
        rec["algorithmic_bytes_per_step"] = 280 * n4
        json.dump(rec, open(f"{O}/r03_pmc_traffic_c4_{sc}{form}.json", "w"), indent=1)
        print("c4", sc, form or "stepwise", "traffic/step MB %.1f" % (rec["traffic_bytes_per_step"] / 1e6), "algorithmic MB %.1f" % (rec["algorithmic_bytes_per_step"] / 1e6), "ratio %.2f" % (rec["traffic_bytes_per_step"] / rec["algorithmic_bytes_per_step"]))
PY
for n in driver driver_one_stream c4_under c4_under_fused c4_level_fused; do tail -1 $O/trace_$n.json | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('$n', r['value'], r['roofline']['us_per_step'], r['roofline']['frac'], r['roofline'].get('streams'))"; done
