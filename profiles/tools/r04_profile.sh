# Round-4 rocprofv3 evidence (run on the GPU box from the repo root:  bash profiles/tools/r04_profile.sh):
#   kernel traces (--kernel-trace --stats) of the driver's command (two chains, and one stream), of a strong-scaling shard (1/8 of config 3:
#   the whole-rollout kernel the auto launch form picks), of the persistent C4 kernel on `under` / `level` (fp32) and `under` (float64);
#   HBM-side traffic (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes, --kernel-trace only) of one C4 control step of the persistent
#   kernel (the new row layout), both scenes;  SQ counters of the persistent kernel (profiles/tools/pmc_c4_fused.sh).
# Outputs under gpurun_out/r04p/; the summaries that are cited get copied into profiles/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04p
mkdir -p $O
trace() { name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -- python3 bench.py "$@" --no-cpu-baseline --no-extras > $O/trace_$name.json 2> $O/trace_$name.err || { echo "trace $name failed"; tail -3 $O/trace_$name.err; }
  f=$(find $O/trace_$name -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_$name.csv
  echo "== $name"; head -3 $O/kernel_stats_$name.csv | cut -c1-200
}
trace driver --gpus 1 --steps 20 --warmup 5
trace driver_one_stream --gpus 1 --steps 20 --warmup 5 --rollout-streams 1
trace shard_of_8 --gpus 1 --steps 2000 --warmup 200 --shard-of 8
trace c4_under_fused --gpus 1 --workload c4 --c4-scene under --steps 200 --warmup 20 --fused-rollout 50
trace c4_level_fused --gpus 1 --workload c4 --c4-scene level --steps 200 --warmup 20 --fused-rollout 50
trace c4_under_fused_f64 --gpus 1 --workload c4 --c4-scene under --steps 200 --warmup 20 --fused-rollout 50 --dtype float64
pmc() { name=$1; c=$2; shift 2
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${name}_$c -- python3 bench.py "$@" --no-cpu-baseline --no-extras > $O/pmc_${name}_$c.log 2>&1 || { echo "pmc $name $c failed"; tail -3 $O/pmc_${name}_$c.log; }
}
for c in FETCH_SIZE WRITE_SIZE; do
  for sc in under level; do
    pmc c4_${sc}_fused $c --workload c4 --c4-scene $sc --steps 200 --warmup 20 --fused-rollout 50
  done
done
python3 - <<'PY'
import glob, csv, json, collections
O = "gpurun_out/r04p"
n4, steps = 262144, 220
for sc in ("under", "level"):
    rec = {"source": f"profiles/tools/r04_profile.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), bench.py --workload c4 --c4-scene {sc} --steps 200 --warmup 20 --fused-rollout 50 (k_cbf_rollout<float, 0, false, 8, false>: per-drone bounds, every step's observation into a 50-slot ring)",
           "unit_note": "counter unit = KiB; FETCH_SIZE x2 as for 16-byte-per-lane streaming reads (MI355X_MICROARCH.md section HBM); the kernel also issues 4-byte and scalar loads, for which the factor is uncalibrated: an upper bound on the reads",
           "drones_per_step_counted": n4, "control_steps_counted": steps}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        tot, k = 0.0, 0
        for f in glob.glob(f"{O}/pmc_c4_{sc}_fused_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c and "k_cbf_rollout" in r["Kernel_Name"]:
                    tot += float(r["Counter_Value"]); k += 1
        rec[c + "_KiB_total"], rec[c + "_launches"] = tot, k
    rec["read_bytes_per_step_corrected"] = rec["FETCH_SIZE_KiB_total"] * 1024 * 2 / steps
    rec["write_bytes_per_step"] = rec["WRITE_SIZE_KiB_total"] * 1024 / steps
    rec["traffic_bytes_per_step"] = rec["read_bytes_per_step_corrected"] + rec["write_bytes_per_step"]
    rec["algorithmic_bytes_per_step"] = 280 * n4
    json.dump(rec, open(f"{O}/r04_pmc_traffic_c4_{sc}_fused.json", "w"), indent=1)
    print("c4", sc, "traffic/step MB %.1f" % (rec["traffic_bytes_per_step"] / 1e6), "algorithmic MB %.1f" % (rec["algorithmic_bytes_per_step"] / 1e6), "ratio %.2f" % (rec["traffic_bytes_per_step"] / rec["algorithmic_bytes_per_step"]))
PY
for n in driver driver_one_stream shard_of_8 c4_under_fused c4_level_fused c4_under_fused_f64; do tail -1 $O/trace_$n.json | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('$n', '%.4g' % r['value'], 'us/step %.2f' % r['roofline']['us_per_step'], 'frac %.3f' % r['roofline']['frac'], r['roofline'].get('streams'), r['config'].get('launch_form'))"; done
MDS_ROUND=r04 bash profiles/tools/pmc_c4_fused.sh under 50 > $O/pmc_sq_under.log 2>&1; tail -3 $O/pmc_sq_under.log
MDS_ROUND=r04 bash profiles/tools/pmc_c4_fused.sh level 50 > $O/pmc_sq_level.log 2>&1; tail -3 $O/pmc_sq_level.log
