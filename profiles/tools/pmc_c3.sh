# HBM traffic of the C3 hot kernel: two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only, for the
# C3 shard (cache-resident, one stream: full-shard launches) and the 4 M-drone shard (HBM-resident), then the per-launch
# means -> gpurun_out/r02_pmc_traffic_{c3,c3big}.json.  Run on the GPU box from the repo root:  bash profiles/tools/pmc_c3.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for w in c3 c3big; do
  if [ $w = c3 ]; then args="--workload c3 --steps 200 --warmup 20 --rollout-streams 1"; else args="--workload c3big --steps 30 --warmup 5 --rollout-streams 1"; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${w}_$c -- python3 bench.py $args --no-cpu-baseline --no-extras > gpurun_out/pmc_${w}_$c.log 2>&1 || { echo "pass $w $c failed"; tail -5 gpurun_out/pmc_${w}_$c.log; exit 1; }
  done
done
python3 - <<'PY'
import glob, csv, json
for w, n in (("c3", 524288), ("c3big", 4194304)):
    out = {"source": "profiles/tools/pmc_c3.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), bench.py --workload %s --rollout-streams 1 --no-extras; kernel k_step_geometric<float,float,true,false,false,false>, one full-shard launch per control step" % w,
           "unit_note": "counter unit = KiB; FETCH_SIZE x2 on gfx950 (128-B requests tallied at 64 B, MI355X_MICROARCH.md section HBM)",
           "drones_per_launch_counted": n}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        tot, k = 0.0, 0
        for f in glob.glob(f"gpurun_out/pmc_{w}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c and "k_step_geometric" in r["Kernel_Name"]:
                    tot += float(r["Counter_Value"]); k += 1
        out[c + "_KiB_mean"] = tot / max(k, 1); out[c + "_launches"] = k
    out["read_bytes_corrected"] = out["FETCH_SIZE_KiB_mean"] * 1024 * 2      # gfx950: 128-B requests tallied at 64 B
    out["write_bytes"] = out["WRITE_SIZE_KiB_mean"] * 1024
    out["traffic_bytes_per_launch"] = out["read_bytes_corrected"] + out["write_bytes"]
    out["algorithmic_bytes_per_launch"] = 212 * n
    json.dump(out, open(f"gpurun_out/r02_pmc_traffic_{w}.json", "w"), indent=1)
    print(w, out)
PY
