# HBM traffic of the C3 hot kernel: two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only,
# then the per-launch means -> gpurun_out/pmc_traffic_c3.json.  Run on the GPU box from the repo root.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_c3_$c -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/pmc_c3_$c.log 2>&1 || { echo "pass $c failed"; exit 1; }
done
python3 - <<'PY'
import glob, csv, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    tot, n = 0.0, 0
    for f in glob.glob(f"gpurun_out/pmc_c3_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and "k_step_geometric" in r["Kernel_Name"]:
                tot += float(r["Counter_Value"]); n += 1
    out[c + "_KiB_mean"] = tot / max(n, 1); out[c + "_launches"] = n
out["read_bytes_corrected"] = out["FETCH_SIZE_KiB_mean"] * 1024 * 2      # gfx950: 128-B requests tallied at 64 B
out["write_bytes"] = out["WRITE_SIZE_KiB_mean"] * 1024
out["traffic_bytes_per_launch"] = out["read_bytes_corrected"] + out["write_bytes"]
json.dump(out, open("gpurun_out/pmc_traffic_c3.json", "w"), indent=1)
print(out)
PY
