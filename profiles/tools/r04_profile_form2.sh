# Round-4 evidence for the form-2 headline (k_rollout_geometric at config 3's full size).  Run on the GPU box from the repo root:
#   bash profiles/tools/r04_profile_form2.sh
#   kernel traces (--kernel-trace --stats) of the driver's command (auto launch form = 2: one 5-step warm-up launch + one 20-step launch)
#   and of a 2000-step run (50 steps per launch), and of the same in form 1;  HBM-side traffic (separate --pmc FETCH_SIZE / WRITE_SIZE
#   passes, --kernel-trace only) and SQ counters (SQ_INSTS_VALU, SQ_WAVES, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_WAIT_INST_ANY) of the 2000-step run.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04f2
mkdir -p $O
trace() { name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -- python3 bench.py "$@" --no-cpu-baseline --no-extras > $O/trace_$name.json 2> $O/trace_$name.err || { echo "trace $name failed"; tail -3 $O/trace_$name.err; }
  f=$(find $O/trace_$name -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_$name.csv
  echo "== $name"; head -3 $O/kernel_stats_$name.csv | cut -c1-220
}
trace driver_form2 --gpus 1 --steps 20 --warmup 5
trace long_form2 --gpus 1 --steps 2000 --warmup 200
trace driver_form1 --gpus 1 --steps 20 --warmup 5 --rollout-form 1
pmc() { name=$1; c=$2; shift 2
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${name}_$(echo $c | tr ' ' '_') -- python3 bench.py "$@" --no-cpu-baseline --no-extras > $O/pmc_${name}_$(echo $c | tr ' ' '_').log 2>&1 || { echo "pmc $name $c failed"; tail -3 $O/pmc_${name}_$(echo $c | tr ' ' '_').log; }
}
for c in FETCH_SIZE WRITE_SIZE; do pmc c3_form2 $c --gpus 1 --steps 2000 --warmup 200; done
pmc c3_form2 "SQ_INSTS_VALU SQ_WAVES" --gpus 1 --steps 2000 --warmup 200
pmc c3_form2 "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" --gpus 1 --steps 2000 --warmup 200
pmc c3_form2 "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" --gpus 1 --steps 2000 --warmup 200
for c in FETCH_SIZE WRITE_SIZE; do pmc c3_form2_driver $c --gpus 1 --steps 20 --warmup 5; done
python3 - <<'PY'
import glob, csv, json
O = "gpurun_out/r04f2"
n = 524288
def collect(tag):
    sq = {}
    for d in glob.glob(f"{O}/pmc_{tag}_*"):
        if not d.endswith(".log") and ("_driver_" in d) == tag.endswith("_driver"):
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    if "k_rollout_geometric" in r["Kernel_Name"]:
                        sq.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return {c: [v for _, v in sorted(rows)] for c, rows in sq.items()}
for tag, L, cmd, pick in (("c3_form2", 50, "--gpus 1 --steps 2000 --warmup 200", slice(None)), ("c3_form2_driver", 20, "--gpus 1 --steps 20 --warmup 5", slice(-1, None))):
    sq = collect(tag)
    rec = {"source": f"profiles/tools/r04_profile_form2.sh: rocprofv3 --pmc <counter> --kernel-trace (separate passes), bench.py {cmd} --no-extras (launch form 2: k_rollout_geometric<float, float, false, false, 0>, {L} control steps per launch, every step's observation written to the same [n, 20] array)" + ("; the timed 20-step launch only (the 5-step warm-up launch left out)" if L == 20 else ""),
           "unit_note": "FETCH_SIZE / WRITE_SIZE: counter unit = KiB; FETCH_SIZE x2 as for 16-byte-per-lane streaming reads (MI355X_MICROARCH.md section HBM)",
           "drones_per_launch_counted": n, "control_steps_per_launch": L}
    for c, vals in sq.items():
        v = vals[pick]
        rec[c + "_per_launch"], rec[c + "_launches"] = sum(v) / len(v), len(v)
    if "FETCH_SIZE" in sq and "WRITE_SIZE" in sq:
        rec["read_bytes_per_launch_corrected"] = rec["FETCH_SIZE_per_launch"] * 1024 * 2
        rec["write_bytes_per_launch"] = rec["WRITE_SIZE_per_launch"] * 1024
        rec["traffic_bytes_per_launch"] = rec["read_bytes_per_launch_corrected"] + rec["write_bytes_per_launch"]
        rec["algorithmic_bytes_per_launch"] = (80 * L + 132) * n
        rec["traffic_over_algorithmic"] = rec["traffic_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"]
    if "SQ_INSTS_VALU" in sq:
        rec["valu_wave_instructions_per_drone_step"] = rec["SQ_INSTS_VALU_per_launch"] / L / (n / 64)
    json.dump(rec, open(f"{O}/r04_pmc_traffic_c3_form2_L{L}.json", "w"), indent=1)
    print(json.dumps({k: v for k, v in rec.items() if k not in ("source", "unit_note")}, indent=1))
PY
for n in driver_form2 long_form2 driver_form1; do tail -1 $O/trace_$n.json | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('$n', '%.4g' % r['value'], 'us/step %.2f' % r['roofline']['us_per_step'], 'frac %.3f' % r['roofline']['frac'], r['roofline'].get('streams'), r['config'].get('launch_form'))"; done
