# SQ counters of k_rollout_geometric at config 3's size (2000 steps, 50 per launch): where a wave's cycles go.  bash profiles/tools/r04_pmc_sq_form2.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04sq
mkdir -p $O
rocprofv3 --list-avail > $O/avail.txt 2>&1
grep -o "SQ_[A-Z_0-9]*" $O/avail.txt | sort -u | tr '\n' ' ' > $O/sq_names.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM SQ_WAVE_DEP_WAIT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F32" "SQ_THREAD_CYCLES_VALU SQ_IFETCH" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_')
  timeout -k 10 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$tag -- python3 bench.py --gpus 1 --steps 2000 --warmup 200 --no-cpu-baseline --no-extras > $O/pmc_$tag.log 2>&1 || echo "pmc $set failed"
done
python3 - <<'PY'
import glob, csv, json
O = "gpurun_out/r04sq"
acc = {}
for f in glob.glob(O + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_rollout_geometric" in r["Kernel_Name"]:
            a = acc.setdefault(r["Counter_Name"], [0.0, 0])
            a[0] += float(r["Counter_Value"]); a[1] += 1
out = {k: v[0] / v[1] for k, v in sorted(acc.items())}
json.dump({"what": "SQ counters of k_rollout_geometric<float, float, false, false, 0>, mean per 50-step launch of 524 288 drones (44 launches; bench.py --gpus 1 --steps 2000 --warmup 200)", "per_launch": out}, open(O + "/r04_pmc_sq_c3_form2.json", "w"), indent=1)
for k, v in out.items(): print("%-28s %.4g" % (k, v))
PY
