cd $GRAFT_REPO_ROOT
for r in 1 2; do for sc in under level; do for l in abl/libmds_f64w1.so multidronesim_amd/libmds.so; do
MDS_LIB_PATH=$PWD/$l timeout -k 10 300 python3 bench.py --workload c4 --c4-scene $sc --dtype float64 --no-cpu-baseline --no-extras --fused-rollout 50 --steps 200 --warmup 20 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sc $l us/step %.2f' % d['roofline']['us_per_step'], flush=True)"
done; done; done
