cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sa
timeout -k 10 420 python3 profiles/tools/r04_form_sweep.py gpurun_out/sa/r04_form_sweep.json 5 > gpurun_out/sa/form_sweep.log 2>&1 || { echo sweep failed; tail -5 gpurun_out/sa/form_sweep.log; exit 1; }
grep -c drones gpurun_out/sa/form_sweep.log
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/sa/driver.json 2> gpurun_out/sa/driver.err || { echo bench failed; tail -5 gpurun_out/sa/driver.err; exit 1; }
tail -1 gpurun_out/sa/driver.json | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('driver', '%.4g' % r['value'], 'us/step %.2f' % r['roofline']['us_per_step'], 'frac %.3f' % r['roofline']['frac'], r['config'].get('launch_form'), r.get('per_step'))"
timeout -k 10 500 bash profiles/tools/r04_profile_form2.sh > gpurun_out/sa/profile.log 2>&1; tail -40 gpurun_out/sa/profile.log
