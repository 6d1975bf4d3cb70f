#!/bin/bash
# workgroup size of the persistent CBF kernel: NW = 4 / 8 / 16 wavefronts (tuning builds in build/, selected through MDS_LIB_PATH)
for nw in 4 8 16; do
  lib=build/libmds_nw$nw.so; [ $nw = 8 ] && lib=multidronesim_amd/libmds.so
  for sc in under level far; do
    MDS_LIB_PATH=$PWD/$lib timeout -k 10 300 python bench.py --workload c4 --c4-scene $sc --no-cpu-baseline --no-extras --fused-rollout 50 --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NW $nw $sc us/step %.2f sane %s' % (d['roofline']['us_per_step'], d['state_sane']))"
  done
done
